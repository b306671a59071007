#!/bin/bash
# border-product scheduling variants through bench.py's stage timing (one line per variant in gpurun_out/syrk_sweep.txt)
out=gpurun_out/syrk_sweep.txt
: > $out
run() {
  echo "== $*" >> $out
  env "$@" timeout -k 10 400 python bench.py --no-cpu --no-dense-leg --steps 60 --warmup 10 --probe 3 > gpurun_out/syrk_one.log 2>&1 || { echo FAILED >> $out; tail -5 gpurun_out/syrk_one.log >> $out; return 1; }
  python - >> $out <<'PY'
import json
for l in open('gpurun_out/syrk_one.log'):
    if l.startswith('{"metric"'):
        z = json.loads(l)
        st = z["roofline"]["exact_joint_pass"]["stages_ms"]
        print("ms_per_step %.3f" % z["ms_per_step"], {k: round(v, 3) for k, v in st.items()}, "parity", z.get("parity"))
PY
}
run SLIDE_SYRK_PLAIN=1 && run SLIDE_SYRK_NOXCD=1 && run SLIDE_SYRK_LDS=0 && run SLIDE_SYRK_LDS=32768 && run SLIDE_SYRK_LDS=65536 SLIDE_SYRK_NOXCD=1
