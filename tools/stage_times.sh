#!/bin/bash
# one line of stage times of the exact joint pass per setting: bash tools/stage_times.sh "VAR=1" "VAR=2 OTHER=3" ...   (BENCH_EXTRA: more bench.py arguments)
out=gpurun_out/stage_times.txt
: > $out
for cfg in "$@"; do
  echo "== $cfg $BENCH_EXTRA" >> $out
  env $cfg timeout -k 10 400 python bench.py --no-cpu --no-dense-leg --steps 60 --warmup 10 --probe 3 $BENCH_EXTRA > gpurun_out/stage_one.log 2>&1 || { echo FAILED >> $out; tail -5 gpurun_out/stage_one.log >> $out; exit 1; }
  python - >> $out <<'PY'
import json
for l in open('gpurun_out/stage_one.log'):
    if l.startswith('{"metric"'):
        z = json.loads(l)
        st = z["roofline"]["exact_joint_pass"]["stages_ms"]
        print("ms_per_step %.3f" % z["ms_per_step"], {k: round(v, 3) for k, v in st.items()}, "to 1e-4:", z.get("convergence", {}).get("ms_to_1e-4"))
PY
done
