#!/bin/bash
# kernel-trace + stats of a short default bench; prints the top kernels (run on the GPU box from the repo root)
set -o pipefail
OUT=gpurun_out/kstats
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-parity --no-dense-leg ${BENCH_ARGS} > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/kstats/trace/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    n = r['Name']
    if any(k in n for k in ('chain', 'pcg', 'k_chol_step_batched', 'k_chol_bwd_chain_batched', 'k_schur_b', 'k_sum')):
        print(n[:70].ljust(70), r['Calls'].rjust(6), ("%.1f" % (float(r['AverageNs']) / 1e3)).rjust(9), 'us avg', r['Percentage'])
PY
rm -rf $OUT/trace
