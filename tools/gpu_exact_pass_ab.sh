#!/bin/bash
# exact-pass parity tests + stage times of the default bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_bench_config.py -x -q -m gpu -k "exact_joint_step and not rccl" > gpurun_out/r5_exact.log 2>&1
rc=$?
tail -8 gpurun_out/r5_exact.log
if [ $rc -ne 0 ]; then exit 1; fi
B="python bench.py --steps 100 --warmup 10 --no-cpu --no-parity --no-dense-leg --probe 0 --no-dense-relmeas --no-place-leg"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d["roofline"]["exact_joint_pass"]; print(sys.argv[1], round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in e["stages_ms"].items()})'
for v in ${VARIANTS:-X=0}; do
  env $v timeout -k 10 300 $B 2> gpurun_out/r5_exact_bench.err | python -c "$pick" "$v" || exit 1
done
