#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu --no-parity --no-dense-leg --probe 0 2> gpurun_out/r20.err | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["exact_joint_pass"]["stages_ms"], d["dense_relmeas"]["ms_per_step"])' || exit 1
timeout -k 10 1000 python -m pytest tests/test_bench_config.py tests/test_gpu_graph.py -x -q -m gpu > gpurun_out/r20.log 2>&1
rc=$?
tail -4 gpurun_out/r20.log
exit $rc
