#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_place.py tests/test_golden.py -x -q -m gpu > gpurun_out/r19a.log 2>&1
rc=$?
tail -4 gpurun_out/r19a.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 120 python tools/assoc_stamps.py 1 && timeout -k 10 120 python tools/assoc_stamps.py 8192 && timeout -k 10 200 python tools/assoc_time.py || exit 1
timeout -k 10 900 python -m pytest tests/test_bench_config.py -x -q -m gpu -k "dense or rank or c4" > gpurun_out/r19b.log 2>&1
rc=$?
tail -4 gpurun_out/r19b.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu --no-parity --no-dense-leg --probe 0 2> gpurun_out/r19c.err | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["dense_relmeas"]["ms_per_step"])'
