#!/bin/bash
# association kernel: tests, phase stamps, launch time (both workgroup sizes)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_graph.py tests/test_golden.py -x -q -m gpu -k "assoc or knn or match or submap or replay or golden or stream" > gpurun_out/r_assoc.log 2>&1
rc=$?
tail -5 gpurun_out/r_assoc.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 120 python tools/assoc_stamps.py 1 && timeout -k 10 120 python tools/assoc_stamps.py 8192 && timeout -k 10 200 python tools/assoc_time.py
