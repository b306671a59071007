#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_bench_config.py tests/test_gpu_graph.py tests/test_golden.py -x -q -m gpu > gpurun_out/r17.log 2>&1
rc=$?
tail -6 gpurun_out/r17.log
exit $rc
