#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "assoc or submap or knn" > gpurun_out/r5_assoc_ab_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r5_assoc_ab_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
for v in ${VARIANTS:-SLIDE_ASSOC_REG=0 SLIDE_ASSOC_REG=1}; do
  env $v timeout -k 10 200 python tools/assoc_ab.py 2>/dev/null | tail -1 || exit 1
done
