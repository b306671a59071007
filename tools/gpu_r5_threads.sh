#!/bin/bash
# diagnosis of the 8-thread rehearsal (numerical failure, not a device fault): which switch matters
mkdir -p gpurun_out
i=0
for v in "SLIDE_CHOL_XCD=1" "SLIDE_CHOL_XCD=0" "SLIDE_CHOL_XCD=1 SLIDE_TEST_THREAD_DIRECT=1" "SLIDE_CHOL_XCD=0 SLIDE_TEST_THREAD_DIRECT=1"; do
  i=$((i+1))
  env $v timeout -k 10 120 python tests/gpu_scenarios.py rank_threads gpurun_out/r5_thr_$i.json C4 8 3 0 > gpurun_out/r5_thr_$i.log 2>&1
  echo "$v rc=$? $(tail -1 gpurun_out/r5_thr_$i.log | cut -c1-200)"
done
