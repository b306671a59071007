#!/bin/bash
# diagnostic: step kernels vs pair kernel on the systems of a C4 exact pass, tile by tile (SLIDE_PAIR_VERIFY=1, one un-captured pass)
mkdir -p gpurun_out
for m in ${MASKS:-2 1}; do
  echo "== SLIDE_CHOL_PAIR=$m"
  SLIDE_CHOL_PAIR=$m SLIDE_PAIR_VERIFY=1 timeout -k 10 200 python tests/gpu_scenarios.py pair_verify gpurun_out/r5_vfy_$m.json C4 1 > gpurun_out/r5_vfy_$m.log 2>&1
  echo "rc=$?"; grep "pair verify" gpurun_out/r5_vfy_$m.log | head -50; tail -2 gpurun_out/r5_vfy_$m.log
done
