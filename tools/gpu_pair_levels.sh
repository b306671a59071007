#!/bin/bash
# which level of an exact joint pass the pair kernel gets wrong: C4 at size with one level at a time
mkdir -p gpurun_out
for m in ${MASKS:-1 2 4 8}; do
  echo "== SLIDE_CHOL_PAIR=$m"
  SLIDE_CHOL_PAIR=$m timeout -k 10 200 python tests/gpu_scenarios.py arrow_parity gpurun_out/r5_lvl_$m.json C4 2 ingest 0 1 > gpurun_out/r5_lvl_$m.log 2>&1
  echo "rc=$?"; tail -3 gpurun_out/r5_lvl_$m.log
  python -c "
import json,sys
try:
  z=json.load(open('gpurun_out/r5_lvl_$m.json')); print('gpu_vs_oracle', z['gpu_vs_oracle'], 'finite', z['finite'])
except Exception as e: print('no json', e)
import slide_slam_amd as s
"
done
