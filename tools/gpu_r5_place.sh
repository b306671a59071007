#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_place.py tests/test_host_entry_points_gpu.py -x -q > gpurun_out/r5_place_tests.log 2>&1
rc=$?
tail -6 gpurun_out/r5_place_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
for v in SLIDE_PLACE_PLAIN=1 SLIDE_PLACE_PLAIN=0; do
env $v timeout -k 10 300 python - <<'PY'
import json, os, sys
sys.path.insert(0, '.')
import torch; torch.zeros(1, device='cuda:0')
import slide_slam_amd as s
import bench
r = bench.place_roofline(s, with_cpu=(os.environ.get('SLIDE_PLACE_PLAIN') == '0'))
for k in ('reference_indoor_maps', 'synthetic_forest_792'):
    print(os.environ.get('SLIDE_PLACE_PLAIN'), k, json.dumps(r[k]))
PY
done
