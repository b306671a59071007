#!/bin/bash
# streaming path: golden replays + incremental / wildfire tests, then the per-frame time with and without the prediction
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_golden.py tests/test_gpu_graph.py -x -q -m gpu > gpurun_out/r5_stream_tests.log 2>&1
rc=$?
tail -6 gpurun_out/r5_stream_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
for v in SLIDE_NO_LIN_SKIP=1 SLIDE_NO_LIN_SKIP=0; do
env $v timeout -k 10 300 python - <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import torch; torch.zeros(1, device='cuda:0')
import slide_slam_amd as s
from slide_slam_amd.replay import replay_single
from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world
cfg = SynthConfig.preset("C4")
wm = make_world(cfg)
lg = make_robot_log(cfg, wm, 0)
for wf in (0.0, 1e-3):
    gb = s.SlideBackend(s.default_params(), 1)
    if wf: gb.graph.set_wildfire(wf)
    out = replay_single(gb, lg, collect=False)
    t = np.array(out["t_frame"]) * 1e3
    print(os.environ.get("SLIDE_NO_LIN_SKIP"), "wildfire", wf, "ms/frame mean %.4f last100 %.4f max %.3f" % (t.mean(), t[-100:].mean(), t.max()), gb.graph.incremental_stats())
PY
done
