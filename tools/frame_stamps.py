# Phase timing of k_assoc_frame (one class's workgroup) on a streaming replay, from the -DSLIDE_STAMPS experiment build:
#   SLIDE_STAMP_BLOCK=<0 cylinders | 1 cubes | 2 points> python -m slide_slam_amd.build --stamps --force
#   SLIDE_LIB_VARIANT=exp_stamps python3 tools/frame_stamps.py [frames]
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slide_slam_amd as s
from slide_slam_amd import api
from slide_slam_amd.synth import SynthConfig, make_world, make_robot_log, frame_detections
from slide_slam_amd.replay import IDENT7
cfg = SynthConfig.preset("C4")
log = make_robot_log(cfg, make_world(cfg), 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
b = s.SlideBackend(s.default_params(), 1)
prev = IDENT7.copy()
acc = []
for k in range(n):
    r = b.process_frame(0, log["rel7"][k], prev, frame_detections(log, k), 0)
    prev = r["pose7"].copy()
    if k >= n - 50:
        st = (C.c_ulonglong * 16)(); api.lib().slide_debug_assoc_stamps(st)
        acc.append(np.array(st[:], dtype=np.float64))
a = np.array(acc)
t = a[:, :7]
names = ["start", "distance words", "select", "compaction + padding", "sort", "staged", "matching done"]
print("counts (cyl, cube, point):", b.counts(), " mean over the last", len(acc), "frames, us:")
for i in range(1, 7):
    print(f"  {names[i]:22s} +{np.mean(t[:, i] - t[:, i - 1]) * 0.01:7.2f}   (at {np.mean(t[:, i] - t[:, 0]) * 0.01:7.2f})")
