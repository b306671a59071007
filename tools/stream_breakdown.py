"""Where a streaming frame's wall time goes (robot 0 of C4, 625 frames, the bench's stream_replay): the C-ABI's own split (ms_association =
upload + k_assoc_frame + read-back + updateMap; ms_graph = factors + update + map refresh + pose) and the Python wrapper's share, means
over the last 100 frames; with the graph's kernel profile (device time per stage) from a second replay."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slide_slam_amd as s
from slide_slam_amd.synth import SynthConfig, make_world, make_robot_log, frame_detections
from slide_slam_amd.replay import IDENT7

cfg = SynthConfig.preset("C4")
log = make_robot_log(cfg, make_world(cfg), 0)
out = {}
for prof in (False, True):
    b = s.SlideBackend(s.default_params(), 1)
    if prof:
        b.graph.set_profiling(True)
    prev = IDENT7.copy()
    ta, tg, tw = [], [], []
    for k in range(len(log["rel7"])):
        det = frame_detections(log, k)
        t0 = time.perf_counter()
        r = b.process_frame(0, log["rel7"][k], prev, det, 0)
        t1 = time.perf_counter()
        prev = r["pose7"].copy()
        ta.append(r["t_assoc"] * 1e3); tg.append(r["t_graph"] * 1e3); tw.append((t1 - t0) * 1e3)
    key = "profiled" if prof else "plain"
    out[key] = dict(ms_frame=float(np.mean(tw[-100:])), ms_association=float(np.mean(ta[-100:])), ms_graph=float(np.mean(tg[-100:])),
                    ms_wrapper=float(np.mean(tw[-100:]) - np.mean(ta[-100:]) - np.mean(tg[-100:])))
    if prof:
        p = b.graph.get_profile()
        out[key]["kernel_us_per_frame"] = {n: dict(us=v["ms"] * 1e3 / len(tw), launches_per_frame=v["launches"] / len(tw)) for n, v in p.items()}
print(json.dumps(out, indent=1))
