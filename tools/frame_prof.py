import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import slide_slam_amd as s
from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world, frame_detections
cfg = SynthConfig.preset("C4"); w = make_world(cfg); log = make_robot_log(cfg, w, 0)
gb = s.SlideBackend(s.default_params(), 1)
prev = np.array([0,0,0,0,0,0,1.0]); ta=[]; tg=[]; tt=[]
for k in range(len(log["rel7"])):
    det = frame_detections(log, k)
    t0=time.perf_counter(); r = gb.process_frame(0, log["rel7"][k], prev, det, 0); t1=time.perf_counter()
    prev = r["pose7"].copy(); ta.append(r["t_assoc"]); tg.append(r["t_graph"]); tt.append(t1-t0)
ta,tg,tt = map(np.array,(ta,tg,tt))
for a,b in ((0,100),(100,300),(300,500),(500,625)):
    print(f"frames {a}-{b}: total {tt[a:b].mean()*1e3:.3f} ms  assoc {ta[a:b].mean()*1e3:.3f}  graph {tg[a:b].mean()*1e3:.3f}  python+other {(tt[a:b]-ta[a:b]-tg[a:b]).mean()*1e3:.3f}")
g = gb.graph
g.set_profiling(True)
det = frame_detections(log, 624)
st = g.stats(); print(st)
for _ in range(3): g.gauss_newton(1)
print({k: round(v["ms"]/3,4) for k,v in g.get_profile().items()})
print("incremental:", g.incremental_stats())
