"""Where does the first exact pass under the Expmap chart differ between GPU and oracle at C4 size?  (GPU box.)
Prints, per robot: the distance of the poses BEFORE the first joint pass (the robots' own ingest solves) and after passes 1 and 2,
the number of relinearised variables of the ingest solve on both sides."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import slide_slam_amd as s
from oracle import pyoracle as po
from slide_slam_amd.distributed import PassDriver, gpu_matcher, setup_local_shards
from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world, frame_detections
from dist_worker import oracle_matcher
preset = sys.argv[1] if len(sys.argv) > 1 else "C4"
chart = 1 if (len(sys.argv) > 2 and sys.argv[2] == "expmap") else 0
chart_o = (1 if sys.argv[3] == "expmap" else 0) if len(sys.argv) > 3 else chart      # the oracle's chart, when it is to differ
cfg = SynthConfig.preset(preset)
if os.environ.get("DIAG_MIX"):      # class mix override: "cyl,cube,ell"
    import dataclasses
    cfg = dataclasses.replace(cfg, class_mix=tuple(float(x) for x in os.environ["DIAG_MIX"].split(",")))
wm = make_world(cfg)
logs = [make_robot_log(cfg, wm, r) for r in range(cfg.robots)]
P = cfg.poses_per_robot; R = len(logs)
L = po.lib(native=True)
O, A = [], []
for lg in logs:
    o = po.OracleBackend(po.OrcParams.default(num_threads=16, pose_chart=chart_o), 1, L=L)
    a = s.SlideBackend(s.default_params(pose_chart=chart), 1)
    for k in range(P):
        o.process_frame(0, lg["rel7"][k], lg["gt7"][k], frame_detections(lg, k), 2)
        a.process_frame(0, lg["rel7"][k], lg["gt7"][k], frame_detections(lg, k), s.FRAME_FOREIGN)
    assert o.ingest_solve() == 0 and a.ingest_solve() == 0
    O.append(o); A.append(a)
poses = lambda sh: np.array([[x.graph.get_pose12(0, k)[1] for k in range(P)] for x in sh])
def show(tag):
    a, o = poses(A), poses(O)
    d = np.linalg.norm((a - o).reshape(R, -1), axis=1) / np.linalg.norm(o.reshape(R, -1), axis=1)
    w = np.abs(a - o).reshape(R, P, 12).max(axis=2)
    print(tag, " ".join(f"{x:.1e}" for x in d), "| worst pose per robot:", [int(w[r].argmax()) for r in range(R)], flush=True)
    rr = np.abs(a - o).reshape(R, P, 12)
    print("      max |dR| per robot:", " ".join(f"{rr[r, :, :9].max():.1e}" for r in range(R)), "| max |dt| [m]:", " ".join(f"{rr[r, :, 9:].max():.1e}" for r in range(R)), flush=True)
    for cls, nm in ((0, "cyl"), (1, "cube"), (2, "ell")):
        out = []
        for r in range(R):
            n = A[r].counts()[("cyl", "cube", "point")[cls]]
            dm = 0.0
            for i in range(min(n, 400)):
                try:
                    ga = np.asarray(A[r].graph.get_landmark(cls, i)[1]); go = O[r].graph.get_landmark(cls, i)
                    go = np.asarray(go[1] if isinstance(go, tuple) else go)
                except Exception:
                    continue
                m = min(len(ga), len(go))
                dm = max(dm, float(np.abs(ga[:m] - go[:m]).max()))
            out.append(dm)
        print(f"      max |d landmark| {nm}:", " ".join(f"{x:.1e}" for x in out), flush=True)
print("relinearised at the ingest solve (gpu / oracle):", [(a.graph.stats().get("n_relin"), o.graph.stats().get("n_relin")) for a, o in zip(A, O)])
show("before pass 1:")
dev = torch.device("cuda", torch.cuda.current_device())
batch = s.CholBatch(R)
for t, a in enumerate(A):
    a.graph.join_chol_batch(batch, t)
bufA, infoA = setup_local_shards(A, gpu_matcher, device=dev)
bufO, infoO = setup_local_shards(O, oracle_matcher)
dA = PassDriver(A, bufA, infoA["n_slots"], batch=batch, device=dev, arrow=True, sep_dim=infoA["sep_dim"], sep_prof=infoA.get("sep_prof"))
dO = PassDriver(O, bufO, infoO["n_slots"], arrow=True, sep_dim=infoO["sep_dim"], sep_prof=infoO.get("sep_prof"))
show("after the merge:")
for i in range(3):
    dA.one_pass(); dO.one_pass()
    show(f"after pass {i + 1}: ")
