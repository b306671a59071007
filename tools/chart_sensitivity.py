"""How far do ulp-level differences in the chart's transcendental functions move an exact joint pass?  (CPU only, oracle only.)

The GPU path and the oracle evaluate Pose3's Expmap / Logmap with the same formulas and branch thresholds (slide_slam_amd/csrc/sl_math.hpp,
oracle/lie.hpp), but sin / acos / tan come from different libraries (the device's ocml, the host's glibc), each correct to an ulp or so.
At C4 size the FIRST pass from the raw ingest state differs by 4.3e-6 between GPU and oracle under the Expmap chart (2e-8 from the second
pass on; 4e-8 throughout under Cayley, which has no transcendental function).  This script takes the oracle twice — as built, and with
every sin / acos / tan of the chart returning the next representable number (-DORC_PERTURB_TRIG) — through the same passes and prints
the distance between the two: the sensitivity of a pass to exactly that kind of difference.

  python tools/chart_sensitivity.py [preset] [passes]"""
import os, subprocess, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po
from slide_slam_amd.distributed import PassDriver, setup_local_shards
from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world, frame_detections
from dist_worker import oracle_matcher

preset = sys.argv[1] if len(sys.argv) > 1 else "C4"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
here = os.path.join(ROOT, "oracle")
flags = "-O3 -march=x86-64-v3 -ffp-contract=off -fopenmp -std=c++17 -fPIC -Wall -Wno-unused-function"
# third argument "numdiff": the second library differs in the step of the numerical Jacobians instead (1.00001e-6 for 1e-6): the
# rounding noise of the cube / cylinder factors' central differences changes, and with it the robots' ingest solves at the 1e-8 .. 1e-7
# level — two equally valid implementations of the same algorithm that START the joint passes a little apart, as GPU and oracle do
mode = sys.argv[3] if len(sys.argv) > 3 else "trig"
define = "-DORC_PERTURB_TRIG" if mode == "trig" else "-DORC_NUMDIFF_DELTA=1.00001e-6"
subprocess.run(["make", "-C", here, "OUT=_build/liboracle_perturb.so", f"CXXFLAGS={flags} {define}", "-B"], check=True, stdout=subprocess.DEVNULL)
Lp = C.CDLL(os.path.join(here, "_build", "liboracle_perturb.so"))
for f in ("orc_graph_create", "orc_backend_create", "orc_backend_graph"):
    getattr(Lp, f).restype = C.c_void_p
L0 = po.lib()
cfg = SynthConfig.preset(preset); wm = make_world(cfg)
logs = [make_robot_log(cfg, wm, r) for r in range(cfg.robots)]
P = cfg.poses_per_robot
ncpu = min(os.cpu_count() or 1, 16)
for chart, name in ((0, "cayley"), (1, "expmap")):
    res = []
    for L in (L0, Lp):
        sh = []
        for lg in logs:
            o = po.OracleBackend(po.OrcParams.default(num_threads=ncpu, pose_chart=chart), 1, L=L)
            for k in range(P):
                o.process_frame(0, lg["rel7"][k], lg["gt7"][k], frame_detections(lg, k), 2)
            assert o.ingest_solve() == 0
            sh.append(o)
        bufs, info = setup_local_shards(sh, oracle_matcher)
        drv = PassDriver(sh, bufs, info["n_slots"], arrow=True, sep_dim=info["sep_dim"], sep_prof=info.get("sep_prof"))
        per = [np.array([[x.graph.get_pose12(0, k)[1] for k in range(P)] for x in sh])]      # before the first joint pass
        for _ in range(passes):
            drv.one_pass()
            per.append(np.array([[x.graph.get_pose12(0, k)[1] for k in range(P)] for x in sh]))
        res.append(per)
    d = [float((np.linalg.norm((a - b).reshape(len(logs), -1), axis=1) / np.linalg.norm(a.reshape(len(logs), -1), axis=1)).max())
         for a, b in zip(*res)]
    print(f"{preset} {name}: oracle vs oracle ({mode} variant), poses before the first joint pass, then after pass 1..{passes}: " + ", ".join(f"{x:.2e}" for x in d), flush=True)
