"""Diagnostic: the bounded back-substitution frame by frame, product vs oracle (kept blocks, first dirty column) on golden C2."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
torch.zeros(1, device="cuda:0")
import slide_slam_amd as s  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from slide_slam_amd.replay import IDENT7  # noqa: E402
from slide_slam_amd.synth import frame_detections  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "replay_C2.npz"))
log = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
gb = s.SlideBackend(s.default_params(), 1)
gb.graph.set_wildfire(1e-3)
ob = po.OracleBackend(po.OrcParams.default(num_threads=8), 1)
ob.graph.set_wildfire(1e-3)
pg = po_ = IDENT7.copy()
rows = []
for k in range(n):
    det = frame_detections(log, k)
    rg = gb.process_frame(0, log["rel7"][k], pg, det, 0)
    ro = ob.process_frame(0, log["rel7"][k], po_, det, 0)
    pg, po_ = rg["pose7"].copy(), ro["pose7"].copy()
    wg, wo = gb.graph.wildfire_stats(), ob.graph.wildfire_stats()
    ig = gb.graph.incremental_stats()
    rows.append((k, wg["kept_last"], wo["kept_last"], ig["last_first_column"], wo["last_cd"], ig["block_columns"], ig["full"]))
bad = [r for r in rows if r[1] != r[2]]
print("frames", n, "differing", len(bad), "kept", sum(r[1] for r in rows), sum(r[2] for r in rows))
print("frame gpu_kept orc_kept gpu_first_col orc_cd T full_count")
for r in bad[:60]:
    print(*r)
