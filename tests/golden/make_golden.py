#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ by running the ORACLE (CPU restatement of the reference;
the reference itself cannot be built or imported here — SURVEY.md §8c) on seeded synthetic frame logs.

  python tests/golden/make_golden.py tiny small        # seconds
  python tests/golden/make_golden.py C2                # minutes (500-pose replay, one solve per frame)
  python tests/golden/make_golden.py --chart=expmap small C2     # the same replays under the Expmap chart, outputs only

Each .npz holds the INPUT frame log (so the fixture does not depend on numpy's RNG stream) and the expected
outputs: per-frame optimised pose7, per-detection landmark ids, final landmark estimates."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from slide_slam_amd.replay import replay_single  # noqa: E402
from slide_slam_amd.synth import SynthConfig, make_dataset  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make(preset, threads=1, chart=None):
    """chart = "expmap": the same replay under the GTSAM_POSE3_EXPMAP chart (SLIDE_CHART_EXPMAP), OUTPUTS ONLY, into
    chart_expmap_<preset>.npz (the inputs are those of replay_<preset>.npz)."""
    data = make_dataset(SynthConfig.preset(preset))
    log = data["logs"][0]
    kw = dict(pose_chart=po.CHART_EXPMAP) if chart == "expmap" else {}
    ob = po.OracleBackend(po.OrcParams.default(num_threads=threads, **kw), 1, L=po.lib(native=threads > 1))
    t0 = time.time()
    out = replay_single(ob, log)
    dt = time.time() - t0
    cnt = ob.counts()
    lm = {}
    for cls, n, key in ((0, cnt["cyl"], "lm_cyl"), (1, cnt["cube"], "lm_cube"), (2, cnt["point"], "lm_point")):
        lm[key] = np.array([ob.graph.get_landmark(cls, i)[1] for i in range(n)]) if n else np.zeros((0, (7, 15, 3)[cls]))
    np.savez_compressed(
        os.path.join(HERE, f"chart_expmap_{preset}.npz" if chart == "expmap" else f"replay_{preset}.npz"),
        **({} if chart == "expmap" else {"in_" + k: v for k, v in log.items()}),
        pose7=np.array(out["pose7"]),
        cyl_id=np.concatenate(out["cyl_id"]) if out["cyl_id"] else np.zeros(0, np.int32),
        cube_id=np.concatenate(out["cube_id"]) if out["cube_id"] else np.zeros(0, np.int32),
        ell_id=np.concatenate(out["ell_id"]) if out["ell_id"] else np.zeros(0, np.int32),
        counts=np.array([cnt["cyl"], cnt["cube"], cnt["point"], cnt["factors"]]), **lm)
    print(f"{preset}: {len(out['pose7'])} frames in {dt:.1f}s, counts {cnt}")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    chart = "expmap" if "--chart=expmap" in sys.argv else None
    for p in args or ["tiny", "small"]:
        make(p, threads=8 if p.startswith("C") else 1, chart=chart)
