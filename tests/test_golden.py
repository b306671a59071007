"""Golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle):
CPU: the oracle still reproduces them; GPU (-m gpu): the HIP path reproduces them through the C-ABI."""
import glob
import os

import numpy as np
import pytest

from slide_slam_amd.replay import replay_single

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "replay_*.npz")))


def _load(path):
    z = np.load(path)
    log = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    return z, log


def _check(out, z, tol):
    for cls in ("cyl", "cube", "ell"):
        got = np.concatenate(out[cls + "_id"]) if len(z[cls + "_id"]) else np.zeros(0, np.int32)
        assert np.array_equal(got, z[cls + "_id"]), f"{cls} ids differ from the golden fixture"
    a, b = np.array(out["pose7"]), z["pose7"]
    rel = np.linalg.norm(a[:, :3] - b[:, :3], axis=1) / np.maximum(np.linalg.norm(b[:, :3], axis=1), 1e-9)
    assert rel.max() <= tol
    qd = np.minimum(np.linalg.norm(a[:, 3:] - b[:, 3:], axis=1), np.linalg.norm(a[:, 3:] + b[:, 3:], axis=1))
    assert qd.max() <= tol


@pytest.mark.parametrize("path", [f for f in FIXTURES if "C2" not in f], ids=os.path.basename)
def test_oracle_reproduces_golden(path):
    from oracle import pyoracle as po
    z, log = _load(path)
    ob = po.OracleBackend(po.OrcParams.default(), 1)
    out = replay_single(ob, log)
    _check(out, z, 1e-9)
    c = ob.counts()
    assert [c["cyl"], c["cube"], c["point"], c["factors"]] == list(z["counts"])


def _expmap_outputs(path):
    """Outputs of the same replay under the Expmap chart (tests/golden/chart_expmap_<preset>.npz, make_golden.py --chart=expmap) or None."""
    alt = os.path.join(os.path.dirname(path), os.path.basename(path).replace("replay_", "chart_expmap_"))
    return np.load(alt) if os.path.exists(alt) else None


@pytest.mark.parametrize("path", [f for f in FIXTURES if "small" in f], ids=os.path.basename)
def test_oracle_reproduces_golden_under_the_expmap_chart(path):
    """The chart question stays open (the reference's only numeric evidence passes under Expmap, its comment names Cayley):
    both charts are pinned by fixtures.  The two charts' replays are NOT the same numbers (iSAM2's relinearisation threshold leaves
    second-order terms of the retraction in the estimate), so a chart mix-up cannot pass."""
    from oracle import pyoracle as po
    z, log = _load(path)
    ze = _expmap_outputs(path)
    assert ze is not None
    ob = po.OracleBackend(po.OrcParams.default(pose_chart=po.CHART_EXPMAP), 1)
    out = replay_single(ob, log)
    _check(out, ze, 1e-9)
    assert np.abs(ze["pose7"] - z["pose7"]).max() > 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("chart", ["cayley", "expmap"])
@pytest.mark.parametrize("path", FIXTURES, ids=os.path.basename)
def test_gpu_reproduces_golden(gpu, path, chart):
    """Identical landmark-id associations, optimised poses <= 1e-4 relative (BASELINE.json north_star) — under BOTH Pose3 charts
    (SLIDE_CHART_CAYLEY: what cubeFactor.h:96-97 names, the default; SLIDE_CHART_EXPMAP: GTSAM_POSE3_EXPMAP builds)."""
    z, log = _load(path)
    kw = {}
    if chart == "expmap":
        z = _expmap_outputs(path)
        if z is None:
            pytest.skip("no Expmap fixture for this preset")
        kw = dict(pose_chart=gpu.api.CHART_EXPMAP)
    gb = gpu.SlideBackend(gpu.default_params(**kw), 1)
    out = replay_single(gb, log)
    _check(out, z, 1e-4)
    c = gb.counts()
    assert [c["cyl"], c["cube"], c["point"], c["factors"]] == list(z["counts"])
    for cls, key in ((0, "lm_cyl"), (1, "lm_cube"), (2, "lm_point")):
        ref = z[key]
        for i in range(0, len(ref), max(1, len(ref) // 9)):
            _, got = gb.graph.get_landmark(cls, i)
            assert np.linalg.norm(got - ref[i]) <= 1e-4 * max(np.linalg.norm(ref[i]), 1.0)
