"""CPU: the oracle's restatement of iSAM2's bounded back-substitution (oracle/graph.hpp Graph::wildfire_bound; [GTSAM]
ISAM2GaussNewtonParams::wildfireThreshold of the ISAM2 the reference runs, graph.cpp:15-18, 260-272) and the frame-by-frame
cross-robot association table (distributed.associate_by_ingest, sloamNode.cpp:912-1002)."""
import os

import numpy as np

from oracle import pyoracle as po
from slide_slam_amd.distributed import associate_by_ingest
from slide_slam_amd.replay import replay_single

HERE = os.path.dirname(os.path.abspath(__file__))


def _log(name):
    z = np.load(os.path.join(HERE, "golden", name))
    return z, {k[3:]: z[k] for k in z.files if k.startswith("in_")}


def test_oracle_wildfire_off_is_the_fixture_and_on_keeps_blocks():
    """Threshold 0 (the default) and an explicit 0 reproduce the golden replay bit for bit; with the reference's 1e-3 the associations
    are unchanged, blocks are kept in most updates once the chain is long, and every frame's returned pose (the newest key frame: always
    in the re-solved part) stays within the threshold's order of the exact update's."""
    z, log = _log("replay_small.npz")
    runs = {}
    for tag, thr in (("default", None), ("zero", 0.0), ("on", 1e-3)):
        ob = po.OracleBackend(po.OrcParams.default(), 1)
        if thr is not None:
            ob.graph.set_wildfire(thr)
        runs[tag] = (replay_single(ob, log), ob.graph.wildfire_stats())
    d, zero, on = runs["default"], runs["zero"], runs["on"]
    assert d[1]["kept_total"] == 0 and zero[1]["kept_total"] == 0
    assert np.array_equal(np.array(d[0]["pose7"]), np.array(zero[0]["pose7"]))
    assert np.array_equal(np.array(d[0]["pose7"]), z["pose7"][: len(d[0]["pose7"])])
    for cls in ("cyl", "cube", "ell"):
        assert all(np.array_equal(a, b) for a, b in zip(d[0][cls + "_id"], on[0][cls + "_id"]))
    assert on[1]["kept_total"] > len(log["rel7"])          # more than a block per update on average
    assert np.abs(np.array(on[0]["pose7"]) - np.array(d[0]["pose7"])).max() < 5e-3


def test_wildfire_rule_on_a_hand_made_chain():
    """The rule itself: T = 4 blocks, the first dirty block column 3, profile c -> c + 1.  Block 2 depends on block 3 (re-solved, moved by
    less than the threshold) -> it is quiet and keeps its old value, and so do blocks 1 and 0 although block 1's dependency (block 2)
    would have moved a lot: the stop is at the HIGHEST quiet block below the dirty column."""
    import ctypes as C
    L = po.lib()
    if not hasattr(L, "orc_wildfire_rule"):
        import pytest
        pytest.skip("liboracle was built without the rule's test hook")
    T, NB = 4, 64
    prev = np.arange(T * NB, dtype=np.float64) * 1e-2
    new = prev.copy()
    new[3 * NB:] += 5e-4                                   # block 3 moved by less than 1e-3
    new[2 * NB:3 * NB] += 0.5                              # block 2 would have moved a lot
    new[:2 * NB] -= 0.25
    prof_last = np.array([1, 2, 3, 3], np.int32)
    out = new.copy()
    kept = L.orc_wildfire_rule(out.ctypes.data_as(C.c_void_p), prev.ctypes.data_as(C.c_void_p), C.c_int(T * NB), C.c_int(T),
                               prof_last.ctypes.data_as(C.c_void_p), C.c_int(3), C.c_double(1e-3))
    assert kept == 3
    assert np.array_equal(out[:3 * NB], prev[:3 * NB]) and np.array_equal(out[3 * NB:], new[3 * NB:])


def test_associate_by_ingest_tables():
    """Two robots, one class: robot 1's landmark 0 is the replica's 1 (shared with robot 0), its landmark 1 a new one; a split (one
    local landmark on two replica landmarks) and a collapse (two local landmarks on one) are counted and resolved first-come."""
    e = np.zeros(0, np.int64)
    own = [[[np.array([0, 1]), np.array([1, 2])], [], []], [[np.array([0]), np.array([0, 1])], [], []]]
    rep = [[[np.array([0, 1]), np.array([1, 2])], [], []], [[np.array([1]), np.array([1, 3])], [], []]]
    gid, n_glob, st = associate_by_ingest(own, rep)
    assert gid[0][0].tolist() == [0, 1, 2] and gid[1][0].tolist() == [1, 3] and n_glob == [4, 0, 0] and st == dict(split=0, collapsed=0)
    rep2 = [[[np.array([0, 1]), np.array([5, 2])], [], []], [[np.array([1]), np.array([1, 1])], [], []]]      # robot 0: landmark 1 split; robot 1: 0 and 1 collapse
    gid2, n2, st2 = associate_by_ingest(own, rep2)
    assert st2 == dict(split=1, collapsed=1)
    assert gid2[0][0].tolist() == [0, 1, 2] and gid2[1][0][0] == 1 and gid2[1][0][1] >= 6 and n2[0] == 7
    assert all(len(g) == 0 for r in (0, 1) for g in gid[r][1:]) and e.size == 0
