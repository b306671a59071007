"""GPU parity of the inter-robot map-to-map association (SlideMatch sweep, CLIPPER affinity) vs the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OPlace(C.Structure):
    _fields_ = [("dilation_factor", C.c_double), ("xy_step", C.c_double), ("yaw_half_range", C.c_double),
                ("yaw_step", C.c_double), ("match_threshold", C.c_double), ("match_threshold_dimension", C.c_double),
                ("disable_yaw_search", C.c_int), ("ignore_dimension", C.c_int), ("min_num_inliers", C.c_int),
                ("use_lsq", C.c_int), ("min_num_map_objects_to_start", C.c_int), ("max_rings", C.c_int)]


def _both_params(gpu, **kw):
    gp = gpu.place_default_params(**kw)
    op = OPlace(gp.dilation_factor, gp.search_xy_step_size, gp.match_yaw_half_range, gp.search_yaw_step_size,
                gp.match_threshold_position, gp.match_threshold_dimension, gp.disable_yaw_search, gp.ignore_dimension,
                gp.min_num_inliers, gp.use_nonlinear_least_squares, gp.min_num_map_objects_to_start, gp.max_rings)
    return gp, op


def _load_map(name):
    """place_recognition_test.cpp:88-95 layout: 'label x y z' rows -> Vector7d with zero dimensions."""
    a = np.loadtxt(os.path.join(HERE, "golden", name))
    out = np.zeros((a.shape[0], 7))
    out[:, :4] = a[:, :4]
    return out


def _synthetic_pair(rng, n=40, extent=12.0, yaw=0.35, shift=(2.5, -1.75)):
    ref = np.zeros((n, 7))
    ref[:, 0] = rng.integers(1, 4, n)
    ref[:, 1:3] = rng.uniform(-extent, extent, (n, 2))
    ref[:, 3] = rng.normal(0, 0.2, n)
    ref[:, 4:7] = rng.uniform(0.3, 2.0, (n, 3))
    keep = rng.permutation(n)[: int(0.7 * n)]
    q = ref[keep].copy()
    c, s = np.cos(-yaw), np.sin(-yaw)
    xy = q[:, 1:3] - np.array(shift)
    q[:, 1] = c * xy[:, 0] - s * xy[:, 1]
    q[:, 2] = s * xy[:, 0] + c * xy[:, 1]
    q[:, 1:3] += rng.normal(0, 0.05, q[:, 1:3].shape)
    return ref, q


@pytest.mark.parametrize("case", ["synthetic", "synthetic_dims", "indoor"])
def test_match_maps_identical_to_oracle(gpu, case):
    rng = np.random.default_rng(7)
    if case == "indoor":
        ref, qry = _load_map("robot0Map_indoor.txt"), _load_map("robot1Map_indoor.txt")
        kw = dict(ignore_dimension=1, search_yaw_step_size=np.deg2rad(5.0), search_xy_step_size=0.5)
    else:
        ref, qry = _synthetic_pair(rng)
        kw = dict(ignore_dimension=int(case == "synthetic"), search_yaw_step_size=np.deg2rad(5.0), search_xy_step_size=0.5)
    for m in (ref, qry):                      # MatchMaps operates on centred maps (findTransformation :752-765)
        m[:, 1:3] -= m[:, 1:3].mean(axis=0)
    gp, op = _both_params(gpu, **kw)
    g = gpu.match_maps(ref, qry, gp)
    best = np.zeros(3)
    pr, pq = np.full(len(qry), -1, np.int32), np.full(len(qry), -1, np.int32)
    inl = po.lib().orc_match_maps(_p(np.ascontiguousarray(ref)), C.c_int(len(ref)), _p(np.ascontiguousarray(qry)),
                                  C.c_int(len(qry)), C.byref(op), _p(best), _p(pr), _p(pq))
    assert g["inliers"] == inl                       # integer work: identical
    assert np.array_equal(g["xyyaw"], best)          # the winning lattice point, bit for bit
    assert np.array_equal(g["ref_idx"], pr[:inl]) and np.array_equal(g["qry_idx"], pq[:inl])
    assert g["candidates"] > 1000


def test_find_inter_loop_closure_recovers_transform(gpu):
    rng = np.random.default_rng(11)
    ref, qry = _synthetic_pair(rng, n=60, yaw=0.35, shift=(2.5, -1.75))
    gp, op = _both_params(gpu, search_yaw_step_size=np.deg2rad(5.0), ignore_dimension=1)
    g = gpu.find_inter_loop_closure(ref, qry, gp)
    tf = np.zeros(16); inl = C.c_int(0); xyz = np.zeros(4)
    ok = po.lib().orc_find_transformation(_p(np.ascontiguousarray(ref)), C.c_int(len(ref)), _p(np.ascontiguousarray(qry)),
                                          C.c_int(len(qry)), C.byref(op), _p(tf), C.byref(inl), _p(xyz))
    assert g["found"] and ok == 1
    assert g["inliers"] == inl.value
    assert np.allclose(g["xyzyaw"], xyz, atol=1e-9)          # Horn (GPU side) vs Kabsch-SVD (oracle): same rotation
    assert abs(g["xyzyaw"][3] - 0.35) < np.deg2rad(3) and np.allclose(g["xyzyaw"][:2], [2.5, -1.75], atol=0.5)


def test_clipper_affinity_matches_golden_and_oracle(gpu):
    import sys
    sys.path.insert(0, HERE)
    from test_oracle_pins import MTRUE, _model_data
    model, data = _model_data()
    A = np.array([(i, j) for i in range(4) for j in range(3)], np.int32)
    M = gpu.clipper_affinity(model, data, A)
    assert np.array_equal(M + M.T + np.eye(12), MTRUE)       # the reference's own golden matrix, exact
    rng = np.random.default_rng(5)
    D1 = rng.uniform(-10, 10, (30, 2)); D2 = D1[rng.permutation(30)[:20]] + rng.normal(0, 0.03, (20, 2))
    A = np.array([(i, j) for i in range(30) for j in range(20)], np.int32)[::3].copy()
    m = len(A)
    Mg = gpu.clipper_affinity(D1, D2, A, sigma=0.1, epsilon=0.3)
    Ao = A.copy(); Mo = np.zeros((m, m))
    prm = (C.c_double * 12)()
    class OP(C.Structure):
        _fields_ = [("tol_u", C.c_double), ("tol_F", C.c_double), ("maxiniters", C.c_int), ("maxoliters", C.c_int),
                    ("beta", C.c_double), ("maxlsiters", C.c_int), ("eps", C.c_double), ("affinityeps", C.c_double),
                    ("rescale_u0", C.c_int), ("sigma", C.c_double), ("epsilon", C.c_double), ("mindist", C.c_double)]
    opp = OP(); po.lib().orc_clipper_default_params(C.byref(opp)); opp.sigma = 0.1; opp.epsilon = 0.3
    po.lib().orc_clipper_affinity(_p(np.ascontiguousarray(D1)), C.c_int(30), _p(np.ascontiguousarray(D2)), C.c_int(20), C.c_int(2),
                                  _p(Ao), C.c_int(m), C.byref(opp), _p(Mo))
    assert np.array_equal(Mg != 0, Mo != 0)                  # sparsity pattern (threshold decisions): identical
    assert np.allclose(Mg, Mo, rtol=1e-14, atol=0)           # exp() is device libm vs glibc: <= 1 ulp
