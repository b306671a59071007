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


def _quat_z(yaw):
    return np.array([0, 0, np.sin(yaw / 2), np.cos(yaw / 2)])


@pytest.mark.parametrize("use_lsq", [1, 0])
def test_find_intra_loop_closure_matches_oracle(gpu, use_lsq):
    """PlaceRecognition::findIntraLoopClosure (place_recognition.cpp:389-496): a submap around an old key pose, the same objects
    detected again from a query pose that has DRIFTED by (0.8, -0.6) m and 4 deg — product vs oracle (identical inliers, transform to
    1e-9), and the recovered correction undoes the drift."""
    rng = np.random.default_rng(5)
    n = 45
    submap = np.zeros((n, 7))
    submap[:, 0] = rng.integers(1, 4, n)
    submap[:, 1:3] = rng.uniform(-12, 12, (n, 2)) + np.array([30.0, 10.0])
    submap[:, 3] = rng.normal(0, 0.2, n)
    submap[:, 4:7] = rng.uniform(0.3, 2.0, (n, 3))
    true_q = np.concatenate([[31.0, 9.0, 1.0], _quat_z(0.6)])
    drift_q = np.concatenate([[31.8, 8.4, 1.0], _quat_z(0.6 + np.deg2rad(4.0))])
    cand = np.concatenate([[29.0, 11.0, 1.0], _quat_z(-0.2)])
    # detections in the TRUE local frame of the query pose
    c, s_ = np.cos(0.6), np.sin(0.6)
    Rq = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]])
    seen = rng.permutation(n)[:30]
    meas = submap[seen].copy()
    meas[:, 1:4] = (submap[seen, 1:4] - true_q[:3]) @ Rq + rng.normal(0, 0.03, (30, 3))
    gp, op = _both_params(gpu, search_yaw_step_size=np.deg2rad(1.0), ignore_dimension=1, use_nonlinear_least_squares=use_lsq,
                          search_xy_step_size=0.25)
    g = gpu.find_intra_loop_closure(meas, submap, drift_q, cand, gp)
    tf = np.zeros(16); inl = C.c_int(0); xyz = np.zeros(4)
    ok = po.lib().orc_find_intra_loop_closure(_p(np.ascontiguousarray(meas)), C.c_int(len(meas)), _p(np.ascontiguousarray(submap)),
                                              C.c_int(n), _p(drift_q), _p(cand), C.byref(op), C.c_double(5.0), C.c_double(5.0),
                                              C.c_double(np.deg2rad(10.0)), _p(tf), C.byref(inl), _p(xyz))
    assert g["found"] and ok == 1
    assert g["inliers"] == inl.value >= 20
    assert np.allclose(g["xyzyaw"], xyz, atol=1e-9) and np.allclose(g["tf"], tf.reshape(4, 4), atol=1e-9)
    # guards of :396-405: empty inputs and fewer than four detections are "not found", not errors
    assert not gpu.find_intra_loop_closure(meas[:3], submap, drift_q, cand, gp)["found"]
    assert not gpu.find_intra_loop_closure(meas[:0], submap, drift_q, cand, gp)["found"]
    assert not gpu.find_intra_loop_closure(meas, submap[:0], drift_q, cand, gp)["found"]
    if use_lsq:
        # the correction maps drifted-frame map coordinates onto the map: yaw ~ -4 deg
        assert abs(g["xyzyaw"][3] + np.deg2rad(4.0)) < np.deg2rad(1.0)


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


# ---- SlideGraph: triangle matching, CLIPPER dense clique, semantic_clipper pipeline -----------------
def _oracle_clipper_params(**kw):
    import ctypes as C

    class OCP(C.Structure):
        _fields_ = [("tol_u", C.c_double), ("tol_F", C.c_double), ("maxiniters", C.c_int), ("maxoliters", C.c_int),
                    ("beta", C.c_double), ("maxlsiters", C.c_int), ("eps", C.c_double), ("affinityeps", C.c_double),
                    ("rescale_u0", C.c_int), ("sigma", C.c_double), ("epsilon", C.c_double), ("mindist", C.c_double)]
    p = OCP()
    po.lib().orc_clipper_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _triangles(points):
    from scipy.spatial import Delaunay
    tri = Delaunay(points, qhull_options="Qt Qbb Qc Qz Q12")       # observation.cpp:25-26 options
    return points[tri.simplices].astype(np.float64)                 # (n, 3, 2)


def _slidegraph_case(seed, n=40, noise=0.01, n_query=25):
    rng = np.random.default_rng(seed)
    ref = rng.uniform(-30, 30, (n, 2))
    yaw = rng.uniform(-np.pi, np.pi)
    R = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
    t = rng.uniform(-5, 5, 2)
    sel = rng.permutation(n)[:n_query]
    qry = (ref[sel] - t) @ R + rng.normal(0, noise, (n_query, 2))   # ref = R qry + t
    return _triangles(ref), _triangles(qry), R, t


@pytest.mark.gpu
def test_match_triangles_identical_to_oracle(gpu):
    import ctypes as C
    for seed in range(3):
        tm, td, _, _ = _slidegraph_case(seed)
        pts, diffs = gpu.match_triangles(tm, td, 0.1)
        cap = len(tm) * len(td)
        op = np.zeros((cap, 3, 4)); od = np.zeros(cap)
        tmf, tdf = np.ascontiguousarray(tm.reshape(-1, 6)), np.ascontiguousarray(td.reshape(-1, 6))
        n = po.lib().orc_match_triangles(tmf.ctypes.data_as(C.c_void_p), C.c_int(len(tm)), tdf.ctypes.data_as(C.c_void_p),
                                         C.c_int(len(td)), C.c_double(0.1), op.ctypes.data_as(C.c_void_p),
                                         od.ctypes.data_as(C.c_void_p), C.c_int(cap))
        assert n == len(diffs) and n > 0
        assert np.array_equal(pts, op[:n])            # same pairs, same order, same vertex order: bit-identical rows
        assert np.array_equal(diffs, od[:n])


@pytest.mark.gpu
def test_dense_clique_matches_oracle(gpu):
    import ctypes as C
    rng = np.random.default_rng(11)
    D1 = rng.uniform(-10, 10, (40, 2))
    perm = rng.permutation(40)[:25]
    D2 = D1[perm] + rng.normal(0, 0.02, (25, 2))
    A = np.array([(i, j) for i in range(40) for j in range(25)], np.int32)[::4].copy()
    for k, j in enumerate(perm[:12]):                  # make sure true pairs are among the putative ones
        A[k] = (j, k)
    M = gpu.clipper_affinity(D1, D2, A, sigma=0.1, epsilon=0.3)
    m = len(A)
    p = gpu.clipper_params(sigma=0.1, epsilon=0.3)
    op = _oracle_clipper_params(sigma=0.1, epsilon=0.3)
    agree = 0
    for seed in range(6):
        u0 = np.random.default_rng(seed).uniform(0, 1, m)
        nodes, u, score = gpu.clipper_dense_clique(M, u0, p)
        on = np.zeros(m, np.int32); ou = np.zeros(m); osc = C.c_double(0)
        n = po.lib().orc_clipper_solve(M.ctypes.data_as(C.c_void_p), C.c_int(m), u0.ctypes.data_as(C.c_void_p), C.byref(op),
                                       on.ctypes.data_as(C.c_void_p), ou.ctypes.data_as(C.c_void_p), C.byref(osc))
        # a local solver on a floating-point path: same clique and score as the oracle from the same start
        assert sorted(nodes.tolist()) == sorted(on[:n].tolist())
        assert abs(score - osc.value) < 1e-6 * max(1.0, abs(osc.value))
        assert np.abs(u - ou).max() < 1e-6
        agree += 1
    assert agree == 6
    nodes, _, _ = gpu.clipper_dense_clique(M, None, p)  # library-drawn start weights
    assert len(nodes) >= 3


@pytest.mark.gpu
def test_dense_clique_batch_equals_single_solves(gpu):
    """slide_clipper_dense_clique_batch: the 28 robot-pair problems of an eight-robot job as ONE launch, a persistent
    workgroup per problem — every problem's clique, weights and score equal the single-problem call's (the same device code per job),
    problems of different sizes, an empty one among them."""
    rng = np.random.default_rng(5)
    p = gpu.clipper_params(sigma=0.1, epsilon=0.3)
    Ms, u0s = [], []
    for j in range(28):
        n1 = int(rng.integers(12, 40))
        D1 = rng.uniform(-10, 10, (n1, 2))
        k = int(rng.integers(6, n1))
        perm = rng.permutation(n1)[:k]
        D2 = D1[perm] + rng.normal(0, 0.02, (k, 2))
        A = np.array([(i, jj) for i in range(n1) for jj in range(k)], np.int32)[::3].copy()
        for q, i in enumerate(perm[:min(6, k)]):
            A[q] = (i, q)
        Ms.append(gpu.clipper_affinity(D1, D2, A, sigma=0.1, epsilon=0.3))
        u0s.append(rng.uniform(0, 1, len(A)))
    Ms.insert(7, np.zeros((0, 0))); u0s.insert(7, None)      # an empty problem
    res = gpu.clipper_dense_clique_batch(Ms, u0s, p)
    assert len(res) == 29 and len(res[7][0]) == 0
    for j, (M, u0) in enumerate(zip(Ms, u0s)):
        if M.shape[0] == 0:
            continue
        nodes, u, score = gpu.clipper_dense_clique(M, u0, p)
        bn, bu, bs = res[j]
        assert sorted(bn.tolist()) == sorted(nodes.tolist()), j
        assert np.array_equal(bu, u) and bs == score, j


@pytest.mark.gpu
def test_dense_clique_large_planted(gpu):
    """m = 4000 putative associations (a 128 MB dense affinity matrix on the host, ~1 % of it non-zero): a planted set of 60 mutually
    consistent associations among random pairwise-consistent noise — the device-resident solve (CSR + one persistent workgroup)
    returns the planted clique (>= 90 % of it: projected gradient ascent is a local solver); score ~ its size (DSD_HEU: omega = round(F))."""
    rng = np.random.default_rng(5)
    m, k = 4000, 60
    M = np.zeros((m, m))
    iu = np.triu_indices(m, 1)
    mask = rng.uniform(0, 1, len(iu[0])) < 0.01
    M[iu[0][mask], iu[1][mask]] = rng.uniform(0.2, 0.9, int(mask.sum()))
    planted = np.sort(rng.permutation(m)[:k])
    for a in range(k):
        for b in range(a + 1, k):
            M[planted[a], planted[b]] = rng.uniform(0.9, 1.0)
    nodes, u, score = gpu.clipper_dense_clique(M, rng.uniform(0, 1, m), gpu.clipper_params())
    # (a local solver: it may trade a few planted members for noise neighbours)
    hit = len(set(nodes.tolist()) & set(planted.tolist()))
    assert hit >= 0.9 * k and abs(len(nodes) - k) <= 0.1 * k, (hit, len(nodes))
    assert abs(score - k) < 0.15 * k
    assert np.all(u >= 0) and abs(np.linalg.norm(u) - 1.0) < 1e-9


def _planted_problem(m, k, density, seed):
    rng = np.random.default_rng(seed)
    M = np.zeros((m, m))
    iu = np.triu_indices(m, 1)
    mask = rng.uniform(0, 1, len(iu[0])) < density
    M[iu[0][mask], iu[1][mask]] = rng.uniform(0.2, 0.9, int(mask.sum()))
    planted = np.sort(rng.permutation(m)[:k])
    for a in range(k):
        for b in range(a + 1, k):
            M[planted[a], planted[b]] = rng.uniform(0.9, 1.0)
    return M, planted, rng.uniform(0, 1, m)


@pytest.mark.gpu
@pytest.mark.parametrize("m,k,density", [(1500, 40, 0.02), (4096, 60, 0.01), (6000, 80, 0.004)])
def test_dense_clique_large_cooperative_equals_one_workgroup(gpu, m, k, density):
    """One LARGE problem (m >= 1024): the sparse product's rows over the waves of many co-resident workgroups (k_clq_solve_coop, one
    grid barrier per gradient evaluation) — iterate for iterate the one-workgroup solve: the same number of gradient evaluations,
    bit-identical weights and score, the same clique; and the clique, weights and score of the oracle from the same start
    (the tolerances of test_dense_clique_matches_oracle).  Both wall times go into the test's output."""
    import ctypes as C
    import os
    import time
    M, planted, u0 = _planted_problem(m, k, density, 17 + m)
    p = gpu.clipper_params()
    res = {}
    for wgs in ("1", None, "37"):          # one workgroup; the default count (m / 64, at most 128); an odd count
        if wgs is None:
            os.environ.pop("SLIDE_CLIPPER_WGS", None)
        else:
            os.environ["SLIDE_CLIPPER_WGS"] = wgs
        try:
            gpu.clipper_dense_clique(M, u0, p)                 # warm (allocations, code load)
            t0 = time.perf_counter()
            nodes, u, score = gpu.clipper_dense_clique(M, u0, p)
            dt = time.perf_counter() - t0
        finally:
            os.environ.pop("SLIDE_CLIPPER_WGS", None)
        res[wgs] = (nodes, u, score, gpu.clipper_last_solve_info(), dt)
    n1, u1, s1, (w1, e1), t1 = res["1"]
    assert w1 == 1
    for key in (None, "37"):
        nc, uc, sc, (wc, ec), tc = res[key]
        assert wc == (37 if key else min(128, (m + 63) // 64)) and wc > 1
        assert ec == e1 and sc == s1 and np.array_equal(uc, u1) and np.array_equal(nc, n1), (key, ec, e1, sc - s1)
    print(f"clipper m={m}: {int(e1)} gradient evaluations; whole call (upload + CSR + solve) {t1 * 1e3:.1f} ms on one workgroup, "
          f"{res[None][4] * 1e3:.1f} ms on {res[None][3][0]}")
    hit = len(set(n1.tolist()) & set(planted.tolist()))
    assert hit >= 0.9 * k, (hit, len(n1))
    op = _oracle_clipper_params()
    on = np.zeros(m, np.int32); ou = np.zeros(m); osc = C.c_double(0)
    n = po.lib().orc_clipper_solve(M.ctypes.data_as(C.c_void_p), C.c_int(m), u0.ctypes.data_as(C.c_void_p), C.byref(op),
                                   on.ctypes.data_as(C.c_void_p), ou.ctypes.data_as(C.c_void_p), C.byref(osc))
    assert sorted(n1.tolist()) == sorted(on[:n].tolist())
    assert abs(s1 - osc.value) < 1e-6 * max(1.0, abs(osc.value))
    assert np.abs(u1 - ou).max() < 1e-6


@pytest.mark.gpu
def test_semantic_clipper_pipeline(gpu):
    import ctypes as C
    found = 0
    for seed in range(4):
        tm, td, R, t = _slidegraph_case(100 + seed)
        pts, _ = gpu.match_triangles(tm, td, 0.1)
        m = 3 * len(pts)
        u0 = np.random.default_rng(seed).uniform(0, 1, m)
        p = gpu.clipper_params(sigma=0.05, epsilon=0.15)
        r = gpu.semantic_clipper(tm, td, p, min_num_pairs=4, matching_threshold=0.1, u0=u0)
        op = _oracle_clipper_params(sigma=0.05, epsilon=0.15)
        tf = np.zeros(16); counts = np.zeros(2, np.int32); inl = np.zeros(max(m, 1), np.int32)
        tmf, tdf = np.ascontiguousarray(tm.reshape(-1, 6)), np.ascontiguousarray(td.reshape(-1, 6))
        ok = po.lib().orc_semantic_clipper(tmf.ctypes.data_as(C.c_void_p), C.c_int(len(tm)), tdf.ctypes.data_as(C.c_void_p), C.c_int(len(td)),
                                           C.byref(op), C.c_int(4), C.c_double(0.1), u0.ctypes.data_as(C.c_void_p),
                                           tf.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), inl.ctypes.data_as(C.c_void_p))
        assert r["n_putative"] == counts[0] == m
        assert r["found"] == bool(ok)
        # the putative list repeats a vertex pair once per matched triangle pair that contains it; such duplicates carry
        # equal weights up to rounding, so WHICH copy makes the top-omega cut is not defined: compare counts, the selected
        # point pairs as a set, and the transform
        assert r["n_inliers"] == counts[1]
        rows = pts.reshape(-1, 4)
        sel_g = {tuple(rows[i]) for i in r["inliers"]}
        sel_o = {tuple(rows[i]) for i in inl[:counts[1]]}
        assert len(sel_g ^ sel_o) <= 2
        assert np.abs(r["tf"].ravel() - tf).max() < 1e-3
        if r["found"]:
            found += 1
            # the recovered transform maps the query frame onto the reference frame (semantic_clipper's "model" = reference):
            # estimate_tf(model, data) maps model -> data, i.e. reference -> query = (R, t)^-1
            Rq = r["tf"][:2, :2]; tq = r["tf"][:2, 3]
            assert np.abs(Rq - R.T).max() < 0.02 and np.abs(tq - (-R.T @ t)).max() < 0.3
    assert found >= 3


@pytest.mark.gpu
def test_estimate_tf2d(gpu):
    rng = np.random.default_rng(3)
    a = rng.uniform(-5, 5, (12, 2))
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    b = a @ R.T + np.array([1.5, -2.0])
    tf = gpu.estimate_tf2d(a, b)
    assert np.abs(tf[:2, :2] - R).max() < 1e-12 and np.abs(tf[:2, 2] - [1.5, -2.0]).max() < 1e-12


@pytest.mark.gpu
def test_run_semantic_clipper_from_maps(gpu):
    """The whole SlideGraph call (semantic_clipper.cpp:140-274) from two object maps: own Delaunay + GPU matching + CLIPPER."""
    found = 0
    for seed in range(4):
        rng = np.random.default_rng(200 + seed)
        n, nq = 45, 28
        ref = np.zeros((n, 7)); ref[:, 0] = 1; ref[:, 1:3] = rng.uniform(-30, 30, (n, 2))
        yaw = rng.uniform(-np.pi, np.pi)
        R = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
        t = rng.uniform(-5, 5, 2)
        sel = rng.permutation(n)[:nq]
        qry = np.zeros((nq, 7)); qry[:, 0] = 1
        qry[:, 1:3] = (ref[sel, 1:3] - t) @ R + rng.normal(0, 0.01, (nq, 2))
        r = gpu.run_semantic_clipper(ref, qry, sigma=0.05, epsilon=0.15, min_num_pairs=4, matching_threshold=0.1)
        if r["found"]:
            found += 1
            assert np.abs(r["tf"][:2, :2] - R.T).max() < 0.02 and np.abs(r["tf"][:2, 3] - (-R.T @ t)).max() < 0.3
    assert found >= 3
