"""Pins the oracle (CPU restatement) against every known-answer test the reference tree holds for this
path.  The cases are transcribed as DATA from the reference's test files:
  clipper_semantic_object/test/affinity_test.cpp:33-107, clipper_test.cpp:14-67,
  backend/sloam/src/test/sloam_test.cpp:20-205, utils_test.cpp:4-25,
  src/test/deprecated/cube_factor_test.cpp:156-260 (non-building upstream; identities, the Retract vector and the
  FactorGraph scenario transcribed by hand).
The GTSAM boundary itself (ISAM2, BetweenFactor, BearingRangeFactor, numericalDerivative) is pinned by NO
reference test -> 'parity unpinned' there (oracle headers)."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _model_data():
    model = np.array([[0, 0, 0], [2, 0, 0], [0, 3, 0], [2, 2, 0]], dtype=np.float64)
    th = np.pi / 8
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    t = np.array([5.0, 3.0, 0.0])
    data = (R.T @ (model - t).T).T           # T_MD^-1 * model
    return model, np.ascontiguousarray(data[:3])


MTRUE = np.array([
    [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], [0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0],
    [0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0], [1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0], [0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0],
    [0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0], [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0],
    [0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0], [0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1]],
    dtype=np.float64)


def _affinity(model, data):
    L = po.lib()
    A = np.zeros((12, 2), np.int32)
    M = np.zeros((12, 12))
    m = L.orc_clipper_affinity(_p(model), C.c_int(4), _p(data), C.c_int(3), C.c_int(3), _p(A), C.c_int(0), None, _p(M))
    return m, A, M


def test_clipper_affinity_golden():
    """affinity_test.cpp: all-to-all association order and the exact 12x12 binary affinity 'from MATLAB'."""
    model, data = _model_data()
    m, A, M = _affinity(model, data)
    assert m == 12
    for i in range(4):
        for j in range(3):
            assert tuple(A[i * 3 + j]) == (i, j)
    full = M + M.T + np.eye(12)            # getAffinityMatrix(): symmetric view + identity
    assert np.array_equal(full, MTRUE)


def _clique(M, seed):
    u0 = np.random.default_rng(seed).uniform(0, 1, 12)
    nodes = np.zeros(12, np.int32)
    n = po.lib().orc_clipper_solve(_p(M), C.c_int(12), _p(u0), None, _p(nodes), None, None)
    return n, nodes[:n]


def test_clipper_dense_clique_golden():
    """clipper_test.cpp:14-67: exactly 3 inlier associations, each (i, i).  The reference draws u0 from a
    std::random_device-seeded generator; projected gradient ascent is a LOCAL solver, and an independent numpy
    transcription of clipper.cpp:172-323 confirms that ~10 % of uniform u0 converge to a 2-clique instead
    (seeds 1, 25, 29 below) — the upstream test is statistically, not deterministically, green.  Pinned here:
    the 3-association answer for the seeds that reach the global optimum, and the success rate."""
    model, data = _model_data()
    m, A, M = _affinity(model, data)
    good = 0
    for seed in range(30):
        n, nodes = _clique(M, seed)
        if n == 3 and all(A[k, 0] == A[k, 1] for k in nodes):
            good += 1
        else:
            assert n == 2           # the only other local optimum of this 12-node graph
    assert good >= 26
    for seed in (0, 2, 3, 4, 5):
        n, nodes = _clique(M, seed)
        assert n == 3 and sorted(A[nodes, 0]) == [0, 1, 2]


def _closest(stamps, q):
    sec = np.array([s for s, _ in stamps], np.int64)
    ns = np.array([n for _, n in stamps], np.int64)
    idx, diff = C.c_int(0), C.c_double(0)
    po.lib().orc_closest_stamp(_p(sec), _p(ns), C.c_int(len(stamps)), C.c_int64(q[0]), C.c_int64(q[1]), C.byref(idx), C.byref(diff))
    return idx.value, diff.value


def test_get_index_closest_pose_mst_pair():
    """sloam_test.cpp:20-57."""
    idx, diff = _closest([], (10, 0))
    assert idx == -1 and diff == np.finfo(np.float64).max
    assert _closest([(5, 0)], (10, 0)) == (0, 5.0)
    pk = [(5, 0), (15, 0), (12, 0)]
    assert _closest(pk, (11, 0)) == (2, 1.0)
    assert _closest(pk, (13, 500000000)) == (1, 1.5)        # tie -> first occurrence
    i, d = _closest(pk, (13, 400000000))
    assert i == 2 and d == pytest.approx(1.4, abs=1e-12)


def _find(packets, counters, host, pending):
    n_r = len(packets)
    sec, ns, off = [], [], [0]
    for pk in packets:
        for (s, n) in pk:
            sec.append(s); ns.append(n)
        off.append(len(sec))
    sec = np.array(sec + [0], np.int64); ns = np.array(ns + [0], np.int64); off = np.array(off, np.int32)
    pc = np.array(counters, np.uint64)
    npend = C.c_int(len(pending))
    m_sec = np.array([p[0][0] for p in pending] + [0], np.int64)
    m_ns = np.array([p[0][1] for p in pending] + [0], np.int64)
    m_rob = np.array([p[1] for p in pending] + [0], np.int32)
    m_odo = np.array([int(p[2]) for p in pending] + [0], np.int32)
    m_tag = np.arange(len(pending) + 1, dtype=np.int32)
    out = np.zeros(4 * max(len(pending), 1), np.int32)
    n = po.lib().orc_find_relmeas(C.c_int(n_r), _p(sec), _p(ns), _p(off), _p(pc), C.c_int(host), C.byref(npend), _p(m_sec),
                                  _p(m_ns), _p(m_rob), _p(m_odo), _p(m_tag), _p(out))
    return n, out.reshape(-1, 4)[: max(n, 0)], npend.value


def test_find_relative_measurement_match():
    """sloam_test.cpp:59-205 (every scenario, in order)."""
    assert _find([[], []], [0, 0], 0, [])[0] == 0
    assert _find([[], []], [0, 0], 0, [((0, 0), 0, False)])[0] == -1                 # robotIndex == host -> throws
    assert _find([[], []], [0, 0], 0, [((0, 0), 1, True)])[0] == -1                  # onlyUseOdom -> throws
    n, m, left = _find([[], []], [0, 0], 0, [((5, 0), 1, False)])
    assert n == 0
    n, m, left = _find([[(5, 0)], [(5, 0)]], [1, 1], 0, [((5, 0), 1, False)])
    assert n == 1 and tuple(m[0][2:]) == (0, 0) and left == 0
    n, m, left = _find([[(5, 0), (7, 0)], [(5, 0), (7, 0)]], [2, 2], 0, [((5, 0), 1, False), ((7, 1000), 1, False)])
    assert n == 2 and tuple(m[0][2:]) == (0, 0) and tuple(m[1][2:]) == (1, 1) and left == 0
    n, m, left = _find([[(5, 0), (7, 0), (9, 8000000)], [(5, 0), (7, 0), (10, 2000000)]], [3, 3], 0, [((10, 0), 1, False)])
    assert n == 0 and left == 1                                                      # > 1 ms -> kept, not matched
    n, m, left = _find([[(4, 0)], [(4, 0)]], [1, 1], 0, [((2, 0), 1, False)])
    assert n == 0 and left == 0                                                      # stale -> pruned


def test_se3_to_pose3_conversion():
    """utils_test.cpp:4-25: Rz(pi/4), t = (1,2,3) survives the pose7 <-> (R,t) conversions to 1e-15."""
    th = np.pi / 4
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    p12 = np.concatenate([R.ravel(), [1.0, 2.0, 3.0]])
    p7 = np.zeros(7); back = np.zeros(12)
    po.lib().orc_pose12_to7(_p(p12), _p(p7))
    po.lib().orc_pose7_to12(_p(p7), _p(back))
    assert np.allclose(back[:9], R.ravel(), atol=1e-15, rtol=0)
    assert np.allclose(back[9:], [1, 2, 3], atol=1e-15, rtol=0)


def test_cube_measurement_identities():
    """deprecated/cube_factor_test.cpp:156-227 (transcribed): Retract/LocalCoordinates statics and the
    chart-vs-Expmap closeness the reference test quantifies (0.01 in Logmap coordinates)."""
    L = po.lib()
    v = np.array([0, 0, 0, 0, 0, 0, 0.1, 0.1, 0.1])
    c15 = np.zeros(15)
    L.orc_cube_Retract_static(_p(v), _p(c15))
    assert np.allclose(c15[:9], np.eye(3).ravel()) and np.allclose(c15[9:12], 0) and np.allclose(c15[12:], 0.1)
    lc = np.zeros(9)
    L.orc_cube_LocalCoordinates_static(_p(c15), _p(lc))
    assert np.allclose(lc, v, atol=1e-15)
    # localCoordinates: scale part = m.scale - q.scale = +0.1
    q15 = c15.copy(); q15[12:] = 0.0
    L.orc_cube_localCoordinates(_p(c15), _p(q15), _p(lc))
    assert np.allclose(lc[6:], 0.1) and np.allclose(lc[:6], 0, atol=1e-15)
    # retract(v) with the EXPMAP chart == pose * Expmap(v[0:6]) exactly; the test's tolerance is 0.01
    xi0 = np.array([0.5, 0.8, 0.3, 100, 12, 10.5])
    pose = np.zeros(12); L.orc_pose_expmap(_p(xi0), _p(pose))
    m15 = np.concatenate([pose, [0.0, 0.0, 0.0]])
    vv = np.array([1.8, 0.2, 1.0, 30, 100, 2.5, 1, 1, 1.0]) * 1e-3   # small step: both charts agree to 0.01
    out = np.zeros(15)
    L.orc_cube_retract(_p(m15), _p(vv), C.c_int(po.CHART_EXPMAP), _p(out))
    ex = np.zeros(12); L.orc_pose_expmap(_p(np.ascontiguousarray(vv[:6])), _p(ex))
    comp = np.zeros(12); L.orc_pose_compose(_p(pose), _p(ex), _p(comp))
    assert np.allclose(out[:12], comp, atol=1e-14)
    out_c = np.zeros(15)
    L.orc_cube_retract(_p(m15), _p(vv), C.c_int(po.CHART_CAYLEY), _p(out_c))
    la, lb = np.zeros(6), np.zeros(6)
    L.orc_pose_logmap(_p(out[:12].copy()), _p(la)); L.orc_pose_logmap(_p(out_c[:12].copy()), _p(lb))
    assert np.abs(la - lb).max() < 0.01
    assert np.allclose(out_c[12:], 1e-3)


def retract_vector_unscaled(chart):
    """deprecated/cube_factor_test.cpp:201-227 AS WRITTEN: cube pose = Expmap(0.5, 0.8, 0.3, 100, 12, 10.5), scale (2, 3.5, 0.5);
    v = (1.8, 0.2, 1.0, 30, 100, 2.5, 1, 1, 1); expectation Logmap(retract(v).pose) == Logmap(pose * Expmap(v[0:6])) within
    0.01 per coordinate and scale + 1 within 0.01.  Returns (max |Logmap difference|, max |scale difference|)."""
    L = po.lib()
    xi0 = np.array([0.5, 0.8, 0.3, 100.0, 12.0, 10.5])
    pose = np.zeros(12); L.orc_pose_expmap(_p(xi0), _p(pose))
    m15 = np.concatenate([pose, [2.0, 3.5, 0.5]])
    v = np.array([1.8, 0.2, 1.0, 30.0, 100.0, 2.5, 1.0, 1.0, 1.0])
    out = np.zeros(15)
    L.orc_cube_retract(_p(m15), _p(v), C.c_int(chart), _p(out))
    ex = np.zeros(12); L.orc_pose_expmap(_p(np.ascontiguousarray(v[:6])), _p(ex))
    comp = np.zeros(12); L.orc_pose_compose(_p(pose), _p(ex), _p(comp))
    la, lb = np.zeros(6), np.zeros(6)
    L.orc_pose_logmap(_p(comp), _p(la)); L.orc_pose_logmap(_p(out[:12].copy()), _p(lb))
    return float(np.abs(la - lb).max()), float(np.abs(out[12:] - (m15[12:] + 1.0)).max())


def test_cube_retract_reference_vector_unscaled_both_charts():
    """The only chart evidence the reference tree holds (cube_factor_test.cpp:201-227, |omega| ~ 2 rad, tolerance 0.01), run
    UNSCALED under both charts.  Recorded outcome: the Expmap chart passes (it is the identity the test writes
    down), the Cayley chart — the default here, named by the reference's own comment cubeFactor.h:96-97 — misses by far more than
    0.01.  That test is under src/test/deprecated/ and is not built (CMakeLists.txt), so it cannot decide which chart the shipped
    GTSAM build uses; both stay selectable (SLIDE_CHART_*)."""
    d_exp, s_exp = retract_vector_unscaled(po.CHART_EXPMAP)
    d_cay, s_cay = retract_vector_unscaled(po.CHART_CAYLEY)
    assert d_exp < 1e-9 and s_exp < 1e-12                  # passes the reference's 0.01 with all the margin there is
    assert d_cay > 0.01 and s_cay < 1e-12                  # would FAIL the reference's expectation: recorded, not hidden


def cube_factor_graph_scenario(G, chart=0):
    """deprecated/cube_factor_test.cpp:229-260 (TEST_F FactorGraph), transcribed: prior on pose A; two addKeyPoseAndBetween calls
    whose relativeMotion arguments are poseA / poseB and whose estimates are poseB / poseC (as written there); one cube at
    Expmap(0.5, 0.8, 0.3, 100, 12, 10.5) observed from the three poses with scales 0.1 / 0.15 / 0.05; solve() after the second and
    after the third observation.  G: any object with the SemanticFactorGraph seam (oracle or product).  Returns the optimised
    scale after each solve."""
    L = po.lib()
    xi0 = np.array([0.5, 0.8, 0.3, 100.0, 12.0, 10.5])
    pose = np.zeros(12); L.orc_pose_expmap(_p(xi0), _p(pose))
    cube7 = np.zeros(7); L.orc_pose12_to7(_p(pose), _p(cube7))
    A = np.array([0, 0, 0, 0, 0, 0, 1.0]); B = A.copy(); B[0] = 0.1; Cc = A.copy(); Cc[0] = -0.1
    G.set_prior(0, A)
    G.add_keypose_between(0, 0, 1, A, B)
    G.add_keypose_between(0, 1, 2, B, Cc)
    G.add_cube(0, 0, 0, A, cube7, [0.1] * 3, False)
    G.add_cube(0, 1, 0, B, cube7, [0.15] * 3, True)
    out = []
    assert G.solve() == 0
    out.append(G.get_landmark(1, 0)[1][12:15].copy())
    G.add_cube(0, 2, 0, Cc, cube7, [0.05] * 3, True)
    assert G.solve() == 0
    out.append(G.get_landmark(1, 0)[1][12:15].copy())
    return out, pose[9:12].copy()


def cube_factor_graph_expected(cube_t):
    """The scale rows of a cube factor are linear and decoupled from every pose (cubeFactor.h:46-87: m.scale - q.scale), so the
    optimum is the mean of the measured scales weighted by 1 / sigma^2 with sigma = noise_model_cube_vec * max(|t_cube_local|, 0.1)
    (graph.cpp:213-218)."""
    d = np.array([np.linalg.norm(cube_t - np.array([x, 0, 0])) for x in (0.0, 0.1, -0.1)])
    w = 1.0 / np.maximum(d, 0.1) ** 2
    s = np.array([0.1, 0.15, 0.05])
    return (w[:2] * s[:2]).sum() / w[:2].sum(), (w * s).sum() / w.sum()


@pytest.mark.parametrize("chart", [po.CHART_CAYLEY, po.CHART_EXPMAP])
def test_cube_factor_graph_known_answer(chart):
    """The one reference-held end-to-end answer for cube factor + solve(): `getCube(0).scale[0] == 0.1 +- 1e-5`
    (cube_factor_test.cpp:260).  With the distance-proportional cube noise of the CURRENT graph.cpp:213-218 the three
    observations (|t_local| = 98.80 / 98.71 / 98.89 m) no longer weigh the same, and the exact optimum is the weighted mean
    0.1000626 — 6.3e-5 from 0.1: the deprecated test (not built, CMakeLists.txt) predates the noise shaping and its 1e-5 cannot
    hold against the shipped code.  Pinned here: the analytic optimum to 1e-9 and the reference's number to 1e-4."""
    G = po.OracleGraph(po.OrcParams.default(pose_chart=chart))
    scales, cube_t = cube_factor_graph_scenario(G, chart)
    e2, e3 = cube_factor_graph_expected(cube_t)
    assert np.allclose(scales[0], e2, atol=1e-9, rtol=0)
    assert np.allclose(scales[1], e3, atol=1e-9, rtol=0)
    assert abs(scales[1][0] - 0.1) < 1e-4
    assert abs(e3 - 0.1000625857) < 1e-9


def test_lie_identities():
    """Expmap o Logmap = id, Cayley o CayleyLocal = id, retract/local inverse pairs (both charts)."""
    L = po.lib()
    rng = np.random.default_rng(0)
    for _ in range(50):
        xi = rng.normal(0, 1, 6) * np.array([0.8, 0.8, 0.8, 5, 5, 5])
        T = np.zeros(12); L.orc_pose_expmap(_p(xi), _p(T))
        back = np.zeros(6); L.orc_pose_logmap(_p(T), _p(back))
        assert np.allclose(back, xi, atol=1e-10)
        w = rng.normal(0, 0.7, 3)
        R = np.zeros(9); L.orc_so3_cayley(_p(w), _p(R))
        assert np.allclose(R.reshape(3, 3) @ R.reshape(3, 3).T, np.eye(3), atol=1e-14)
        wb = np.zeros(3); L.orc_so3_cayley_local(_p(R), _p(wb))
        assert np.allclose(wb, w, atol=1e-12)
        for chart in (0, 1):
            Y = np.zeros(12); L.orc_pose_retract(_p(T), _p(xi * 0.3), C.c_int(chart), _p(Y))
            loc = np.zeros(6); L.orc_pose_local(_p(T), _p(Y), C.c_int(chart), _p(loc))
            assert np.allclose(loc, xi * 0.3, atol=1e-10)


def _lin(ftype, x0, x1type, x1, z, sigma, chart=0, whiten=0):
    r = np.zeros(9); J0 = np.zeros(81); J1 = np.zeros(81)
    m = po.lib().orc_linearize(C.c_int(ftype), _p(x0), C.c_int(x1type), _p(x1) if x1 is not None else None, _p(z), _p(sigma),
                               C.c_int(chart), C.c_double(1e-6), _p(r), _p(J0), _p(J1), C.c_int(whiten))
    d1 = {po.V_POSE: 6, po.V_POINT: 3, po.V_CUBE: 9, po.V_CYL: 7}[x1type]
    return r[:m], J0[: m * 6].reshape(m, 6), J1[: m * d1].reshape(m, d1)


def test_analytic_jacobians_match_central_differences():
    """Between / bearing-range analytic Jacobians (the forms GTSAM uses) vs numerical differentiation through
    the same chart; cylinder H2 is analytically [I6 0; 0 -1]."""
    L = po.lib()
    rng = np.random.default_rng(3)

    def rand_pose():
        T = np.zeros(12); L.orc_pose_expmap(_p(rng.normal(0, 1, 6) * np.array([.5, .5, .5, 4, 4, 4])), _p(T)); return T

    def retract(T, d, chart):
        o = np.zeros(12); L.orc_pose_retract(_p(T), _p(np.ascontiguousarray(d)), C.c_int(chart), _p(o)); return o
    sig = np.ones(9)
    for chart in (0, 1):
        x = rand_pose(); p = rng.normal(0, 5, 3)
        q = x[:9].reshape(3, 3).T @ (p - x[9:]); z = np.concatenate([q / np.linalg.norm(q) + rng.normal(0, 0.01, 3), [np.linalg.norm(q) + 0.1]])
        z[:3] /= np.linalg.norm(z[:3])
        r, J0, J1 = _lin(po.F_BR, x, po.V_POINT, p, z, sig, chart)
        eps = 1e-6
        for j in range(6):
            d = np.zeros(6); d[j] = eps
            rp, _, _ = _lin(po.F_BR, retract(x, d, chart), po.V_POINT, p, z, sig, chart)
            rm, _, _ = _lin(po.F_BR, retract(x, -d, chart), po.V_POINT, p, z, sig, chart)
            # the bearing rows live in the basis of the PREDICTED bearing in J but of the MEASURED bearing in r
            # (GTSAM's own inconsistency, restated); they agree to first order in the residual
            assert np.allclose((rp - rm)[2] / (2 * eps), J0[2, j], atol=1e-6)
            assert np.allclose((rp - rm)[:2] / (2 * eps), J0[:2, j], atol=5e-2)
        a, b, zz = rand_pose(), rand_pose(), rand_pose()
        r, J0, J1 = _lin(po.F_BETWEEN, a, po.V_POSE, b, zz, sig, chart)
        assert np.allclose(J1, np.eye(6))
    x = rand_pose()
    cyl = np.array([4, 5, 0.1, 0.01, -0.02, 1.0, 0.3]); zc = np.array([3, 3, -2.9, 0.01, 0, 1.0, 0.31])
    r, J0, J1 = _lin(po.F_CYL, x, po.V_CYL, cyl, zc, sig)
    expect = np.eye(7); expect[6, 6] = -1
    assert np.allclose(J1, expect, atol=1e-8)


# ---- SlideGraph restatement (oracle/slidegraph.hpp): no reference fixture exists, so these pin its own invariants -------------
def _tri_case(seed, n=30, nq=20):
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    ref = rng.uniform(-20, 20, (n, 2))
    yaw = rng.uniform(-np.pi, np.pi)
    R = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
    t = rng.uniform(-3, 3, 2)
    qry = (ref[rng.permutation(n)[:nq]] - t) @ R
    tri = lambda pts: pts[Delaunay(pts, qhull_options="Qt Qbb Qc Qz Q12").simplices].astype(np.float64)
    return tri(ref), tri(qry), R, t


def test_oracle_triangle_matching_and_tf():
    tm, td, R, t = _tri_case(2)
    tmf, tdf = np.ascontiguousarray(tm.reshape(-1, 6)), np.ascontiguousarray(td.reshape(-1, 6))
    cap = len(tm) * len(td)
    pts = np.zeros((cap, 3, 4)); diffs = np.zeros(cap)
    n = po.lib().orc_match_triangles(_p(tmf), C.c_int(len(tm)), _p(tdf), C.c_int(len(td)), C.c_double(1e-6), _p(pts), _p(diffs), C.c_int(cap))
    assert n > 0 and diffs[:n].max() < 1e-6
    # noise-free: every matched triangle pair is a congruent pair, and its sorted vertices correspond under the true motion
    a, b = pts[:n, :, :2].reshape(-1, 2), pts[:n, :, 2:].reshape(-1, 2)
    assert np.abs((b @ R.T + t) - a).max() < 1e-6          # reference = R query + t
    tf = np.zeros(9)
    po.lib().orc_estimate_tf2d(_p(np.ascontiguousarray(b)), _p(np.ascontiguousarray(a)), C.c_int(len(a)), _p(tf))
    tf = tf.reshape(3, 3)
    assert np.abs(tf[:2, :2] - R).max() < 1e-9 and np.abs(tf[:2, 2] - t).max() < 1e-8
    # a reflected point set: the reference's "negate column 1" fix still returns a proper rotation
    po.lib().orc_estimate_tf2d(_p(np.ascontiguousarray(b * [1, -1])), _p(np.ascontiguousarray(a)), C.c_int(len(a)), _p(tf.reshape(-1)))
    assert abs(np.linalg.det(tf[:2, :2]) - 1.0) < 1e-12


def test_oracle_profile_cholesky_equals_the_dense_loops():
    """The oracle's blocked Cholesky and substitutions work inside the row profile of the assembled matrix (graph.hpp chol_profile,
    what keeps the CPU baseline from being a dense n^3 / 3 on a pose chain).  The entries skipped are exact zeros, so the solution must
    equal the dense loops' bit for bit — on a band, on a band with a far coupling (fill up to it), and on a dense matrix."""
    import ctypes as C
    L = po.lib()
    rng = np.random.default_rng(7)
    n = 300
    for case in ("band", "band+loop", "dense"):
        G = np.zeros((n, n))
        if case == "dense":
            G = rng.normal(size=(n, n))
        else:
            for i in range(n):
                j0 = max(0, i - 20)
                G[i, j0:i + 1] = rng.normal(size=i + 1 - j0)
            if case == "band+loop":
                G[250:256, 10:16] = rng.normal(size=(6, 6))
        A = np.tril(G @ G.T + n * np.eye(n)) if case == "dense" else None
        if A is None:
            # a banded SPD matrix: B B^T of a banded lower factor keeps the band; the far block is added symmetrically and dominated by the diagonal
            B = np.tril(G)
            A = B @ B.T + n * np.eye(n)
            mask = np.abs(np.subtract.outer(np.arange(n), np.arange(n))) <= 20
            if case == "band+loop":
                mask[250:256, 10:16] = True
                mask[10:16, 250:256] = True
            A = np.tril(A * mask)
        A = np.ascontiguousarray(A)
        b = rng.normal(size=n)
        xd, xp = np.zeros(n), np.zeros(n)
        rows = C.c_int(0)
        rc = L.orc_chol_solve_both(A.ctypes.data_as(C.c_void_p), C.c_int(n), b.ctypes.data_as(C.c_void_p), xd.ctypes.data_as(C.c_void_p),
                                   xp.ctypes.data_as(C.c_void_p), C.byref(rows))
        assert rc == 0, case
        assert np.array_equal(xd, xp), case
        full = A + np.tril(A, -1).T
        assert np.abs(full @ xd - b).max() < 1e-9 * np.abs(b).max() * np.linalg.cond(full)
        if case == "band":
            assert rows.value < n * n // 4        # the profile is a band
