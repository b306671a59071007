"""GPU parity tests of the individual kernels, through the C-ABI, against the oracle / numpy."""
import ctypes as C
import zlib

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _spd(n, rng, cond=1e4):
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    d = np.exp(rng.uniform(0, np.log(cond), n))
    return (Q * d) @ Q.T


@pytest.mark.parametrize("n", [6, 64, 65, 130, 150, 200, 1000, 1990])   # T = 1, 1, 2, 3, 3, 4, 16, 32 block columns (odd and even, with and without padding)
def test_dense_spd_solve(gpu, n):
    """FP64-MFMA blocked Cholesky + substitutions vs numpy (fp64): relative 1e-9 on a cond-1e4 system."""
    rng = np.random.default_rng(n)
    A = _spd(n, rng)
    b = rng.normal(size=n)
    x, _ = gpu.dense_spd_solve(A, b)
    ref = np.linalg.solve(A, b)
    assert np.linalg.norm(x - ref) <= 1e-9 * np.linalg.norm(ref)


@pytest.mark.parametrize("n", [6, 64, 65, 130, 200, 330, 1000, 1990, 3776])   # T = 1, 1, 2, 3, 4, 6, 16, 32, 59 block columns
def test_dense_spd_solve_left_looking_persistent(gpu, n):
    """The left-looking persistent factorisation (k_chol_ll: ONE launch for all block columns, chain and tile tasks started in ticket
    order, flags + release / acquire fences between them instead of kernel boundaries) vs numpy AND vs the step kernels' result:
    the same arithmetic per tile, only the order of the left-looking sums differs."""
    rng = np.random.default_rng(n)
    A = _spd(n, rng)
    b = rng.normal(size=n)
    x, _ = gpu.dense_spd_solve(A, b, method=1)
    ref = np.linalg.solve(A, b)
    assert np.linalg.norm(x - ref) <= 1e-9 * np.linalg.norm(ref)
    x0, _ = gpu.dense_spd_solve(A, b, method=0)
    assert np.linalg.norm(x - x0) <= 1e-10 * np.linalg.norm(ref)
    # and again on the same buffers (the flags and the ticket counter are cleared per launch), timed
    x2, ms = gpu.dense_spd_solve(A, b, repeats=3, method=1)
    assert np.array_equal(x2, x), "the persistent factorisation is not deterministic"


@pytest.mark.parametrize("n", [6, 64, 65, 128, 130, 200, 256, 330, 1000, 1990, 3776])   # T = 1, 1, 2, 2, 3, 4, 4, 6, 16, 32, 59: odd and even, the last column alone or in a pair
def test_dense_spd_solve_two_columns_per_launch(gpu, n):
    """The pair kernel (k_chol_pair_batched: block columns k, k + 1 in ONE launch; every type-A workgroup carries the sub-diagonal tile
    and the next diagonal block redundantly, the previous pair's panels as pending panels, one rank-128 pass per launch behind them) vs
    numpy AND vs the step kernels' result; deterministic; no flag wait gave up."""
    rng = np.random.default_rng(n)
    A = _spd(n, rng)
    b = rng.normal(size=n)
    x, _ = gpu.dense_spd_solve(A, b, method=2)
    ref = np.linalg.solve(A, b)
    assert gpu.pair_timeouts() == 0
    assert np.linalg.norm(x - ref) <= 1e-9 * np.linalg.norm(ref)
    x0, _ = gpu.dense_spd_solve(A, b, method=0)
    assert np.linalg.norm(x - x0) <= 1e-10 * np.linalg.norm(ref)
    x2, ms = gpu.dense_spd_solve(A, b, repeats=3, method=2)
    assert np.array_equal(x2, x), "the pair kernel is not deterministic"


def test_dense_spd_solve_two_columns_per_launch_rejects_indefinite(gpu):
    for bad in (40, 100, 140):      # in the first column of a pair, in the second, in a later pair
        A = np.eye(200)
        A[bad, bad] = -1.0
        with pytest.raises(gpu.SlideError):
            gpu.dense_spd_solve(A, np.ones(200), method=2)
    assert gpu.pair_timeouts() == 0


def test_dense_spd_solve_left_looking_rejects_indefinite(gpu):
    A = np.eye(200)
    A[140, 140] = -1.0
    with pytest.raises(gpu.SlideError):
        gpu.dense_spd_solve(A, np.ones(200), method=1)


def test_dense_spd_solve_more_block_columns_than_cus(gpu):
    """T = 261 block columns: more workgroups in the chained backward substitution than the part has CUs (its waits are on
    workgroups of lower index only) and several rounds of type-A workgroups in the first step kernels; residual check."""
    n = 16700
    rng = np.random.default_rng(1)
    B = rng.normal(size=(n, 64))
    A = B @ B.T / 64.0
    A[np.diag_indices(n)] += 2.0 + rng.uniform(0, 1, n)
    b = rng.normal(size=n)
    x, _ = gpu.dense_spd_solve(A, b)
    assert np.linalg.norm(A @ x - b) <= 1e-12 * np.linalg.norm(b)


def test_dense_spd_solve_rejects_indefinite(gpu):
    A = np.eye(70)
    A[40, 40] = -1.0
    with pytest.raises(gpu.SlideError):
        gpu.dense_spd_solve(A, np.ones(70))


@pytest.mark.parametrize("n,K", [(1, 30), (29, 30), (700, 1000), (5000, 1000), (16384, 50), (10000, 1), (30000, 1000), (50000, 1000),
                                 (20000, 16384), (3000, 3000)])
def test_submap_knn_matches_oracle(gpu, n, K):
    """Exact float32 K-NN gate (radix K-select + sort of the K survivors): identical index lists (nearest first, ties by index).
    30000 points: the distance words still fit in LDS; 50000: recomputed per select pass; K >= n: no selection at all."""
    rng = np.random.default_rng(n + K)
    cloud = rng.uniform(-60, 60, (n, 3)).astype(np.float32)
    if n > 10:
        cloud[5] = cloud[3]          # exact distance tie
    q = rng.uniform(-60, 60, 3)
    got = gpu.submap_knn(cloud, q, K)
    out = np.zeros(max(min(K, n), 1), np.int32)
    k = po.lib().orc_knn_f32(cloud.ctypes.data_as(C.c_void_p), C.c_int(n), q.ctypes.data_as(C.c_void_p), C.c_int(K),
                             out.ctypes.data_as(C.c_void_p))
    assert k == len(got)
    assert np.array_equal(got, out[:k])


@pytest.mark.parametrize("case", ["all_equal", "two_shells", "tie_across_the_cut", "dense_cluster_at_the_cut", "one_far_outlier",
                                  "robot_on_a_landmark"])
def test_submap_knn_ties_at_the_selection_boundary(gpu, case):
    """Many exactly equidistant points around the K-th neighbour: the select has to go on into the index bits (lowest indices
    win), which is the reference order the oracle defines for FLANN's unspecified tie order.  Round 4 (K-th key found by a histogram
    over bins linear in the squared distance): clouds that defeat the bins — thousands of DISTINCT distances inside one bin at the
    cut, one landmark so far away that every other falls into bin 0, the robot standing on a landmark — go through the digit passes
    restricted to the K-th key's bin and must give the same lists."""
    rng = np.random.default_rng(3)
    n, K = 6000, 1000
    if case == "dense_cluster_at_the_cut":
        cloud = rng.uniform(-60, 60, (n, 3)).astype(np.float32)
        sh = rng.normal(0, 1, (3000, 3))
        sh = sh / np.linalg.norm(sh, axis=1)[:, None] * (20.0 + rng.uniform(0, 1e-3, 3000))[:, None]      # a thin shell through the cut
        cloud[rng.permutation(n)[:3000]] = sh.astype(np.float32)
    elif case == "one_far_outlier":
        cloud = rng.uniform(-60, 60, (n, 3)).astype(np.float32)
        cloud[17] = (3.0e6, 0.0, 0.0)
    elif case == "robot_on_a_landmark":
        cloud = rng.uniform(-60, 60, (n, 3)).astype(np.float32)
        cloud[40] = 0.0
        cloud[41] = 0.0
    elif case == "all_equal":
        cloud = np.tile(np.array([[3.0, 4.0, 0.0]], np.float32), (n, 1))
    elif case == "two_shells":
        ang = rng.uniform(0, 2 * np.pi, n)
        r = np.where(np.arange(n) % 3 == 0, 5.0, 9.0)
        cloud = np.zeros((n, 3), np.float32)
        cloud[:, 0] = 0.0; cloud[:, 1] = 0.0; cloud[:, 2] = r          # two distinct distances, thousands of ties each
    else:
        cloud = rng.uniform(-60, 60, (n, 3)).astype(np.float32)
        d = ((cloud.astype(np.float64)) ** 2).sum(1)
        kth = np.argsort(d, kind="stable")[K - 1]
        for j in rng.permutation(n)[:40]:
            cloud[j] = cloud[kth]                                       # 40 copies of the K-th neighbour itself
    q = np.zeros(3)
    got = gpu.submap_knn(cloud, q, K)
    out = np.zeros(K, np.int32)
    k = po.lib().orc_knn_f32(cloud.ctypes.data_as(C.c_void_p), C.c_int(n), q.ctypes.data_as(C.c_void_p), C.c_int(K),
                             out.ctypes.data_as(C.c_void_p))
    assert k == K == len(got)
    assert np.array_equal(got, out)


def test_knn_capacity_error(gpu):
    """The map is unbounded; only K (a configuration constant: 50 / 30 / 1000 in the reference) is limited by the LDS sort buffer."""
    cloud = np.zeros((20000, 3), np.float32)
    assert len(gpu.submap_knn(cloud, np.zeros(3), 10)) == 10
    with pytest.raises(gpu.SlideError, match="sort buffer"):
        gpu.submap_knn(cloud, np.zeros(3), 16385)


def test_assoc_sweep_batch_matches_oracle(gpu):
    """The batched association sweep at the headline sizes (BASELINE configs[3]: N_map = 10 k, K = 1000, N_obs = 20) over 5000 query
    frames: identical map indices to the oracle's getSubmap + matchEllipsoidModels per frame.  Inputs: synth.assoc_sweep_case, the
    generator of bench.py's association leg (the timed data is the tested data)."""
    from slide_slam_amd.synth import assoc_sweep_case
    n_map, K, n_obs, n_q = 10000, 1000, 20, 5000
    cloud, model, label, qpos, obs, olab = assoc_sweep_case(77, n_map, n_obs, n_q)      # the generator bench.py's association leg times
    got, ms = gpu.assoc_sweep_batch(cloud, model, label, qpos, obs, olab, K, 0.75)
    assert got.shape == (n_q, n_obs) and ms > 0
    L = po.lib()
    sub = np.zeros(K, np.int32)
    exp = np.full(n_obs, -1, np.int32)
    n_match = 0
    for i in range(0, n_q, 5):                                                   # every fifth frame through the oracle (seconds)
        k = L.orc_knn_f32(_p(cloud), C.c_int(n_map), _p(qpos[i]), C.c_int(K), _p(sub))
        assert k == K
        sm = np.ascontiguousarray(model[sub]); sl = np.ascontiguousarray(label[sub])
        L.orc_match_boxes(C.c_int(2), C.c_int(n_obs), _p(np.ascontiguousarray(obs[i])), _p(np.ascontiguousarray(olab[i])), C.c_int(K),
                          _p(sm), _p(sl), C.c_double(0.75), _p(exp))
        want = np.where(exp >= 0, sub[np.maximum(exp, 0)], -1)
        assert np.array_equal(got[i], want), i
        n_match += int((want >= 0).sum())
    assert n_match > 0.5 * (n_q // 5) * n_obs


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_submap_knn_randomized_clouds(gpu):
    """The select over clouds of many shapes and sizes (uniform, Gaussian clusters, a lattice with thousands of exact ties, a line, a
    cloud of duplicates, distances spanning twelve orders of magnitude), K from 1 to beyond n: identical lists to the oracle.  Covers
    the histogram path, the candidate ranks, the digit-pass fallback and the uncached map (n beyond the LDS cache)."""
    rng = np.random.default_rng(2024)
    kinds = ["uniform", "clusters", "lattice", "line", "duplicates", "wide"]
    for case in range(48):
        kind = kinds[case % len(kinds)]
        n = int(rng.choice([1, 2, 63, 64, 65, 500, 1023, 1025, 4000, 9000, 20000, 41000]))
        K = int(rng.choice([1, 2, 50, 63, 64, 65, 1000, 1024, 2500]))
        if kind == "uniform":
            cloud = rng.uniform(-80, 80, (n, 3))
        elif kind == "clusters":
            c = rng.uniform(-80, 80, (8, 3))
            cloud = c[rng.integers(0, 8, n)] + rng.normal(0, 0.5, (n, 3))
        elif kind == "lattice":
            cloud = rng.integers(-6, 7, (n, 3)).astype(np.float64) * 2.0
        elif kind == "line":
            cloud = np.zeros((n, 3)); cloud[:, 0] = rng.uniform(-200, 200, n)
        elif kind == "duplicates":
            base = rng.uniform(-30, 30, (max(1, n // 50), 3))
            cloud = base[rng.integers(0, len(base), n)]
        else:
            cloud = rng.normal(0, 1, (n, 3)) * 10.0 ** rng.uniform(-6, 6, (n, 1))
        cloud = np.ascontiguousarray(cloud.astype(np.float32))
        q = rng.uniform(-5, 5, 3) if kind != "lattice" else np.zeros(3)
        got = gpu.submap_knn(cloud, q, K)
        out = np.zeros(max(min(K, n), 1), np.int32)
        k = po.lib().orc_knn_f32(cloud.ctypes.data_as(C.c_void_p), C.c_int(n), q.ctypes.data_as(C.c_void_p), C.c_int(K),
                                 out.ctypes.data_as(C.c_void_p))
        assert k == len(got) == min(K, n), (case, kind, n, K)
        assert np.array_equal(got, out[:k]), (case, kind, n, K)


@pytest.mark.parametrize("n_obs,n_labels,lab0", [(1, 1, 5), (20, 12, -3), (64, 7, 1000000), (70, 3, 1), (33, 40, 0)])
def test_assoc_sweep_label_groups_and_detection_counts(gpu, n_obs, n_labels, lab0):
    """The matching groups the staged survivors by label when a frame has at most 64 detections and scans all survivors behind a label
    select beyond that; labels are arbitrary integers.  One detection, 64, 70 (ungrouped), more distinct labels than detections can
    carry, negative and large labels, detections whose label no landmark has: identical map indices to the oracle."""
    rng = np.random.default_rng(n_obs * 131 + n_labels)
    n_map, K, n_q = 5000, 1000, 40
    model = np.column_stack([rng.uniform(0, 150, n_map), rng.uniform(0, 150, n_map), rng.normal(0, 0.3, n_map)])
    cloud = (model + rng.normal(0, 0.05, model.shape)).astype(np.float32)
    label = (lab0 + 7 * rng.integers(0, n_labels, n_map)).astype(np.int32)
    qpos = np.column_stack([rng.uniform(30, 120, n_q), rng.uniform(30, 120, n_q), np.full(n_q, 2.0)])
    obs = np.zeros((n_q, n_obs, 3)); olab = np.zeros((n_q, n_obs), np.int32)
    for i in range(n_q):
        d2 = ((model[:, :2] - qpos[i, :2]) ** 2).sum(1)
        near = np.argsort(d2)[:n_obs]
        obs[i] = model[near] + rng.normal(0, 0.1, (n_obs, 3))
        olab[i] = label[near]
        olab[i, ::5] = lab0 + 7 * n_labels + 3                                   # a label no landmark carries
    got, _ = gpu.assoc_sweep_batch(cloud, model, label, qpos, obs, olab, K, 0.75)
    L = po.lib()
    sub = np.zeros(K, np.int32)
    exp = np.full(n_obs, -1, np.int32)
    n_match = 0
    for i in range(n_q):
        k = L.orc_knn_f32(_p(cloud), C.c_int(n_map), _p(qpos[i]), C.c_int(K), _p(sub))
        assert k == K
        sm = np.ascontiguousarray(model[sub]); sl = np.ascontiguousarray(label[sub])
        L.orc_match_boxes(C.c_int(2), C.c_int(n_obs), _p(np.ascontiguousarray(obs[i])), _p(np.ascontiguousarray(olab[i])), C.c_int(K),
                          _p(sm), _p(sl), C.c_double(0.75), _p(exp))
        want = np.where(exp >= 0, sub[np.maximum(exp, 0)], -1)
        assert np.array_equal(got[i], want), (i, got[i], want)
        n_match += int((want >= 0).sum())
    assert n_match > 0.5 * n_q * n_obs * 0.8 or n_obs == 1


def _sweep_vs_oracle(gpu, cloud, model, label, qpos, obs, olab, K, thresh=0.75):
    """gpu.assoc_sweep_batch against the oracle's getSubmap + matchEllipsoidModels, frame by frame; returns the matches found."""
    n_map, (n_q, n_obs) = len(cloud), olab.shape
    got, _ = gpu.assoc_sweep_batch(cloud, model, label, qpos, obs, olab, K, thresh)
    L = po.lib()
    k_eff = min(K, n_map)
    sub = np.zeros(max(k_eff, 1), np.int32)
    exp = np.full(n_obs, -1, np.int32)
    n_match = 0
    for i in range(n_q):
        k = L.orc_knn_f32(_p(cloud), C.c_int(n_map), _p(qpos[i]), C.c_int(K), _p(sub))
        assert k == k_eff
        sm = np.ascontiguousarray(model[sub[:k]]); sl = np.ascontiguousarray(label[sub[:k]])
        L.orc_match_boxes(C.c_int(2), C.c_int(n_obs), _p(np.ascontiguousarray(obs[i])), _p(np.ascontiguousarray(olab[i])), C.c_int(k),
                          _p(sm), _p(sl), C.c_double(thresh), _p(exp))
        want = np.where(exp >= 0, sub[np.maximum(exp, 0)], -1)
        assert np.array_equal(got[i], want), (i, got[i], want)
        n_match += int((want >= 0).sum())
    return n_match


@pytest.mark.parametrize("kind", ["coincident", "outlier", "tiny", "full_rounds", "k_beyond_n", "one", "beyond_rounds"])
def test_assoc_sweep_degenerate_clouds(gpu, kind):
    """The register-resident sweep kernel (k_assoc_sweep_r, round 5) on the clouds that leave its main path: thousands of coincident
    landmarks around the K-th neighbour (the K-th key's bin overflows the candidate list: digit passes, ties by map index), one far
    outlier (every other key in the first bin), maps smaller than a workgroup, a map that fills all twenty rounds of 512 keys exactly,
    K beyond the map, a map of one landmark — and a map of 12 000 landmarks, beyond the twenty rounds (round 4's kernel takes it).
    Identical map indices to the oracle."""
    rng = np.random.default_rng(zlib.crc32(kind.encode()))
    n_obs, n_q, K = 12, 24, 1000
    n_map = dict(coincident=6000, outlier=5000, tiny=300, full_rounds=10240, k_beyond_n=700, one=1, beyond_rounds=12000)[kind]
    model = np.column_stack([rng.uniform(0, 120, n_map), rng.uniform(0, 120, n_map), rng.normal(0, 0.3, n_map)])
    if kind == "coincident":
        model[1500:4500] = model[1500]                       # 3000 landmarks in one spot
        model[4500:4600] = model[1500] + rng.normal(0, 1e-4, (100, 3))
    if kind == "outlier":
        model[777] = [1.0e6, -2.0e6, 5.0e5]
    label = rng.integers(1, 5, n_map).astype(np.int32)
    cloud = model.astype(np.float32)
    qpos = np.column_stack([rng.uniform(30, 90, n_q), rng.uniform(30, 90, n_q), np.full(n_q, 2.0)])
    if kind == "coincident":
        qpos[::2] = model[1500] + rng.normal(0, 8.0, (len(qpos[::2]), 3))      # the spot within the K nearest, partly
    obs = np.zeros((n_q, n_obs, 3)); olab = np.zeros((n_q, n_obs), np.int32)
    for i in range(n_q):
        near = np.argsort(((model[:, :2] - qpos[i, :2]) ** 2).sum(1))[:n_obs]
        near = np.resize(near, n_obs)
        obs[i] = model[near] + rng.normal(0, 0.1, (n_obs, 3))
        olab[i] = label[near]
    n_match = _sweep_vs_oracle(gpu, cloud, model, label, qpos, obs, olab, K)
    assert n_match > 0.5 * n_q * n_obs


@pytest.mark.parametrize("offset", [0.0, 7000.0])
def test_assoc_sweep_near_ties_survive_the_float_screening(gpu, offset):
    """The matching screens the submap in float (coordinates relative to the robot) and applies the reference's double-precision rule
    to the candidates inside the error bound of the float minimum (assoc_kernels.hip).  Cases the screening could get wrong if the
    bound were too tight: a detection exactly halfway between two landmarks of its label (an exact double-precision tie: the first
    in submap order wins), pairs whose distances differ by 1e-13 .. 1e-7 relative (far below float resolution), all of it 7 km from
    the origin (float spacing there: 0.5 mm).  Identical map indices to the oracle's getSubmap + matchEllipsoidModels."""
    rng = np.random.default_rng(123)
    n_map, K, n_obs, n_q = 4000, 1000, 20, 64
    model = np.column_stack([rng.uniform(0, 200, n_map), rng.uniform(0, 100, n_map), rng.normal(0, 0.3, n_map)]) + offset
    label = rng.integers(1, 4, n_map).astype(np.int32)
    qpos = np.column_stack([rng.uniform(40, 160, n_q), rng.uniform(30, 70, n_q), np.full(n_q, 2.0)]) + offset
    obs = np.zeros((n_q, n_obs, 3)); olab = np.zeros((n_q, n_obs), np.int32)
    eps = [0.0, 1e-13, 1e-11, 1e-9, 1e-7]
    for i in range(n_q):
        d2 = ((model[:, :2] - qpos[i, :2]) ** 2).sum(1)
        near = np.argsort(d2)[:2 * n_obs]
        for o in range(n_obs):
            a, b = near[2 * o], near[2 * o + 1]
            # landmark b becomes the mirror image of landmark a about the detection, pulled in by a relative eps
            det = model[a] + rng.normal(0, 0.15, 3)
            e = eps[(i + o) % len(eps)]
            model[b] = det + (det - model[a]) * (1.0 - e)
            label[b] = label[a]
            obs[i, o] = det
            olab[i, o] = label[a]
    cloud = model.astype(np.float32)
    got, _ = gpu.assoc_sweep_batch(cloud, model, label, qpos, obs, olab, K, 0.75)
    L = po.lib()
    sub = np.zeros(K, np.int32)
    exp = np.full(n_obs, -1, np.int32)
    n_match = 0
    for i in range(n_q):
        k = L.orc_knn_f32(_p(cloud), C.c_int(n_map), _p(qpos[i]), C.c_int(K), _p(sub))
        assert k == K
        sm = np.ascontiguousarray(model[sub]); sl = np.ascontiguousarray(label[sub])
        L.orc_match_boxes(C.c_int(2), C.c_int(n_obs), _p(np.ascontiguousarray(obs[i])), _p(np.ascontiguousarray(olab[i])), C.c_int(K),
                          _p(sm), _p(sl), C.c_double(0.75), _p(exp))
        want = np.where(exp >= 0, sub[np.maximum(exp, 0)], -1)
        assert np.array_equal(got[i], want), (i, got[i], want)
        n_match += int((want >= 0).sum())
    assert n_match > 0.8 * n_q * n_obs


@pytest.mark.parametrize("cls", [1, 2])
@pytest.mark.parametrize("n_cur,n_map", [(0, 5), (7, 0), (20, 1000), (64, 333)])
def test_match_boxes(gpu, cls, n_cur, n_map):
    rng = np.random.default_rng(cls * 1000 + n_cur + n_map)
    mxyz = rng.uniform(-20, 20, (n_map, 3))
    mlab = rng.integers(1, 4, n_map).astype(np.int32)
    xyz = rng.uniform(-20, 20, (n_cur, 3))
    lab = rng.integers(1, 4, n_cur).astype(np.int32)
    if n_cur and n_map:
        k = min(n_cur, n_map) // 2
        xyz[:k] = mxyz[:k] + rng.normal(0, 0.2, (k, 3))      # true matches
        lab[:k] = mlab[:k]
        if n_map > 10:
            mxyz[9] = mxyz[2]                                 # exact duplicate -> first index must win
            mlab[9] = mlab[2]
    thr = 2.0 if cls == 1 else 0.75
    got = gpu.match_boxes(cls, xyz, lab, mxyz, mlab, thr)
    exp = np.full(max(n_cur, 1), -1, np.int32)
    po.lib().orc_match_boxes(C.c_int(cls), C.c_int(n_cur), _p(np.ascontiguousarray(xyz)), _p(lab), C.c_int(n_map),
                             _p(np.ascontiguousarray(mxyz)), _p(mlab), C.c_double(thr), _p(exp))
    assert np.array_equal(got, exp[:n_cur])


@pytest.mark.parametrize("n_cur,n_map", [(0, 3), (5, 0), (12, 50), (40, 300)])
def test_match_cylinders(gpu, n_cur, n_map):
    rng = np.random.default_rng(n_cur * 7 + n_map)
    mroot = np.column_stack([rng.uniform(-30, 30, (n_map, 2)), rng.normal(0, 0.3, n_map)])
    mray = np.column_stack([rng.normal(0, 0.02, (n_map, 2)), np.ones(n_map)])
    mlab = rng.integers(1, 3, n_map).astype(np.int32)
    root = np.column_stack([rng.uniform(-30, 30, (n_cur, 2)), rng.normal(0, 0.3, n_cur)])
    ray = np.column_stack([rng.normal(0, 0.02, (n_cur, 2)), np.ones(n_cur)])
    lab = rng.integers(1, 3, n_cur).astype(np.int32)
    if n_cur and n_map:
        k = min(n_cur, n_map) // 2
        root[:k] = mroot[:k] + rng.normal(0, 0.3, (k, 3))
        lab[:k] = mlab[:k]
    got = gpu.match_cylinders(root, ray, lab, mroot, mray, mlab, 2.0)
    exp = np.full(max(n_cur, 1), -1, np.int32)
    rad = np.zeros(max(n_cur, 1)); mrad = np.zeros(max(n_map, 1))
    po.lib().orc_match_cylinders(C.c_int(n_cur), _p(np.ascontiguousarray(root)), _p(np.ascontiguousarray(ray)), _p(rad), _p(lab),
                                 C.c_int(n_map), _p(np.ascontiguousarray(mroot)), _p(np.ascontiguousarray(mray)), _p(mrad),
                                 _p(mlab), C.c_double(2.0), _p(exp))
    assert np.array_equal(got, exp[:n_cur])
