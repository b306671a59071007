"""The step kernels (one block column per launch) and the pair kernel (two per launch) on BORDERED systems with a profile, as the exact
joint passes hand them to the factorisation: a band segment with its active border rows in their own order, the bands'
second level, a leaf of the separator system.  The system is built from a known factor: A = L0 L0^T inside a monotone tile profile,
border rows B = W0 L0^T (zero before their first block column), right-hand side b = L0 y0 — so the factorisation must return L0, W0 and y0
(numpy only states the expectation; reference: ISAM2Params::CHOLESKY of backend/sloam/src/factorgraph/graph.cpp:15, whose elimination of
a pose chain with shared landmarks this is)."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NB = 64


def _system(rng, T, nbr, prof=None, first=None, ord_=None, b0=0, kofs=0):
    """Returns (S flat, ld, L0, W0 (physical row order), y0).  first[j]: first block column (robot numbering: + kofs) of the j-th active
    border row, physical tile row ord_[j]; 1 << 30 = never."""
    n = T * NB
    B0 = b0 or T
    ld = (B0 + nbr + 1) * NB
    pf = list(prof) if prof is not None else [T - 1] * T
    L0 = np.zeros((n, n))
    for c in range(T):
        for r in range(c, pf[c] + 1):
            blk = rng.normal(size=(NB, NB)) * (0.25 / np.sqrt(NB))
            if r == c:
                blk = np.tril(blk)
                blk[np.diag_indices(NB)] = 1.0 + rng.uniform(0, 1, NB)
            L0[r * NB:(r + 1) * NB, c * NB:(c + 1) * NB] = blk
    A = L0 @ L0.T
    W0 = np.zeros((nbr * NB, n))
    for j in range(nbr):
        f = 0 if first is None else first[j] - kofs
        row = j if ord_ is None else ord_[j]
        if first is not None and first[j] >= (1 << 30):
            continue
        W0[row * NB:(row + 1) * NB, max(f, 0) * NB:] = rng.normal(size=(NB, n - max(f, 0) * NB)) * 0.5
    Bm = W0 @ L0.T
    y0 = rng.normal(size=n)
    b = L0 @ y0
    S = np.zeros((n, ld))                      # S[col, row]: column-major
    S[:, T * NB:] = 1e30                        # whatever lies between a view's band and its border: never to be touched
    S[:, B0 * NB:] = 0.0
    for c in range(n):
        S[c, :n] = A[:, c]                      # (both triangles: the kernels read the lower one and the diagonal sub-tiles)
    for c in range(T):                          # outside the profile: structurally zero in memory
        S[c * NB:(c + 1) * NB, (pf[c] + 1) * NB:n] = 0.0
    S[:, B0 * NB:(B0 + nbr) * NB] = Bm.T
    S[:, (B0 + nbr) * NB] = b
    return S.ravel(), ld, L0, W0, y0


def _check(gpu, S, ld, T, nbr, L0, W0, y0, prof, first, ord_, b0, kofs, n_copies, method, tol=2e-10):
    out = gpu.api.debug_chol_bordered(S, ld, T, nbr, prof=prof, bfirst=first, ord=ord_, b0=b0, kofs=kofs, n_copies=n_copies, method=method)
    assert gpu.pair_timeouts() == 0
    assert out["status"][1] == 0, "flagged not positive definite"
    assert out["copy_diff"] == 0.0, "the copies of one system in a launch sequence differ"
    n = T * NB
    B0 = b0 or T
    So = out["S"].reshape(n, ld)
    pf = list(prof) if prof is not None else [T - 1] * T
    scale = np.abs(L0).max()
    for c in range(T):
        for r in range(c + 1, pf[c] + 1):
            got = So[c * NB:(c + 1) * NB, r * NB:(r + 1) * NB].T
            want = L0[r * NB:(r + 1) * NB, c * NB:(c + 1) * NB]
            assert np.abs(got - want).max() <= tol * scale * 50, ("band tile", r, c, np.abs(got - want).max())
    gotW = So[:, B0 * NB:(B0 + nbr) * NB].T
    assert np.abs(gotW - W0).max() <= tol * max(np.abs(W0).max(), 1.0) * 50, ("border", np.abs(gotW - W0).max())
    goty = So[:, (B0 + nbr) * NB]
    assert np.abs(goty - y0).max() <= tol * np.abs(y0).max() * 50, ("rhs", np.abs(goty - y0).max())
    Ld = out["Ld"].reshape(T, NB, NB)           # [k][col][row]
    Wi = out["Winv"].reshape(T, 4, 16, 16)      # [k][b][c][r] = inv(L_bb)[r][c]
    for k in range(T):
        D = L0[k * NB:(k + 1) * NB, k * NB:(k + 1) * NB]
        for I in range(4):
            for J in range(I):
                got = Ld[k, 16 * J:16 * J + 16, 16 * I:16 * I + 16].T
                assert np.abs(got - D[16 * I:16 * I + 16, 16 * J:16 * J + 16]).max() <= tol * 50, ("Ld", k, I, J)
            inv = np.linalg.inv(D[16 * I:16 * I + 16, 16 * I:16 * I + 16])
            assert np.abs(Wi[k, I].T - inv).max() <= 1e-9 * np.abs(inv).max(), ("Winv", k, I)
    # rows between the band and a view's border, and everything outside the profile, are as they were
    if B0 > T:
        assert np.all(So[:, T * NB:B0 * NB] == 1e30)
    for c in range(T):
        assert np.all(So[c * NB:(c + 1) * NB, (pf[c] + 1) * NB:n] == 0.0), ("outside the profile, column", c)


CASES = {
    # the bands' second level: dense, few block columns, a dozen border rows, eight robots side by side
    "second_level_T4": dict(T=4, nbr=12, n_copies=8),
    "second_level_T5": dict(T=5, nbr=11, n_copies=8),
    "one_column": dict(T=1, nbr=3, n_copies=2),
    "two_columns": dict(T=2, nbr=1, n_copies=1),
    # a leaf of the separator system: dense profile, the top block's rows as border further down
    "leaf": dict(T=9, nbr=5, b0=9 + 7, kofs=0, n_copies=2),
    # a band segment: profile three tiles wide, border rows that start at different block columns in an order of their own
    "segment": dict(T=12, nbr=9, band=3, first="sorted", b0=12 + 4, kofs=21, n_copies=6),
    "segment_odd": dict(T=11, nbr=6, band=2, first="sorted", b0=11 + 2, kofs=3, n_copies=3),
    "segment_decoupled": dict(T=8, nbr=4, band=2, first="sorted", kofs=0, n_copies=2, cut=4),
}


@pytest.mark.parametrize("method", [0, 2])
@pytest.mark.parametrize("case", sorted(CASES))
def test_bordered_factorisation_returns_the_known_factor(gpu, case, method):
    kw = dict(CASES[case])
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    T, nbr = kw["T"], kw["nbr"]
    b0, kofs, n_copies = kw.get("b0", 0), kw.get("kofs", 0), kw.get("n_copies", 1)
    prof = None
    if "band" in kw:
        prof = [min(c + kw["band"], T - 1) for c in range(T)]
        if "cut" in kw:                       # block column cut - 1 couples to nothing below it (a decoupled chain: prof[c] == c)
            for c in range(kw["cut"]):
                prof[c] = min(prof[c], kw["cut"] - 1)
    first = ord_ = None
    if kw.get("first") == "sorted":
        # first columns in the robot's numbering: a few rows from the segment's first column on (or before it), others later, one never
        f = sorted(int(x) for x in rng.integers(kofs - 2, kofs + T, nbr - 1)) + [1 << 30]
        f = [max(x, 0) for x in f]
        first = f
        ord_ = [int(x) for x in rng.permutation(nbr)]
    S, ld, L0, W0, y0 = _system(rng, T, nbr, prof, first, ord_, b0, kofs)
    _check(gpu, S, ld, T, nbr, L0, W0, y0, prof, first, ord_, b0, kofs, n_copies, method)
