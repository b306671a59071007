"""Layout of the separator system of the exact joint step (slide_slam_amd.distributed.separator_offsets): the nested dissection over the
robots that slide_chol_batch_set_separator_blocks factors leaf by leaf, and the packed exchange buffer's segments a job that spans GPUs
all-reduces (slide_chol_batch_sep_segment / _sep_exchange_len).  CPU only: layout arithmetic, no device."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from slide_slam_amd import distributed as d          # noqa: E402

DIM = d.SLOT_DIM


def _grid_job(rng, n_per_group=40):
    """Eight robots on a 2 x 4 grid of cells (robots 0-3 the top row, 4-7 the bottom row): shared landmarks between horizontal and
    vertical neighbours and a few seen from three cells."""
    groups = [(0, 1), (1, 2), (2, 3), (4, 5), (5, 6), (6, 7), (0, 4), (1, 5), (2, 6), (3, 7), (0, 4, 5), (1, 5, 6), (2, 6, 7)]
    obs = []
    for g in groups:
        for _ in range(int(rng.integers(n_per_group // 2, n_per_group))):
            obs.append((int(rng.integers(0, 3)), frozenset(g)))
    order = rng.permutation(len(obs))
    return [obs[i] for i in order]


def _layout(obs, R, **kw):
    saved = d.slot_observers
    d.slot_observers = lambda gid, n_global: obs
    try:
        return d.separator_offsets([None] * R, None, **kw)
    finally:
        d.slot_observers = saved


def _check_disjoint(obs, off):
    m = int(off[-1])
    used = np.zeros(m, bool)
    for i, (cls, _) in enumerate(obs):
        a, b = int(off[i]), int(off[i]) + DIM[cls]
        assert 0 <= a and b <= m and not used[a:b].any()
        used[a:b] = True
    return used


def test_plain_layout_is_a_permutation_of_the_slots_coordinates():
    obs = _grid_job(np.random.default_rng(1))
    off, prof, blocks = _layout(obs, 8, dissect=False)
    assert blocks is None
    used = _check_disjoint(obs, off)
    assert used.all()                                    # no padding without a dissection
    Ts = (int(off[-1]) + 63) // 64
    assert len(prof) == Ts and all(prof[c] >= c for c in range(Ts)) and all(np.diff(prof) >= 0) and prof[-1] == Ts - 1


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_dissected_layout_leaves_do_not_couple(seed):
    obs = _grid_job(np.random.default_rng(seed))
    off, prof, blocks = _layout(obs, 8)
    assert blocks is not None
    Ta, Tb, used_a, used_b = blocks
    TL = Ta + Tb
    used = _check_disjoint(obs, off)
    # the blocks start on tile boundaries, the padding sits at the leaves' ends only
    assert (Ta - 1) * 64 < used_a <= Ta * 64 and (Tb - 1) * 64 < used_b <= Tb * 64
    assert used[:used_a].all() and not used[used_a:Ta * 64].any()
    assert used[Ta * 64:Ta * 64 + used_b].all() and not used[Ta * 64 + used_b:TL * 64].any() and used[TL * 64:].all()
    # every slot of a leaf is seen from ONE side of a bipartition of the robots, the top block's slots from both
    side = {}
    for i, (cls, w) in enumerate(obs):
        blk = 0 if off[i] < Ta * 64 else (1 if off[i] < TL * 64 else 2)
        assert (off[i] + DIM[cls] - 1 < Ta * 64) if blk == 0 else True
        for r in w:
            if blk < 2:
                assert side.setdefault(r, blk) == blk, (r, w)
    a = {r for r, b in side.items() if b == 0}
    b = {r for r, s_ in side.items() if s_ == 1}
    assert a and b and not (a & b)
    for i, (cls, w) in enumerate(obs):
        if off[i] >= TL * 64:
            assert (set(w) & a) and (set(w) & b) or not (set(w) <= a or set(w) <= b)
    # the profile ends each leaf at its own last tile row (the leaves share no tile), the top block fills in
    Ts = (int(off[-1]) + 63) // 64
    assert len(prof) == Ts and prof[Ta - 1] == Ta - 1 and prof[TL - 1] == TL - 1 and all(prof[TL:] == Ts - 1)
    assert all(prof[c] <= Ta - 1 for c in range(Ta)) and all(prof[c] >= c for c in range(Ts)) and all(np.diff(prof) >= 0)


def test_forced_bipartition_follows_the_ranks_and_falls_back():
    obs = _grid_job(np.random.default_rng(4))
    off, prof, blocks = _layout(obs, 8, force_a={0, 1, 2, 3})
    assert blocks is not None
    Ta = blocks[0]
    for i, (cls, w) in enumerate(obs):
        if off[i] < Ta * 64:
            assert set(w) <= {0, 1, 2, 3}
    # two robots: every shared slot is seen from both — nothing to dissect, forced or not
    two = [(0, frozenset((0, 1)))] * 30
    assert _layout(two, 2, force_a={0})[2] is None and _layout(two, 2)[2] is None


def test_packed_exchange_segments_tile_the_buffer():
    import slide_slam_amd as s
    m, nrel, Ta, Tb = 3830, 2, 25, 23
    full = s.CholBatch.sep_buffer_len(m, nrel)
    cut = s.CholBatch.sep_exchange_len(m, nrel, Ta, Tb)
    assert cut == full - 4096 * Ta * Tb
    segs = [s.CholBatch.sep_segment(m, nrel, Ta, Tb, w) for w in range(3)]
    assert segs[0][0] == 0 and segs[1][0] == segs[0][1] and segs[2][0] == segs[0][1] + segs[1][1]
    assert sum(ln for _, ln in segs) == cut
    Tt = (m + 63) // 64 + 1                               # landmark tiles + one tile of lambda coordinates
    assert segs[0][1] == 4096 * sum(Tt + 1 - t - Tb for t in range(Ta))
    assert segs[2][1] == 4096 * sum(Tt + 1 - t for t in range(Ta + Tb, Tt))
