"""Robots sharded over GPUs: cross-robot landmark association and the distributed Gauss-Newton pass.

The reference keeps a FULL replica of every robot's graph in every `sloam_node` and gossips packets over ROS topics
(databaseManager.cpp:219-279; ingestion sloamNode.cpp:912-1002): its solve() optimises the JOINT graph.  Here robot r's poses, factors
and maps live on one GPU only (8 / N robots per GPU, one process per GPU, all robots of a process in one CholBatch driven by ONE host
thread: `PassDriver`); landmarks observed by several robots are replicated and kept identical.

A pass = one joint Gauss-Newton iteration.  Three variants, selected on the PassDriver:

  exact joint step (arrow=True, the default of bench.py)
      The shared landmarks are NOT eliminated into the robots' pose systems: they are the separator of the joint graph.  Every robot
      eliminates its private landmarks and its poses (banded Cholesky with the separator's coupling rows as a border), forms its Schur
      complement onto the separator with one FP64-MFMA product; ONE all-reduce(sum) of the packed separator system per pass; every rank
      factors it and substitutes back.  This is exactly the step the reference's replica takes with one solve().  Inter-robot
      relative-pose factors (addRelativeMeasFactor, graph.cpp:247-258) ride along exactly as six further separator coordinates each
      (setup_ghosts).
  PCG (pcg_iters > 0)
      phase 0 | AR 54/slot | phase 1 (global H_ll, Schur, factor own block) | { AR 9/slot | matvec | AR 2 | update } x iterations |
      AR 9/slot | phase 2: preconditioned conjugate gradients on the global reduced pose system with the robots' own factors as
      preconditioner; inexact (C3 needs ~200 iterations per pass for 1e-8), kept for comparison; pcg_tol ends it on the residual.
  block-Jacobi (neither)
      every robot solves its own block with the global landmark blocks; stalls once robots share many landmarks.

`shard` is any object with ``landmark_table(cls)``, ``graph.set_shared``, ``graph.dist_phase`` — the product's ``SlideBackend`` on a
GPU, or (tests only) the oracle wrapper, which restates every variant on the CPU; `comm` moves the exchange buffer.
`DistributedGraph` / `ThreadGroup` are the round-1 drivers (one thread per robot), kept with their tests.
"""
from __future__ import annotations

import numpy as np

CLASS_THRESH = {0: "cylinder_match_thresh", 1: "cuboid_match_thresh", 2: "ellipsoid_match_thresh"}


def gpu_matcher(cls, xyz, lab, gxyz, glab, thresh):
    """The cross-robot merge's matcher on the device: the stand-alone HIP matchers of the C-ABI (sloam::match*Models rule,
    sloam.cpp:73-203).  Cylinders exchange their roots only; vertical rays reproduce the matcher's point-at-height rule."""
    from . import api as s
    if cls == 0:
        n, m = len(lab), len(glab)
        return s.match_cylinders(xyz, np.tile([0.0, 0.0, 1.0], (n, 1)), lab, gxyz, np.tile([0.0, 0.0, 1.0], (m, 1)), glab, thresh)
    return s.match_boxes(cls, xyz, lab, gxyz, glab, thresh)


def associate_global(tables, thresh=(2.0, 2.0, 0.75), matcher=None):
    """Deterministic cross-robot landmark association, identical on every rank.

    tables[r][cls] = (xyz [n,3], label [n]).  Robots are merged in rank order (as a host replica ingests the
    other robots' packets): each landmark of robot r is matched against the global landmarks created by robots
    < r of the same class with the reference's matcher rule (nearest, strict '<', label gate for ellipsoids and
    cylinders) via `matcher(cls, xyz, label, map_xyz, map_label, thresh) -> index or -1`.
    Returns gid[r][cls] (global id per local landmark) and n_global[cls]."""
    R = len(tables)
    gid = [[None] * 3 for _ in range(R)]
    n_global = [0, 0, 0]
    for cls in range(3):
        gxyz = np.zeros((0, 3))
        glab = np.zeros(0, np.int32)
        for r in range(R):
            xyz, lab = tables[r][cls]
            n = len(lab)
            ids = np.full(n, -1, np.int64)
            if n and len(glab):
                m = matcher(cls, xyz, lab, gxyz, glab, thresh[cls])
                ids[:] = m
                # two local landmarks may not collapse onto the same global one: keep the first
                seen = set()
                for i in range(n):
                    if ids[i] >= 0:
                        if int(ids[i]) in seen:
                            ids[i] = -1
                        else:
                            seen.add(int(ids[i]))
            new = np.nonzero(ids < 0)[0]
            ids[new] = len(glab) + np.arange(len(new))
            gxyz = np.vstack([gxyz, xyz[new]]) if len(new) else gxyz
            glab = np.concatenate([glab, lab[new]]) if len(new) else glab
            gid[r][cls] = ids
        n_global[cls] = len(glab)
    return gid, n_global


def associate_by_ingest(own_ids, replica_ids):
    """Cross-robot association as the REFERENCE makes it (sloamNode.cpp:912-1002): a host replica ingests every robot's packets frame
    by frame and associates their detections against its own, still moving, maps.  own_ids[r][cls]: per key frame the landmark id
    robot r's OWN graph gave every detection; replica_ids[r][cls]: the ids the replica gave the same detections.  The co-occurrences
    define the global id of every local landmark.  Two things the sharded layout cannot express are counted and resolved first-come:
    `split` — the replica put detections of ONE local landmark on several of its landmarks; `collapsed` — it put two local
    landmarks of one robot on the same one (the second gets an id of its own).  Returns (gid, n_global, stats) as associate_global does."""
    R = len(own_ids)
    gid = [[None] * 3 for _ in range(R)]
    n_global = [0, 0, 0]
    stats = dict(split=0, collapsed=0)
    for cls in range(3):
        nglob = 0
        for r in range(R):
            for ids in replica_ids[r][cls]:
                if len(ids):
                    nglob = max(nglob, int(np.max(ids)) + 1)
        extra = nglob
        for r in range(R):
            nloc = 0
            for ids in own_ids[r][cls]:
                if len(ids):
                    nloc = max(nloc, int(np.max(ids)) + 1)
            g = np.full(nloc, -1, np.int64)
            for a, b in zip(own_ids[r][cls], replica_ids[r][cls]):
                for lo, gl in zip(a, b):
                    lo, gl = int(lo), int(gl)
                    if lo < 0 or gl < 0:
                        continue
                    if g[lo] < 0:
                        g[lo] = gl
                    elif g[lo] != gl:
                        stats["split"] += 1
            seen = set()
            for lo in range(nloc):
                if g[lo] < 0 or int(g[lo]) in seen:
                    stats["collapsed"] += int(g[lo] >= 0)
                    g[lo] = extra
                    extra += 1
                else:
                    seen.add(int(g[lo]))
            gid[r][cls] = g
        n_global[cls] = extra
    return gid, n_global, stats


def shared_slots(gid, n_global, rank):
    """Slot table of this rank: slots enumerate (cls, global id) pairs observed by >= 2 robots, in a fixed
    global order; the owner is the lowest rank observing the landmark."""
    R = len(gid)
    cls_out, idx_out, own_out = [], [], []
    for cls in range(3):
        count = np.zeros(n_global[cls], np.int32)
        first = np.full(n_global[cls], -1, np.int32)
        for r in range(R):
            g = gid[r][cls]
            count[g] += 1
            for gg in g:
                if first[gg] < 0:
                    first[gg] = r
        mine = {int(g): i for i, g in enumerate(gid[rank][cls])}
        for g in np.nonzero(count >= 2)[0]:
            if int(g) in mine:
                cls_out.append(cls); idx_out.append(mine[int(g)]); own_out.append(int(first[g] == rank))
            else:
                cls_out.append(-1); idx_out.append(0); own_out.append(0)
    return np.array(cls_out, np.int32), np.array(idx_out, np.int64), np.array(own_out, np.int32)


SLOT_DIM = {0: 7, 1: 9, 2: 3}        # tangent dimension per class: cylinder [ray, root, radius], cube [pose, scale], point


def slot_observers(gid, n_global):
    """Observer robots of every shared slot, in shared_slots' slot order (by class, then by global id)."""
    R = len(gid)
    obs = []
    for cls in range(3):
        who = [[] for _ in range(n_global[cls])]
        for r in range(R):
            for g in gid[r][cls]:
                who[int(g)].append(r)
        obs += [(cls, tuple(w)) for w in who if len(w) >= 2]
    return obs


def _cuthill_mckee(adj, nodes):
    """Cuthill-McKee order of `nodes` (every connected component from its lowest-degree node, neighbours by increasing degree)."""
    nodes = set(nodes)
    pos, order = {}, []
    for start in sorted(nodes, key=lambda r: (len(adj[r] & nodes), r)):
        if start in pos:
            continue
        queue = [start]
        pos[start] = len(order); order.append(start)
        while queue:
            r = queue.pop(0)
            for x in sorted(adj[r] & nodes, key=lambda q: (len(adj[q] & nodes), q)):
                if x not in pos:
                    pos[x] = len(order); order.append(x); queue.append(x)
    return pos


def _layout_block(obs, members, adj, start, off, tile):
    """Lay the slots `members` out from coordinate `start` (a tile boundary) along a Cuthill-McKee order of their observers; returns
    (used coordinates, tile profile of the block in absolute tile rows).  Coordinate g reaches the last coordinate any of its
    observers observes INSIDE the block."""
    robots = sorted({r for i in members for r in obs[i][1]})
    pos = _cuthill_mckee(adj, robots)
    keys = sorted((min(pos[r] for r in obs[i][1]), max(pos[r] for r in obs[i][1]), i) for i in members)
    o = start
    for _, _, i in keys:
        off[i] = o
        o += SLOT_DIM[obs[i][0]]
    used = o - start
    nt = (used + tile - 1) // tile
    t0 = start // tile
    last = {}
    for i in members:
        for r in obs[i][1]:
            last[r] = max(last.get(r, 0), off[i] + SLOT_DIM[obs[i][0]] - 1)
    prof = np.arange(t0, t0 + nt, dtype=np.int64)
    for i in members:
        reach = max(last[r] for r in obs[i][1]) // tile
        for t in range(off[i] // tile, (off[i] + SLOT_DIM[obs[i][0]] - 1) // tile + 1):
            prof[t - t0] = max(prof[t - t0], reach)
    return used, np.maximum.accumulate(prof)


def separator_offsets(gid, n_global, tile=64, dissect=True, force_a=None):
    """Layout of the separator system of the exact joint step, the same on every rank: off[slot] = offset of the slot's tangent
    coordinates (cylinder 7, cube 9, point 3), off[n_slots] = the dimension m; prof = its tile-level profile (prof[c] = last tile row
    of tile column c that can be non-zero); blocks = None or (Ta, Tb, used_a, used_b).  Two shared landmarks couple in the separator
    system only if some robot observes both, so the slots are laid out along a Cuthill-McKee order of the robots' adjacency graph
    (robots adjacent = they share a slot).  With `dissect` the robots are first split into two sets (every bipartition tried; the one
    with the shortest chain of block columns max(leaf a, leaf b) + top wins, if it beats the plain layout by 15 %): the slots seen
    only from one set form that set's LEAF block, the slots seen from both the TOP block behind them; leaf blocks start at tile
    boundaries (the coordinates between a leaf's last slot and the next tile boundary are padding no slot uses — m counts them).  The
    leaves do not couple, so slide_chol_batch_set_separator_blocks factors them side by side.  Slot order itself (shared_slots) is
    unchanged: only the coordinates are permuted.  force_a: a set of robots that must be the first set (a job whose ranks split in two
    halves along the dissection: slide_chol_batch_set_separator_owner) — taken if both leaves and the top block are non-empty."""
    obs = slot_observers(gid, n_global)
    R = len(gid)
    adj = [set() for _ in range(R)]
    for _, w in obs:
        for a in w:
            adj[a].update(x for x in w if x != a)
    n = len(obs)
    off = np.zeros(n + 1, np.int64)
    dims = np.array([SLOT_DIM[c] for c, _ in obs], np.int64)
    total = int(dims.sum())
    tiles = lambda d: (d + tile - 1) // tile
    best = None
    if dissect and 3 <= R <= 16 and n > 0:
        masks = [sum(1 << r for r in w) for _, w in obs]
        mk = np.array(masks, np.int64)
        # (slots with the same observer set move together: the search works on the distinct sets' summed dimensions)
        umask, inv = np.unique(mk, return_inverse=True)
        udim = np.bincount(inv, weights=dims.astype(np.float64), minlength=len(umask)).astype(np.int64)
        forced = sum(1 << r for r in force_a) if force_a else 0
        for a_mask in ([forced] if forced else range(1, 1 << (R - 1))):           # robot R-1 always on the b side: every bipartition once
            da = int(udim[(umask & ~a_mask) == 0].sum())
            db = int(udim[(umask & a_mask) == 0].sum())
            if da == 0 or db == 0:
                continue
            dt = total - da - db
            if dt == 0:
                continue
            cost = max(tiles(da), tiles(db)) + tiles(dt)
            if best is None or cost < best[0]:
                best = (cost, a_mask)
        if best is not None and best[0] > 0.85 * tiles(total) and not forced:
            best = None
    if best is None:
        used, prof = _layout_block(obs, list(range(n)), adj, 0, off, tile)
        off[n] = used
        return off.astype(np.int32), prof.astype(np.int32), None
    a_mask = best[1]
    masks = [sum(1 << r for r in w) for _, w in obs]
    in_a = [i for i in range(n) if masks[i] & ~a_mask == 0]
    in_b = [i for i in range(n) if masks[i] & a_mask == 0]
    top = [i for i in range(n) if i not in set(in_a) and i not in set(in_b)]
    used_a, prof_a = _layout_block(obs, in_a, adj, 0, off, tile)
    Ta = tiles(used_a)
    used_b, prof_b = _layout_block(obs, in_b, adj, Ta * tile, off, tile)
    Tb = tiles(used_b)
    used_t, prof_t = _layout_block(obs, top, adj, (Ta + Tb) * tile, off, tile)
    off[n] = (Ta + Tb) * tile + used_t
    Ts = Ta + Tb + tiles(used_t)
    prof = np.concatenate([prof_a, prof_b, np.full(tiles(used_t), Ts - 1, np.int64)])      # (the top block fills in: dense)
    return off.astype(np.int32), prof.astype(np.int32), (int(Ta), int(Tb), int(used_a), int(used_b))


class DistributedGraph:
    def __init__(self, shard, comm, rank, world):
        self.shard, self.comm, self.rank, self.world = shard, comm, rank, world
        self.n_slots = 0

    def setup(self, matcher, thresh=(2.0, 2.0, 0.75)):
        """All-gather the landmark tables, associate them globally, install the shared slots and adopt the
        owners' values for every shared landmark."""
        mine = [self.shard.landmark_table(cls) for cls in range(3)]
        tables = self.comm.all_gather_object(mine)
        gid, n_global = associate_global(tables, thresh, matcher)
        cls, idx, own = shared_slots(gid, n_global, self.rank)
        self.n_slots = len(cls)
        self.shard.graph.set_shared(cls, idx, own)
        self.buf = self.comm.alloc(max(self.n_slots, 1) * 54)
        self._phase(10)
        self.comm.all_reduce(self.buf, self.n_slots * 15)
        self._phase(11)
        return dict(n_slots=self.n_slots, n_global=n_global)

    def setup_ghosts(self, relmeas, local_robot=0):
        """relmeas: the job's inter-robot measurements [(pose index k, robot a, robot b, rel7 a->b)] (or with a fifth entry: the
        other robot's pose index), identical on every rank.  Ghost slots = the sorted (robot, k) pairs they touch; this rank adds the factors that involve its robot."""
        from .synth import relmeas_keys
        rm = [relmeas_keys(e) for e in relmeas]          # (ka, a, kb, b, rel7)
        keys = sorted({(a, ka) for (ka, a, kb, b, _) in rm} | {(b, kb) for (ka, a, kb, b, _) in rm})
        slot = {key: i for i, key in enumerate(keys)}
        self.n_gslots = len(keys)
        own_robot = np.array([local_robot if r == self.rank else -1 for (r, _) in keys], np.int32)
        own_idx = np.array([k for (_, k) in keys], np.int64)
        self.shard.graph.set_ghosts(own_robot, own_idx)
        for (ka, a, kb, b, rel) in rm:
            if a == self.rank:
                self.shard.graph.add_relative_meas_ghost(rel, ka, local_robot, slot[(b, kb)], True)
            if b == self.rank:
                self.shard.graph.add_relative_meas_ghost(rel, kb, local_robot, slot[(a, ka)], False)
        self.gbuf = self.comm.alloc(max(self.n_gslots, 1) * 12)
        return self.n_gslots

    def _phase(self, ph):
        self.shard.graph.dist_phase(ph, self.comm.handle(self.buf))

    def gauss_newton(self, iterations=1):
        if getattr(self, "local_batch", False) and not getattr(self, "n_gslots", 0):
            # every robot of the job sits in one CholBatch on this GPU: exchanges are device-side sums, no host barrier per phase
            for _ in range(iterations):
                self.shard.graph.dist_pass_local(self.comm.handle(self.buf))
            return
        for _ in range(iterations):
            if getattr(self, "n_gslots", 0):
                self.shard.graph.dist_phase(20, self.comm.handle(self.gbuf))
                self.comm.all_reduce(self.gbuf, self.n_gslots * 12)
                self.shard.graph.dist_phase(21, self.comm.handle(self.gbuf))
            self._phase(0)
            self.comm.all_reduce(self.buf, self.n_slots * 54)
            self._phase(1)
            self.comm.all_reduce(self.buf, self.n_slots * 9)
            self._phase(2)


def setup_local_shards(shards, matcher, base=None, rank=0, world=1, thresh=(2.0, 2.0, 0.75), device=None, assoc=None):
    """Cross-robot association for ALL robot shards of this process from one host thread (virtual rank of local shard t =
    rank * len(shards) + t; every process holds the same number of shards): all-gather the landmark tables, run the
    deterministic merge, install the shared slots of every local shard and adopt the owners' values.  `base` (a TorchComm) joins
    the processes of a multi-GPU job; None = single process.  Returns (exchange buffers [one per shard], info).
    assoc = (gid, n_global) of ALL the job's robots (associate_by_ingest: the reference's frame-by-frame association of a host
    replica) replaces the merge of the final maps (mode "ingest"; single process or identical on every rank)."""
    R = len(shards)
    mine = [[sh.landmark_table(cls) for cls in range(3)] for sh in shards]
    parts = base.all_gather_object(mine) if (base is not None and world > 1) else [mine]
    tables = [t for part in parts for t in part]
    if assoc is not None:
        gid, n_global = assoc
        for r, tb in enumerate(tables):
            for cls in range(3):
                if len(gid[r][cls]) != len(tb[cls][1]):
                    raise ValueError(f"ingest association: robot {r} class {cls} holds {len(tb[cls][1])} landmarks, the id table {len(gid[r][cls])}")
    else:
        gid, n_global = associate_global(tables, thresh, matcher)
    bufs, n_slots = [], 0
    if device is not None:
        import torch
        alloc = lambda n: _device_zeros(torch, n, device)
        handle = lambda b: b.data_ptr()
    else:
        alloc = lambda n: np.zeros(n)
        handle = lambda b: b
    import os
    dissect = os.environ.get("SLIDE_SEP_DISSECT", "1") != "0"
    owner = None
    sep_off = None
    n_robots = world * R
    # The dissection is the SAME at every rank count: the first half of the job's robots against the second (on the 2 x 4 grid of C4: its
    # two rows — also the best of all bipartitions), so that the sums of a job are the same bits at 1, 2, 4 and 8 ranks (a whole pass on
    # one GPU takes per-half partial sums too: CholBatch::enqueue_arrow).  SLIDE_SEP_FREE_SPLIT=1: a single process searches all
    # bipartitions instead (the shortest chain; not comparable bit for bit with a job of several ranks).
    if dissect and n_robots >= 4 and n_robots % 2 == 0 and not (world == 1 and os.environ.get("SLIDE_SEP_FREE_SPLIT") == "1"):
        sep_off, sep_prof, sep_blocks = separator_offsets(gid, n_global, dissect=True, force_a=set(range(n_robots // 2)))
        if sep_blocks is None:
            sep_off = None
        elif world >= 2 and world % 2 == 0 and device is not None and os.environ.get("SLIDE_SEP_OWNED", "1") != "0":
            # the ranks split in two halves along the dissection: every rank then factors one leaf only and only the top block crosses
            # between the halves (slide_chol_batch_set_separator_owner); every rank of a half computes the half's result (no leader:
            # the exchanges are pairwise, PassDriver._sep_exchange)
            half = 0 if rank < world // 2 else 1
            owner = dict(leaf=half, leader=True, rank=rank, half_ranks=list(range(half * (world // 2), (half + 1) * (world // 2))))
    if sep_off is None:
        sep_off, sep_prof, sep_blocks = separator_offsets(gid, n_global, dissect=dissect and world == 1 and os.environ.get("SLIDE_SEP_FREE_SPLIT") == "1")
    if sep_blocks is not None:
        sep_prof = (sep_prof, sep_blocks, owner)        # (travels with the profile to PassDriver: the dissection is a property of the layout)
    for t, sh in enumerate(shards):
        cls, idx, own = shared_slots(gid, n_global, rank * R + t)
        n_slots = len(cls)
        sh.graph.set_shared(cls, idx, own)
        if hasattr(sh.graph, "set_separator"):
            sh.graph.set_separator(sep_off)
        bufs.append(alloc(max(n_slots, 1) * 54))
    # value broadcast: every replica of a shared landmark adopts its owner's value (phase 10, all-reduce(sum) of 15 / slot, phase 11)
    for t, sh in enumerate(shards):
        sh.graph.dist_phase(10, handle(bufs[t]))          # (returns after its kernels have completed)
    if n_slots:
        n15 = n_slots * 15
        if device is None:
            total = np.sum([b[:n15] for b in bufs], axis=0)
        else:
            total = torch.stack([b[:n15] for b in bufs]).sum(0)
        if base is not None and world > 1:
            base.all_reduce(total, n15)
        for b in bufs:
            b[:n15] = total
        if device is not None:      # (torch's stream only: a device-wide wait would invalidate a capture another rank thread has going)
            torch.cuda.current_stream(device).synchronize()
    for t, sh in enumerate(shards):
        sh.graph.dist_phase(11, handle(bufs[t]))
    return bufs, dict(n_slots=n_slots, n_global=n_global, sep_dim=int(sep_off[-1]), sep_off=sep_off, sep_prof=sep_prof)


def _device_zeros(torch, n, device):
    """n zero doubles on the device, COMPLETE when this returns.  torch.zeros only enqueues the fill on torch's current stream; the
    buffer's first writers are kernels of non-blocking HIP streams, which nothing orders behind that stream — a fill that is still
    queued (the streams of several rank threads share hardware queues) would land on top of what they wrote."""
    buf = torch.zeros(n, dtype=torch.float64, device=device)
    torch.cuda.current_stream(device).synchronize()
    return buf


class PassDriver:
    """One distributed Gauss-Newton pass of all robot shards of this process, driven by ONE host thread.

    With a CholBatch (`batch`, HIP shards): the pass is the batch's captured launch sequences — the whole pass as one hipGraph when
    the job is this process alone, or its three parts with the two cross-GPU all-reduces of buffer 0 issued ON THE BATCH'S STREAM
    between them (RCCL through torch.distributed on an ExternalStream: stream-ordered, no host synchronisation until the end of the
    pass).  Without one (oracle shards on the CPU, or un-batched HIP shards): the same sequence spelled out with dist_phase calls
    and host-side sums — the CPU rehearsal of exactly this control flow."""

    def __init__(self, shards, bufs, n_slots, batch=None, base=None, world=1, device=None, pcg_iters=0, pcg_tol=0.0, arrow=False, sep_dim=0, sep_prof=None):
        self.shards, self.bufs, self.n_slots, self.batch, self.base, self.world, self.device = shards, bufs, n_slots, batch, base, world, device
        self.ptrs = [b.data_ptr() for b in bufs] if device is not None else None
        if batch is not None and base is not None and hasattr(base, "bind_stream"):
            base.bind_stream(batch.stream())      # (ranks as threads: their collectives wait for this stream, not for the device)
        self.passes = 0
        # Collectives of a cut pass: stream-ordered = issued under torch's ExternalStream of the batch's HIP stream (no host
        # synchronisation until the end of the pass).  That path has run with ONE RCCL rank only (no multi-GPU node was available), so a
        # job that spans processes defaults to host-synchronous collectives — one synchronisation per exchange, i.e. one per pass of the
        # exact joint step — unless SLIDE_STREAM_ORDERED=1 asks for the stream-ordered ones (bench.py reports which ran).
        import os
        self.stream_ordered = world == 1 or os.environ.get("SLIDE_STREAM_ORDERED") == "1"
        self.force_parts = False        # True (rehearsal): the cut pass + collectives even when the job is this process alone
        # joint solve: PCG on the global reduced system after the factorisations (0: block-Jacobi over robots).  pcg_iters is the
        # upper bound per pass; with pcg_tol > 0 the solve ends as soon as sqrt(r^T M^-1 r) has fallen to pcg_tol times its first
        # value (every rank sees the same all-reduced scalars, so every rank takes the same decision).
        self.pcg_iters = pcg_iters if n_slots > 0 else 0
        self.pcg_tol = float(pcg_tol)
        self.pcg_history = []           # iterations that did work, per pass
        # EXACT joint step ("arrow"): the shared landmarks stay as the separator of the joint graph; every robot eliminates its
        # private landmarks and poses, ONE all-reduce(sum) of the separator system (sep_dim^2 + sep_dim doubles), every rank solves it
        # and substitutes back — the Gauss-Newton step of the reference's full replica, no inner iteration.
        self.arrow = bool(arrow) and n_slots > 0
        self.sep_dim = int(sep_dim)
        self.sbufs = None
        self.sep = None
        self._sep_segs = None
        self.sep_owner = None
        self._pair_groups = None
        if self.arrow:
            self.pcg_iters = 0
            if device is None:
                m = self.sep_dim
                self.sbufs = [np.zeros(m * m + 2 * m) for _ in shards]
            elif batch is None:
                raise ValueError("the exact joint step of HIP shards runs as a batched pass: join the shards to a CholBatch")
        if batch is not None:
            batch.set_exact_joint(self.arrow, 0, 0)      # (the exchange buffer of a cut pass is installed on first use: _sep_exchange_buffer)
            if self.arrow and sep_prof is not None:
                blocks, owner = None, None
                if isinstance(sep_prof, tuple):
                    sep_prof, blocks, owner = (tuple(sep_prof) + (None,))[:3]
                self.sep_owner = owner
                batch.set_separator_profile(sep_prof)    # (tile profile of the separator system: the same on every rank)
                batch.set_separator_blocks(*(blocks if blocks is not None else (0, 0, 0, 0)))
                batch.set_separator_owner(owner["leaf"] if owner else -1, owner["leader"] if owner else True)
                self.sep_blocks = blocks
            if self.arrow and world > 1 and base is not None:
                # every rank must have taken the same layout decisions (per-rank SLIDE_SEP_OWNED / SLIDE_SEP_DISSECT settings would hang the
                # job inside the first pass instead of failing here), and the halves' groups are created NOW, by every rank, in the same
                # order (they used to be created lazily inside the first pass)
                mine = (int(self.sep_dim), int(n_slots), tuple(int(v) for v in (getattr(self, "sep_blocks", None) or ())),
                        None if self.sep_owner is None else (int(self.sep_owner["leaf"]), tuple(self.sep_owner["half_ranks"])))
                seen = base.all_gather_object(mine)
                lay = {(m[0], m[1], m[2], m[3] is None) for m in seen}
                if len(lay) != 1:
                    raise RuntimeError(f"exact joint step: the ranks disagree on the separator layout (dimension, slots, blocks, owned): {sorted(map(str, lay))}")
                if self.sep_owner is not None:
                    # PAIRWISE exchanges only (a sum of two operands is the same bits on both sides and in either order): round j inside a
                    # half pairs rank r with r ^ (1 << j) — with the robots laid out rank-major that is the binary tree over the robot index
                    # which k_sep_gather sums along on every GPU — and one more round pairs the halves.  Every rank creates every group,
                    # in the same order.
                    hs = len(self.sep_owner["half_ranks"])
                    assert hs & (hs - 1) == 0, "the ranks of a half must be a power of two"
                    self._pair_groups = []
                    j = 1
                    while j < world:
                        mine = None
                        for r in range(world):
                            if not r & j:
                                g = base.new_group((r, r | j))
                                if self.sep_owner["rank"] in (r, r | j):
                                    mine = g
                        self._pair_groups.append(mine)
                        j <<= 1
            import os
            # nested dissection of the robots' own bands (slide_chol_batch_set_segments; every segment carries only the border rows that
            # are non-zero in it).  Measured on one MI355X: 8 robots x 625 poses 3.06 ms per pass uncut, 2.64 / 2.54 / 2.60 with
            # 2 / 3 / 4 segments; 2 robots x 500 poses 1.37 uncut, 1.12 / 1.08 / 1.04 / 1.04 with 2 / 3 / 4 / 6 — the fewer robots share
            # the GPU, the more segments fit side by side
            seg = os.environ.get("SLIDE_SEGMENTS")
            # (the count is the same whatever the number of robots on this GPU — 4 segments were 4 % faster with few robots —: a job's
            # arithmetic must not depend on how its robots are spread over ranks)
            batch.set_segments((int(seg) if seg else 3) if self.arrow else 1)
        if batch is not None:           # (always pushed, zero included: a batch or graph may still hold an earlier driver's setting)
            batch.set_pcg(self.pcg_iters, self.pcg_tol)
        else:
            for sh in shards:
                sh.graph.set_pcg(self.pcg_iters, self.pcg_tol)

    def setup_ghosts(self, relmeas, rank=0):
        """Inter-robot relative-pose measurements of the job, [(pose index k, robot a, robot b, rel7 a->b)], identical on every rank
        (addRelativeMeasFactor, graph.cpp:247-258).  Both robots hold the factor, each with its own pose as the variable and the other
        pose as a GHOST — its current estimate, refreshed at the start of every pass (12 doubles per ghost slot, one small all-reduce),
        at which the factor is linearised.  Exact joint step: the factor's six linearised residuals join the separator ("lambda"
        coordinates), each robot couples to them through its own Jacobian, and the step is exactly the joint replica's.  PCG / block-Jacobi
        passes: the cross block J_a^T J_b is left out of the step (gradient exact).  Virtual robot of local shard t = rank * R + t."""
        from .synth import relmeas_keys
        R = len(self.shards)
        rm = [relmeas_keys(e) for e in relmeas]          # (ka, a, kb, b, rel7): pose ka of robot a sees pose kb of robot b (ka == kb for 4-tuples)
        keys = sorted({(a, ka) for (ka, a, kb, b, _) in rm} | {(b, kb) for (ka, a, kb, b, _) in rm})
        slot = {key: i for i, key in enumerate(keys)}
        self.n_gslots = len(keys)
        self.n_relmeas = len(rm)
        for t, sh in enumerate(self.shards):
            v = rank * R + t
            sh.graph.set_ghosts(np.array([0 if r == v else -1 for (r, _) in keys], np.int32), np.array([k for (_, k) in keys], np.int64))
            ids = []
            for i, (ka, a, kb, b, rel) in enumerate(rm):
                if a == v:
                    sh.graph.add_relative_meas_ghost(rel, ka, 0, slot[(b, kb)], True)
                    ids.append(i)
                if b == v:
                    sh.graph.add_relative_meas_ghost(rel, kb, 0, slot[(a, ka)], False)
                    ids.append(i)
            if self.arrow:
                # exact joint step: the factor enters through six separator coordinates of its own (its linearised residual), coupled to
                # each robot's band by that robot's own Jacobian — the step stays exactly the joint replica's
                sh.graph.set_ghost_ids(np.array(ids, np.int32), len(relmeas))
        n12 = 12 * self.n_gslots
        if self.arrow:
            self.lam_dim = 6 * len(relmeas)
            if self.device is None:
                m = self.sep_dim + self.lam_dim
                self.sbufs = [np.zeros(m * m + 2 * m) for _ in self.shards]
        if self.device is None:
            self.gbufs = [np.zeros(max(n12, 1)) for _ in self.shards]
        else:
            # (the batched pass runs the ghost exchange through the shards' exchange buffers: they must hold 12 doubles per ghost slot)
            if any(b.numel() < n12 for b in self.bufs):
                raise ValueError("setup_ghosts: the exchange buffers are smaller than 12 doubles per ghost slot")
        return self.n_gslots

    def _ghost_refresh_host(self):
        """CPU shards: dist_phase 20 (pack the owned ghost poses' estimates), sum, dist_phase 21 (adopt)."""
        n12 = 12 * self.n_gslots
        for sh, b in zip(self.shards, self.gbufs):
            sh.graph.dist_phase(20, b)
        tot = self.gbufs[0]
        for b in self.gbufs[1:]:
            tot[:n12] += b[:n12]
        if self.world > 1:
            self.base.all_reduce(tot, n12)
        for sh in self.shards:
            sh.graph.dist_phase(21, tot)

    def _sep_exchange_buffer(self):
        """The separator system's exchange buffer of a cut pass (packed lower tile columns): this GPU's partial sum after part 0,
        all-reduced across the GPUs on the batch's stream, read back by part 2."""
        if self.sep is None:
            import torch
            self.sep_len = self.batch.sep_buffer_len(self.sep_dim, getattr(self, "n_relmeas", 0))
            self.sep = _device_zeros(torch, max(self.sep_len, 1), self.device)
            self.batch.set_exact_joint(True, self.sep.data_ptr(), self.sep_len)
            blocks = getattr(self, "sep_blocks", None)
            if blocks is not None:      # (a dissected layout: the zero block between the leaves is not part of the packed exchange)
                self.sep_len = self.batch.sep_exchange_len(self.sep_dim, getattr(self, "n_relmeas", 0), blocks[0], blocks[1])
        return self.sep

    def _sep_exchange(self, sep, stream, times=None):
        """The separator system's exchange(s) of a cut pass between part 0 and part 2.  Plain: ONE all-reduce of the packed system.  A
        rank that owns a leaf (the ranks split in two halves along the dissection): all-reduce of the own leaf's segment within the
        own half (nothing when the half is this rank alone), part 1 (the leaf is factored, its Schur complement joins the top block),
        all-reduce of the top block's segment over all ranks."""
        own = getattr(self, "sep_owner", None)
        if own is None:
            if self.world > 1 or self.base is not None:
                self.base.all_reduce_on(sep, self.sep_len, stream)
            return
        if self._sep_segs is None:
            nr = getattr(self, "n_relmeas", 0)
            Ta, Tb = self.sep_blocks[0], self.sep_blocks[1]
            self._sep_segs = [self.batch.sep_segment(self.sep_dim, nr, Ta, Tb, w) for w in range(3)]
        def lap(key):      # (timed_cut_pass: a device synchronisation after every step)
            if times is not None:
                import time
                import torch
                torch.cuda.synchronize()
                now = time.perf_counter()
                times[key] = (now - times["_t"]) * 1e3
                times["_t"] = now
        # inside the half: the own leaf's segment AND the top block's, round by round (after log2(half) rounds every rank of the half holds
        # the half's sums — the same bits on all of them)
        n_in = len(self._pair_groups) - 1
        loff, lln = self._sep_segs[own["leaf"]]
        toff, tln = self._sep_segs[2]
        for j in range(n_in):
            self.base.all_reduce_on(sep, lln, stream, off=loff, group=self._pair_groups[j])
            self.base.all_reduce_on(sep, tln, stream, off=toff, group=self._pair_groups[j])
        lap("exchange_leaf_ms")
        self.batch.pass_part(self.ptrs, 1)          # every rank: the leaf factored, its Schur complement onto the half's top block
        lap("part1_ms")
        self.base.all_reduce_on(sep, tln, stream, off=toff, group=self._pair_groups[-1])      # between the halves: (A - S_a) + (B - S_b)
        lap("exchange_top_ms")

    def _exchange(self, count):
        """all-reduce(sum) of buffer 0's first `count` doubles across the processes, ordered behind the batch's stream."""
        if (self.world > 1 or (self.force_parts and self.base is not None)) and count:
            self.base.all_reduce_on(self.bufs[0], count, self.batch.stream() if (self.batch is not None and self.stream_ordered) else None)

    def _local_sum(self, count):
        if not count:
            return
        tot = self.bufs[0][:count].copy() if self.device is None else self.bufs[0][:count].clone()
        for b in self.bufs[1:]:
            tot += b[:count]
        self.bufs[0][:count] = tot

    def _local_bcast(self, count):
        for b in self.bufs[1:]:
            b[:count] = self.bufs[0][:count]
        if self.device is not None:          # (torch's stream; the shards' own streams read the buffers next)
            import torch
            torch.cuda.current_stream(self.device).synchronize()

    def _all(self, count):
        self._local_sum(count); self._exchange(count); self._local_bcast(count)

    def one_pass(self):
        n54, n9, K = 54 * self.n_slots, 9 * self.n_slots, self.pcg_iters
        if self.batch is not None:
            if self.world == 1 and not self.force_parts:
                self.batch.pass_all(self.ptrs)
            elif self.arrow:
                sep = self._sep_exchange_buffer()
                if getattr(self, "n_gslots", 0):
                    self.batch.pass_part(self.ptrs, 20)
                    self._exchange(12 * self.n_gslots)
                self.batch.pass_part(self.ptrs, 0)
                self._sep_exchange(sep, self.batch.stream() if self.stream_ordered else None)
                self.batch.pass_part(self.ptrs, 2)
            else:
                if getattr(self, "n_gslots", 0):
                    self.batch.pass_part(self.ptrs, 20)
                    self._exchange(12 * self.n_gslots)
                self.batch.pass_part(self.ptrs, 0)
                self._exchange(n54)
                self.batch.pass_part(self.ptrs, 1)
                for it in range(K):
                    self._exchange(n9)
                    self.batch.pass_part(self.ptrs, 10)
                    self._exchange(2)
                    self.batch.pass_part(self.ptrs, 12 if it == K - 1 else 11)
                self._exchange(n9)
                self.batch.pass_part(self.ptrs, 2)
        else:
            h = (lambda b: b.data_ptr()) if self.device is not None else (lambda b: b)

            def each(ph):
                for sh, b in zip(self.shards, self.bufs):
                    sh.graph.dist_phase(ph, h(b))
            if getattr(self, "n_gslots", 0):
                if self.device is None:
                    self._ghost_refresh_host()
                else:
                    # un-batched HIP shards: the same 20 | all-reduce of 12 doubles per ghost slot | 21 sequence through the shards'
                    # exchange buffers (these passes used to linearise the relative-pose factors at the stale first ghosts)
                    each(20)
                    self._all(12 * self.n_gslots)
                    each(21)
            each(0)
            if self.arrow:
                m = self.sep_dim + getattr(self, "lam_dim", 0)
                for sh, b in zip(self.shards, self.sbufs):
                    sh.graph.dist_phase(40, b)
                tot = self.sbufs[0]
                for b in self.sbufs[1:]:
                    tot[:m * m + m] += b[:m * m + m]
                if self.world > 1:
                    self.base.all_reduce(tot, m * m + m)
                self.shards[0].graph.dist_phase(41, tot)         # (factors and solves the separator, leaves the solution in the buffer)
                for sh in self.shards[1:]:
                    sh.graph.dist_phase(42, tot)
                self.passes += 1
                return
            self._all(n54)
            each(1)
            gamma0, done = None, 0
            for it in range(K):
                self._all(n9)
                each(31)
                self._all(2)
                gamma = float(self.bufs[0][0])
                if gamma0 is None:
                    gamma0 = gamma
                conv = (not gamma > 0.0) or (self.pcg_tol > 0.0 and gamma <= self.pcg_tol ** 2 * gamma0)
                done += 0 if conv else 1
                last = conv or it == K - 1
                each(33 if last else 32)
                if last:
                    break
            if K:
                self.pcg_history.append(done)
            self._all(n9)
            each(2)
        self.passes += 1

    def gauss_newton(self, iterations=1):
        for _ in range(iterations):
            self.one_pass()

    def timed_cut_pass(self):
        """One exact joint pass cut at its exchange with a device synchronisation after every part: {part0_ms, exchange_ms, part2_ms,
        exchange_bytes} — separates the collective's latency from the kernels' (diagnostic, N > 1 or force_parts)."""
        import time
        import torch
        if not (self.arrow and self.batch is not None):
            raise ValueError("timed_cut_pass: exact joint passes of a CholBatch only")
        sep = self._sep_exchange_buffer()
        out = {}
        if getattr(self, "n_gslots", 0):
            self.batch.pass_part(self.ptrs, 20)
            self._exchange(12 * self.n_gslots)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        self.batch.pass_part(self.ptrs, 0)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        times = {"_t": t1} if getattr(self, "sep_owner", None) is not None else None
        self._sep_exchange(sep, None, times)      # (a rank that owns a leaf runs part 1 in here: its own entry below)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if times:
            times.pop("_t")
            out.update(times)
        self.batch.pass_part(self.ptrs, 2)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        self.passes += 1
        own = getattr(self, "sep_owner", None)
        n_in = 0 if own is None else len(self._pair_groups) - 1
        xb = 8 * self.sep_len if own is None else 8 * (n_in * (self._sep_segs[own["leaf"]][1] + self._sep_segs[2][1]) + self._sep_segs[2][1])
        out.update(part0_ms=(t1 - t0) * 1e3, exchange_ms=(t2 - t1) * 1e3, part2_ms=(t3 - t2) * 1e3, exchange_bytes=xb)
        return out


class TorchComm:
    """torch.distributed plumbing: `nccl` (= RCCL over xGMI) with device buffers, or `gloo` with host buffers
    (CPU tests; also lets several ranks share one GPU by staging through the host)."""

    def __init__(self, device=None, stage_through_host=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.device = device
        self.stage = stage_through_host
        self._ext = {}

    def all_gather_object(self, obj):
        out = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(out, obj)
        return out

    def alloc(self, n):
        if self.device is None:
            return np.zeros(n)
        return _device_zeros(self.torch, n, self.device)

    def handle(self, buf):
        return buf if self.device is None else buf.data_ptr()

    def new_group(self, ranks):
        """A sub-group of the job's ranks (every rank of the job must make the same calls in the same order)."""
        return self.dist.new_group(ranks=list(ranks))

    def all_reduce(self, buf, n, off=0, group=None):
        if n == 0:
            return
        if self.device is None:
            t = self.torch.from_numpy(buf[off:off + n])
            self.dist.all_reduce(t, group=group)
        elif self.stage:
            h = buf[off:off + n].cpu()
            self.dist.all_reduce(h, group=group)
            buf[off:off + n].copy_(h)
            self.torch.cuda.synchronize()
        else:
            self.dist.all_reduce(buf[off:off + n], group=group)
            self.torch.cuda.synchronize()


    def all_reduce_on(self, buf, n, stream_ptr=None, off=0, group=None):
        """all-reduce(sum) of buf[:n] ordered behind the work already queued on the HIP stream `stream_ptr` (and ahead of what
        is queued on it afterwards).  nccl (RCCL over xGMI): issued under torch's ExternalStream of that stream — no host
        synchronisation.  gloo / host staging (CPU tests, several ranks on one GPU): synchronous."""
        if n == 0:
            return
        if self.device is None or self.stage or stream_ptr is None:
            if self.device is not None:
                self.torch.cuda.synchronize()
            return self.all_reduce(buf, n, off, group)
        ext = self._ext.get(stream_ptr)
        if ext is None:
            ext = self._ext[stream_ptr] = self.torch.cuda.ExternalStream(stream_ptr, device=self.device)
        with self.torch.cuda.stream(ext):
            self.dist.all_reduce(buf[off:off + n], group=group)


class LocalRanks:
    """The ranks of a job as THREADS of one process (all on the process's GPU, or on the host): the TorchComm interface over barriers.
    A rehearsal vehicle — the GPU boxes allow few processes on the card, so the 8-rank arrangement of BASELINE configs[3] (one robot
    per rank, the halves of the job four ranks each) is driven as eight threads with one CholBatch each; collectives are sums in rank
    order (deterministic), sub-groups as in torch.distributed.  comm(rank) is what a rank passes as `base`."""

    def __init__(self, world, device=None, timeout=600.0):
        import threading
        self.world, self.device, self.timeout = world, device, timeout
        self.lock = threading.Lock()
        self.barriers = {}
        self.slots = {}

    def barrier_of(self, ranks):
        import threading
        with self.lock:
            b = self.barriers.get(ranks)
            if b is None:
                b = self.barriers[ranks] = threading.Barrier(len(ranks))
            return b

    def comm(self, rank):
        return LocalRankComm(self, rank)

    def abort(self):
        with self.lock:
            for b in self.barriers.values():
                b.abort()


class LocalRankComm:
    def __init__(self, job, rank):
        self.job, self.rank, self.device = job, rank, job.device
        self.all = tuple(range(job.world))
        if self.device is not None:
            import torch
            self.torch = torch

    def bind_stream(self, stream_ptr):
        """The HIP stream this rank's passes run on (PassDriver: its CholBatch's): what _sync waits for beside torch's own stream."""
        self.stream_ptr = stream_ptr

    def _sync(self):
        # NOT torch.cuda.synchronize(): a device-wide synchronisation of one rank thread invalidates the stream capture another rank
        # thread has going at that moment (round 5 finding, gpurun_out/r5_threads_capture2.log: part 1's captures came back
        # "invalidated" while the other threads ran their exchanges).  The rank waits for ITS stream and for torch's current one.
        if self.device is not None:
            ptr = getattr(self, "stream_ptr", None)
            if ptr:
                self.torch.cuda.ExternalStream(int(ptr), device=self.device).synchronize()
            self.torch.cuda.current_stream(self.device).synchronize()

    def _rendezvous(self, ranks, item, reduce_fn):
        """Every rank of `ranks` deposits `item`; the lowest rank runs reduce_fn(items in rank order); returns its result to all."""
        job = self.job
        bar = job.barrier_of(ranks)
        key = (ranks, "in")
        with job.lock:
            job.slots.setdefault(key, {})[self.rank] = item
        bar.wait(job.timeout)
        if self.rank == ranks[0]:
            with job.lock:
                items = [job.slots[key][r] for r in ranks]
            res = reduce_fn(items)
            with job.lock:
                job.slots[(ranks, "out")] = res
        bar.wait(job.timeout)
        with job.lock:
            res = job.slots[(ranks, "out")]
        bar.wait(job.timeout)       # (nobody deposits the next item before everybody has read this result)
        return res

    def all_gather_object(self, obj):
        return self._rendezvous(self.all, obj, list)

    def alloc(self, n):
        if self.device is None:
            return np.zeros(n)
        return _device_zeros(self.torch, n, self.device)

    def handle(self, buf):
        return buf if self.device is None else buf.data_ptr()

    def new_group(self, ranks):
        return tuple(ranks)

    def all_reduce(self, buf, n, off=0, group=None):
        if n == 0:
            return
        ranks = self.all if group is None else tuple(group)
        if self.rank not in ranks:
            return
        self._sync()

        def total(views):
            tot = views[0].copy() if self.device is None else views[0].clone()
            for v in views[1:]:
                tot += v
            for v in views:
                v[:] = tot
            self._sync()
            return None
        self._rendezvous(ranks, buf[off:off + n], total)

    def all_reduce_on(self, buf, n, stream_ptr=None, off=0, group=None):
        return self.all_reduce(buf, n, off, group)


class ThreadGroup:
    """Several robot shards in ONE process (one thread each, all on the process's GPU or host): the communicator of the
    BASELINE "8-robot graph at 1/2/4/8 GPUs" layout, 8 / N robots per GPU.  Virtual rank of thread t = rank * n_local + t.
    The group leader (thread 0) adds the local shards' buffers and, when the job spans processes, all-reduces the sum with
    torch.distributed (`base`, a TorchComm); the shards' HIP streams run their phases concurrently on the one GPU."""

    def __init__(self, n_local, base=None, rank=0, world=1):
        import threading
        self.n, self.base, self.rank, self.world = n_local, base, rank, world
        self.barrier = threading.Barrier(n_local)
        self.slots = [None] * n_local
        self.result = None

    def comm(self, t, device=None):
        return ThreadComm(self, t, device)


class ThreadComm:
    def __init__(self, group, t, device=None):
        self.g, self.t, self.device = group, t, device
        if device is not None:
            import torch
            self.torch = torch

    @property
    def vrank(self):
        return self.g.rank * self.g.n + self.t

    @property
    def vworld(self):
        return self.g.world * self.g.n

    def all_gather_object(self, obj):
        g = self.g
        g.slots[self.t] = obj
        g.barrier.wait()
        if self.t == 0:
            local = list(g.slots)
            if g.world > 1:
                parts = g.base.all_gather_object(local)
                g.result = [o for part in parts for o in part]
            else:
                g.result = local
        g.barrier.wait()
        out = list(g.result)
        g.barrier.wait()
        return out

    def alloc(self, n):
        if self.device is None:
            return np.zeros(n)
        return _device_zeros(self.torch, n, self.device)

    def handle(self, buf):
        return buf if self.device is None else buf.data_ptr()

    def all_reduce(self, buf, n):
        if n == 0:
            return
        g = self.g
        g.slots[self.t] = buf
        g.barrier.wait()
        if self.t == 0:
            if self.device is None:
                total = np.sum([b[:n] for b in g.slots], axis=0)
                if g.world > 1:
                    g.base.all_reduce(total, n)
                for b in g.slots:
                    b[:n] = total
            else:
                total = self.torch.stack([b[:n] for b in g.slots]).sum(0)
                if g.world > 1:
                    g.base.all_reduce(total, n)
                for b in g.slots:
                    b[:n].copy_(total)
                self.torch.cuda.synchronize()
        g.barrier.wait()
