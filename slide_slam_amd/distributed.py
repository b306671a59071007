"""One robot per GPU: cross-robot landmark association and the distributed Gauss-Newton pass.

The reference keeps a FULL replica of every robot's graph in every `sloam_node` and gossips packets over ROS
topics (databaseManager.cpp:219-279; ingestion sloamNode.cpp:912-1002).  Here robot r's poses, factors and maps
live on GPU r only; landmarks observed by several robots are replicated, and each Gauss-Newton pass exchanges
just their normal-equation blocks:

    phase 0 (local)   relinearise, linearise, per-landmark partial sums  H_ll^(r), g_l^(r)
    all-reduce(sum)   54 doubles per shared landmark                         [RCCL over xGMI]
    phase 1 (local)   invert the GLOBAL H_ll, Schur-reduce, factor and solve the robot's own pose system,
                      t_l^(r) = sum_f E_f^T delta_p
    all-reduce(sum)   9 doubles per shared landmark
    phase 2 (local)   landmark back-substitution with the global t_l, retract

i.e. block-Jacobi over robots on the reduced pose system with the exact gradient; its fixed point is the joint
optimum the reference's replica converges to.

Inter-robot relative-pose factors (addRelativeMeasFactor, graph.cpp:247-258) couple poses of two ranks: each of the two
ranks holds the factor with its own pose as the variable and the other pose as a *ghost* — a constant refreshed at the
start of every pass (phase 20: pack the owned ghost poses' estimates, all-reduce(sum) of 12 doubles per ghost slot,
phase 21: adopt).  Same block-Jacobi argument: the cross block J_a^T J_b is dropped from the step, the gradient is exact.

`shard` is any object with ``landmark_table(cls)``, ``graph.set_shared``, ``graph.dist_phase`` — the product's
``SlideBackend`` on a GPU, or (tests only) the oracle wrapper; `comm` moves the exchange buffer.
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
        """relmeas: the job's inter-robot measurements [(pose index k, robot a, robot b, rel7 a->b)], identical on every
        rank.  Ghost slots = the sorted (robot, k) pairs they touch; this rank adds the factors that involve its robot."""
        keys = sorted({(a, k) for (k, a, b, _) in relmeas} | {(b, k) for (k, a, b, _) in relmeas})
        slot = {key: i for i, key in enumerate(keys)}
        self.n_gslots = len(keys)
        own_robot = np.array([local_robot if r == self.rank else -1 for (r, _) in keys], np.int32)
        own_idx = np.array([k for (_, k) in keys], np.int64)
        self.shard.graph.set_ghosts(own_robot, own_idx)
        for (k, a, b, rel) in relmeas:
            if a == self.rank:
                self.shard.graph.add_relative_meas_ghost(rel, k, local_robot, slot[(b, k)], True)
            if b == self.rank:
                self.shard.graph.add_relative_meas_ghost(rel, k, local_robot, slot[(a, k)], False)
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


class TorchComm:
    """torch.distributed plumbing: `nccl` (= RCCL over xGMI) with device buffers, or `gloo` with host buffers
    (CPU tests; also lets several ranks share one GPU by staging through the host)."""

    def __init__(self, device=None, stage_through_host=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.device = device
        self.stage = stage_through_host

    def all_gather_object(self, obj):
        out = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(out, obj)
        return out

    def alloc(self, n):
        if self.device is None:
            return np.zeros(n)
        return self.torch.zeros(n, dtype=self.torch.float64, device=self.device)

    def handle(self, buf):
        return buf if self.device is None else buf.data_ptr()

    def all_reduce(self, buf, n):
        if n == 0:
            return
        if self.device is None:
            t = self.torch.from_numpy(buf[:n])
            self.dist.all_reduce(t)
        elif self.stage:
            h = buf[:n].cpu()
            self.dist.all_reduce(h)
            buf[:n].copy_(h)
            self.torch.cuda.synchronize()
        else:
            self.dist.all_reduce(buf[:n])
            self.torch.cuda.synchronize()


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
        return self.torch.zeros(n, dtype=self.torch.float64, device=self.device)

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
