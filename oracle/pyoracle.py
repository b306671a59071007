"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes access to oracle/_build/liboracle.so (the CPU restatement of the reference hot path).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (slide_slam_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

F_PRIOR, F_BETWEEN, F_BR, F_CUBE, F_CYL = range(5)
V_POSE, V_POINT, V_CUBE, V_CYL = range(4)
CHART_CAYLEY, CHART_EXPMAP = 0, 1


def _cpu_tag() -> str:
    """Short hash of this host's CPU model and feature flags: a -march=native build made on another machine (the build travels with
    the tree to the GPU box) must not be loaded there."""
    import hashlib
    try:
        with open("/proc/cpuinfo") as fh:
            txt = fh.read()
        keep = [ln for ln in txt.splitlines() if ln.startswith(("model name", "flags"))][:2]
    except OSError:
        keep = []
    return hashlib.sha1("\n".join(keep).encode()).hexdigest()[:10]


def _native_name() -> str:
    return f"liboracle_native_{_cpu_tag()}.so"


def build(native: bool = False) -> str:
    """Compile the oracle (g++).  native=True builds a -march=native copy for CPU-baseline timing (one per kind of host CPU)."""
    out = f"_build/{_native_name()}" if native else "_build/liboracle.so"
    flags = "-O3 -march=native" if native else "-O3 -march=x86-64-v3"
    cmd = ["make", "-C", _HERE, f"OUT={out}",
           f"CXXFLAGS={flags} -ffp-contract=off -fopenmp -std=c++17 -fPIC -Wall -Wno-unused-function"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, out)


def lib(native: bool = False):
    global _LIB
    if _LIB is not None and not native:
        return _LIB
    path = os.path.join(_HERE, "_build", _native_name() if native else "liboracle.so")
    try:
        path = build(native)          # make: a no-op when the library is newer than its sources, so a stale build cannot be loaded
    except (subprocess.CalledProcessError, OSError):
        if not os.path.exists(path):
            raise
    L = C.CDLL(path)
    L.orc_graph_create.restype = C.c_void_p
    L.orc_backend_create.restype = C.c_void_p
    L.orc_backend_graph.restype = C.c_void_p
    if not native:
        _LIB = L
    return L


class OrcParams(C.Structure):
    _fields_ = [("pose_chart", C.c_int), ("relin_threshold", C.c_double),
                ("prior_sigma", C.c_double * 6), ("odom_sigma", C.c_double * 6), ("cube_sigma", C.c_double * 9),
                ("relmeas_sigma", C.c_double * 6), ("cyl_sigma", C.c_double), ("bearing_sigma", C.c_double),
                ("cyl_thresh", C.c_double), ("cube_thresh", C.c_double), ("ell_thresh", C.c_double),
                ("num_threads", C.c_int)]

    @staticmethod
    def default(**kw):
        p = OrcParams()
        p.pose_chart = CHART_CAYLEY
        p.relin_threshold = 0.1
        for i in range(6):
            p.prior_sigma[i] = 1e-6
            p.odom_sigma[i] = 0.1
            p.relmeas_sigma[i] = 0.1
        for i in range(9):
            p.cube_sigma[i] = 0.1
        p.cyl_sigma = 400.0
        p.bearing_sigma = 1.0
        p.cyl_thresh, p.cube_thresh, p.ell_thresh = 2.0, 2.0, 0.75
        p.num_threads = 1
        for k, v in kw.items():
            if isinstance(v, (list, tuple, np.ndarray)):
                arr = getattr(p, k)
                for i, x in enumerate(v):
                    arr[i] = float(x)
            else:
                setattr(p, k, v)
        return p


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleGraph:
    """SemanticFactorGraph seam (reference graph.h:70-121) over the CPU restatement."""

    def __init__(self, params: OrcParams | None = None, handle=None, L=None):
        self.L = L or lib()
        self.own = handle is None
        self.h = C.c_void_p(self.L.orc_graph_create(C.byref(params) if params else None)) if handle is None else handle

    def __del__(self):
        if getattr(self, "own", False) and self.h:
            self.L.orc_graph_destroy(self.h)
            self.h = None

    def set_prior(self, robot, pose7):
        self.L.orc_graph_set_prior(self.h, C.c_int(robot), _p(_d(pose7)))

    def add_keypose_between(self, robot, frm, to, rel7, est7):
        self.L.orc_graph_add_keypose_between(self.h, C.c_int(robot), C.c_uint64(frm), C.c_uint64(to), _p(_d(rel7)),
                                             _p(_d(est7)))

    def add_loop_closure(self, rel7, i1, r1, i2, r2):
        self.L.orc_graph_add_loop_closure(self.h, _p(_d(rel7)), C.c_uint64(i1), C.c_int(r1), C.c_uint64(i2), C.c_int(r2))

    def add_relative_meas(self, rel7, i1, r1, i2, r2):
        self.L.orc_graph_add_relative_meas(self.h, _p(_d(rel7)), C.c_uint64(i1), C.c_int(r1), C.c_uint64(i2), C.c_int(r2))

    def add_point_landmark(self, idx, xyz):
        self.L.orc_graph_add_point_landmark(self.h, C.c_uint64(idx), _p(_d(xyz)))

    def add_range_bearing(self, robot, pose_idx, lm_idx, bearing, rng):
        self.L.orc_graph_add_range_bearing(self.h, C.c_int(robot), C.c_uint64(pose_idx), C.c_uint64(lm_idx),
                                           _p(_d(bearing)), C.c_double(rng))

    def add_cube(self, robot, pose_idx, cube_idx, pose7, cube7, scale, exists):
        self.L.orc_graph_add_cube(self.h, C.c_int(robot), C.c_uint64(pose_idx), C.c_uint64(cube_idx), _p(_d(pose7)),
                                  _p(_d(cube7)), _p(_d(scale)), C.c_int(int(exists)))

    def add_cylinder(self, robot, pose_idx, cyl_idx, pose7, root, ray, radius, exists):
        self.L.orc_graph_add_cylinder(self.h, C.c_int(robot), C.c_uint64(pose_idx), C.c_uint64(cyl_idx), _p(_d(pose7)),
                                      _p(_d(root)), _p(_d(ray)), C.c_double(radius), C.c_int(int(exists)))

    def solve(self):
        return int(self.L.orc_graph_solve(self.h))

    def set_relin_threshold(self, thr):
        self.L.orc_graph_set_relin_threshold(self.h, C.c_double(thr))

    def set_wildfire(self, thr):
        """[GTSAM] iSAM2's wildfire threshold on the back-substitution (1e-3 in the reference's build); 0 = exact (default)."""
        self.L.orc_graph_set_wildfire(self.h, C.c_double(thr))

    def wildfire_stats(self):
        out = (C.c_longlong * 3)()
        self.L.orc_graph_wildfire_stats(self.h, out)
        return dict(kept_total=int(out[0]), kept_last=int(out[1]), last_cd=int(out[2]))

    def get_pose(self, robot, idx):
        out = np.zeros(7)
        st = self.L.orc_graph_get_pose(self.h, C.c_int(robot), C.c_uint64(idx), _p(out))
        return st, out

    def get_pose12(self, robot, idx):
        out = np.zeros(12)
        st = self.L.orc_graph_get_pose12(self.h, C.c_int(robot), C.c_uint64(idx), _p(out))
        return st, out

    def get_landmark(self, cls, idx):
        out = np.zeros(15)
        st = self.L.orc_graph_get_landmark(self.h, C.c_int(cls), C.c_uint64(idx), _p(out))
        return st, out[: (7, 15, 3)[cls]]

    def set_shared(self, cls, idx, owner):
        cls, owner = _i(cls), _i(owner)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        rc = self.L.orc_graph_set_shared(self.h, _p(cls), _p(idx), _p(owner), C.c_int(len(cls)))
        if rc != 0:
            raise RuntimeError("set_shared: landmark missing")

    def dist_phase(self, phase, buf):
        """buf: numpy float64 array (host) — same layout as the product's device buffer."""
        rc = self.L.orc_graph_dist_phase(self.h, C.c_int(phase), _p(buf))
        if rc != 0:
            raise RuntimeError(f"oracle dist_phase {phase} failed: {rc}")
        return rc

    def set_pcg(self, iterations, tol=0.0):
        """Joint solve of the sharded pass (dist_phase 31 / 32 / 33): upper bound on the PCG iterations and the relative tolerance
        on sqrt(r^T M^-1 r) below which the iterations become no-ops (0: every iteration counts)."""
        self.L.orc_graph_set_pcg(self.h, C.c_int(int(iterations)), C.c_double(float(tol)))

    def pcg_stats(self):
        out = np.zeros(4)
        self.L.orc_graph_pcg_stats(self.h, _p(out))
        return dict(iterations=int(out[0]), gamma_first=out[1], gamma_last=out[2], state=int(out[3]))

    def set_separator(self, offsets):
        """Exact joint step (dist_phase 40 / 41 / 42): offsets[slot] = offset of the shared slot's tangent coordinates in the separator
        system of all shared landmarks (len n_slots + 1, the same on every rank)."""
        off = _i(offsets)
        self.L.orc_graph_set_separator(self.h, _p(off), C.c_int(len(off)))

    def set_ghost_ids(self, ids, n_total):
        """Exact joint step: ids[i] = index, in the job's list of n_total relative-pose measurements, of this graph's i-th ghost factor
        (the factor then enters through six separator coordinates of its own instead of being frozen at the ghost pose)."""
        a = _i(ids)
        self.L.orc_graph_set_ghost_ids(self.h, _p(a), C.c_int(len(a)), C.c_int(int(n_total)))

    def keep_factor(self, on=True):
        self.L.orc_graph_keep_factor(self.h, C.c_int(int(on)))

    def pose_covariance(self, robot, idx):
        """getPoseCovariance (graph.cpp:314-323) at the linearisation point of the last solve()."""
        out = np.zeros(36)
        st = self.L.orc_graph_pose_covariance(self.h, C.c_int(robot), C.c_uint64(idx), _p(out))
        return st, out.reshape(6, 6)

    def set_ghosts(self, own_robot, own_idx):
        r = _i(own_robot)
        i = np.ascontiguousarray(own_idx, dtype=np.int64)
        if self.L.orc_graph_set_ghosts(self.h, _p(r), _p(i), C.c_int(len(r))) != 0:
            raise RuntimeError("set_ghosts: pose missing")

    def add_relative_meas_ghost(self, rel7, idx, robot, slot, local_first):
        self.L.orc_graph_add_relative_meas_ghost(self.h, _p(_d(rel7)), C.c_uint64(idx), C.c_int(robot), C.c_int(slot),
                                                 C.c_int(int(local_first)))

    def stats(self):
        out = np.zeros(8)
        self.L.orc_graph_stats(self.h, _p(out))
        return dict(n_pose=int(out[0]), n_lm=int(out[1]), n_factors=int(out[2]), n_relin=int(out[3]),
                    t_linearize=out[4], t_schur=out[5], t_chol=out[6], t_total=out[7])


class OracleBackend:
    """runSLOAMNode seam (reference sloamNode.cpp:762-1036) over the CPU restatement."""

    def __init__(self, params: OrcParams | None = None, num_robots: int = 13, L=None):
        self.L = L or lib()
        self.n_robots = num_robots
        self.h = C.c_void_p(self.L.orc_backend_create(C.byref(params) if params else None, C.c_int(num_robots)))
        self.graph = OracleGraph(handle=C.c_void_p(self.L.orc_backend_graph(self.h)), L=self.L)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_backend_destroy(self.h)
            self.h = None

    def process_frame(self, robot, rel7, prev7, det, mode=0):
        nc, nb, ne = len(det["cyl_label"]), len(det["cube_label"]), len(det["ell_label"])
        out7 = np.zeros(7)
        cm, bm, em = np.full(nc, -1, np.int32), np.full(nb, -1, np.int32), np.full(ne, -1, np.int32)
        cid, bid, eid = np.full(nc, -1, np.int32), np.full(nb, -1, np.int32), np.full(ne, -1, np.int32)
        tm = np.zeros(2)
        a = [_d(det["cyl_root"]), _d(det["cyl_ray"]), _d(det["cyl_radius"]), _i(det["cyl_label"]),
             _d(det["cube_pose7"]), _d(det["cube_scale"]), _i(det["cube_label"]),
             _d(det["ell_pose7"]), _d(det["ell_scale"]), _i(det["ell_label"])]
        st = self.L.orc_backend_process_frame(
            self.h, C.c_int(mode), C.c_int(robot), _p(_d(rel7)), _p(_d(prev7)),
            C.c_int(nc), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]),
            C.c_int(nb), _p(a[4]), _p(a[5]), _p(a[6]),
            C.c_int(ne), _p(a[7]), _p(a[8]), _p(a[9]),
            _p(out7), _p(cm), _p(bm), _p(em), _p(cid), _p(bid), _p(eid), _p(tm))
        return dict(status=int(st), pose7=out7, cyl_match=cm, cube_match=bm, ell_match=em, cyl_id=cid, cube_id=bid,
                    ell_id=eid, t_assoc=tm[0], t_graph=tm[1])

    def ingest_solve(self):
        return int(self.L.orc_backend_ingest_solve(self.h))

    def end_frame(self, robot):
        out = np.zeros(7)
        st = self.L.orc_backend_end_frame(self.h, C.c_int(robot), _p(out))
        return int(st), out

    def counts(self):
        out = np.zeros(4, np.uint64)
        pc = np.zeros(self.n_robots, np.uint64)
        self.L.orc_backend_counts(self.h, _p(out), _p(pc), C.c_int(self.n_robots))
        return dict(cyl=int(out[0]), cube=int(out[1]), point=int(out[2]), factors=int(out[3]), poses=pc.astype(np.int64))

    def landmark_table(self, cls):
        n = self.counts()[("cyl", "cube", "point")[cls]]
        xyz = np.zeros((max(n, 1), 3))
        lab = np.zeros(max(n, 1), np.int32)
        k = self.L.orc_backend_landmark_table(self.h, C.c_int(cls), _p(xyz), _p(lab), C.c_int(n))
        return xyz[:k], lab[:k]

    def map_model(self, cls, idx):
        out = np.zeros(7)
        hits, label = C.c_int(0), C.c_int(0)
        st = self.L.orc_backend_map_model(self.h, C.c_int(cls), C.c_int(idx), _p(out), C.byref(hits), C.byref(label))
        return int(st), out[: 7 if cls == 0 else 6], hits.value, label.value
