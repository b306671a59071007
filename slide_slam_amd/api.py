"""ctypes binding of include/slide_gpu.h (the C-ABI drop-in boundary of the sloam backend hot path).

The shared library carries gfx950 HIP kernels only.  There is no CPU fallback: importing works
anywhere (so that symbol/ABI tests run without a GPU), but every compute call fails loudly with
``SlideError`` when the library or a gfx950 device is missing.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libslide_gpu.so")
_LIB = None

SLIDE_OK, SLIDE_MISSING = 0, 1
ERR = {-1: "SLIDE_ERR_INVALID", -2: "SLIDE_ERR_NOT_SPD", -3: "SLIDE_ERR_CAPACITY", -4: "SLIDE_ERR_HIP", -5: "SLIDE_ERR_RUNTIME"}
CHART_CAYLEY, CHART_EXPMAP = 0, 1
CLS_CYLINDER, CLS_CUBE, CLS_ELLIPSOID = 0, 1, 2
FRAME_HOST, FRAME_HOST_DEFERRED, FRAME_FOREIGN = 0, 1, 2

# every symbol include/slide_gpu.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = [
    "slide_default_params", "slide_device_check", "slide_last_error", "slide_version",
    "slide_graph_create", "slide_graph_destroy", "slide_graph_set_prior", "slide_graph_add_keypose_between",
    "slide_graph_add_loop_closure", "slide_graph_add_relative_meas", "slide_graph_add_point_landmark",
    "slide_graph_add_range_bearing", "slide_graph_add_cube", "slide_graph_add_cylinder", "slide_graph_solve",
    "slide_graph_gauss_newton", "slide_graph_get_pose", "slide_graph_get_pose12", "slide_graph_get_all_poses",
    "slide_graph_get_landmark", "slide_graph_get_pose_covariance", "slide_graph_stats", "slide_graph_rejected_count", "slide_graph_set_shared", "slide_graph_dist_phase", "slide_chol_batch_create", "slide_chol_batch_destroy", "slide_graph_join_chol_batch", "slide_graph_dist_pass_local", "slide_chol_batch_pass", "slide_chol_batch_pass_part", "slide_chol_batch_stream", "slide_chol_batch_set_pcg", "slide_graph_set_pcg", "slide_chol_batch_set_pcg_tolerance", "slide_graph_set_pcg_tolerance", "slide_graph_set_separator", "slide_chol_batch_set_exact_joint", "slide_chol_batch_sep_buffer_len", "slide_chol_batch_sep_exchange_len", "slide_chol_batch_profile_exact_joint", "slide_graph_get_border_profile", "slide_graph_get_incremental_stats", "slide_graph_set_wildfire", "slide_graph_get_wildfire_stats", "slide_graph_get_segments", "slide_graph_get_segment_table", "slide_chol_batch_set_segments", "slide_clipper_dense_clique_batch", "slide_clipper_last_solve_info", "slide_last_device_ms", "slide_chol_batch_set_separator_profile", "slide_chol_batch_set_separator_blocks", "slide_chol_batch_set_separator_owner", "slide_chol_batch_sep_segment", "slide_graph_set_incremental", "slide_graph_set_ghost_ids", "slide_graph_get_pcg_stats", "slide_graph_get_tile_profile", "slide_graph_set_dense_profile", "slide_graph_chi2", "slide_chol_batch_profile", "slide_graph_set_ghosts", "slide_graph_add_relative_meas_ghost",
    "slide_backend_landmark_table", "slide_graph_set_profiling", "slide_graph_get_profile",
    "slide_dense_spd_solve", "slide_dense_spd_solve_ex", "slide_debug_chol_bordered", "slide_debug_pair_timeouts", "slide_submap_knn", "slide_assoc_match_cylinders", "slide_assoc_match_boxes", "slide_assoc_sweep_batch_device", "slide_assoc_sweep_batch",
    "slide_backend_create", "slide_backend_destroy", "slide_backend_process_frame", "slide_backend_ingest_solve",
    "slide_backend_end_frame", "slide_backend_graph", "slide_backend_counts", "slide_backend_map_model",
    "slide_place_default_params", "slide_match_maps", "slide_find_inter_loop_closure", "slide_find_intra_loop_closure",
    "slide_loop_candidate_idx", "slide_clipper_affinity",
    "slide_closest_stamp", "slide_clipper_default_params", "slide_clipper_dense_clique", "slide_match_triangles",
    "slide_estimate_tf2d", "slide_semantic_clipper", "slide_find_relative_meas_match", "slide_delaunay_2d", "slide_run_semantic_clipper",
    "slide_pick_next_measurement", "slide_in_loop_closure_region",
]


class SlideError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("pose_chart", C.c_int), ("relinearize_threshold", C.c_double), ("noise_floor", C.c_double),
                ("noise_model_prior_first_pose_vec", C.c_double * 6), ("noise_model_odom_vec", C.c_double * 6),
                ("noise_model_cube_vec", C.c_double * 9), ("noise_model_rel_meas_vec", C.c_double * 6),
                ("cylinder_sigma", C.c_double), ("bearing_range_sigma", C.c_double), ("numdiff_delta", C.c_double),
                ("cylinder_match_thresh", C.c_double), ("cuboid_match_thresh", C.c_double),
                ("ellipsoid_match_thresh", C.c_double), ("knn_cylinder", C.c_int), ("knn_cube", C.c_int),
                ("knn_ellipsoid", C.c_int), ("number_of_robots", C.c_int), ("device", C.c_int)]


class Detections(C.Structure):
    _fields_ = [("n_cyl", C.c_int), ("cyl_root", C.c_void_p), ("cyl_ray", C.c_void_p), ("cyl_radius", C.c_void_p),
                ("cyl_label", C.c_void_p), ("n_cube", C.c_int), ("cube_pose7", C.c_void_p), ("cube_scale", C.c_void_p),
                ("cube_label", C.c_void_p), ("n_ell", C.c_int), ("ell_pose7", C.c_void_p), ("ell_scale", C.c_void_p),
                ("ell_label", C.c_void_p)]


class FrameResult(C.Structure):
    _fields_ = [("out_pose7", C.c_double * 7), ("cyl_match", C.c_void_p), ("cube_match", C.c_void_p),
                ("ell_match", C.c_void_p), ("cyl_id", C.c_void_p), ("cube_id", C.c_void_p), ("ell_id", C.c_void_p),
                ("optimized", C.c_int), ("ms_association", C.c_double), ("ms_graph", C.c_double)]


class PlaceParams(C.Structure):
    _fields_ = [("dilation_factor", C.c_double), ("search_xy_step_size", C.c_double), ("match_yaw_half_range", C.c_double),
                ("search_yaw_step_size", C.c_double), ("match_threshold_position", C.c_double),
                ("match_threshold_dimension", C.c_double), ("disable_yaw_search", C.c_int), ("ignore_dimension", C.c_int),
                ("min_num_inliers", C.c_int), ("use_nonlinear_least_squares", C.c_int),
                ("min_num_map_objects_to_start", C.c_int), ("max_rings", C.c_int)]


def lib():
    """Load libslide_gpu.so (built by slide_slam_amd.build / __graft_entry__.build)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SlideError(f"{LIB_PATH} is missing: run `python -m slide_slam_amd.build` (hipcc, gfx950). "
                             "There is no CPU fallback.")
        # (SLIDE_LIB_VARIANT=<name>: an experiment build _lib/<name>.so of the SAME sources with other compile-time constants,
        # tools/build_variant.py — kernel tuning only; it is a build of this library, not another path)
        var = os.environ.get("SLIDE_LIB_VARIANT")
        path = os.path.join(_HERE, "_lib", var + ".so") if var else LIB_PATH
        if var and not os.path.exists(path):
            raise SlideError(f"{path} is missing (SLIDE_LIB_VARIANT)")
        L = C.CDLL(path)
        L.slide_last_error.restype = C.c_char_p
        L.slide_version.restype = C.c_char_p
        L.slide_graph_create.restype = C.c_void_p
        L.slide_backend_create.restype = C.c_void_p
        L.slide_backend_graph.restype = C.c_void_p
        L.slide_chol_batch_create.restype = C.c_void_p
        L.slide_graph_rejected_count.restype = C.c_int64
        L.slide_graph_rejected_count.argtypes = [C.c_void_p]
        L.slide_chol_batch_destroy.argtypes = [C.c_void_p]
        L.slide_chol_batch_destroy.restype = None
        L.slide_chol_batch_stream.restype = C.c_void_p
        L.slide_chol_batch_stream.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def last_error() -> str:
    return lib().slide_last_error().decode()


def _check(rc, allow_missing=False):
    if rc == SLIDE_OK or (allow_missing and rc == SLIDE_MISSING):
        return rc
    raise SlideError(f"{ERR.get(rc, rc)}: {last_error()}")


def default_params(**kw) -> Params:
    p = Params()
    lib().slide_default_params(C.byref(p))
    for k, v in kw.items():
        if isinstance(v, (list, tuple, np.ndarray)):
            arr = getattr(p, k)
            for i, x in enumerate(v):
                arr[i] = float(x)
        else:
            setattr(p, k, v)
    return p


def device_check(device: int = -1) -> None:
    _check(lib().slide_device_check(C.c_int(device)))


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return C.c_void_p(a.ctypes.data)      # (ndarray.ctypes.data_as costs 2.9 us a call, this 1.1: a streaming frame passes eighteen pointers)


class SlideGraph:
    """SemanticFactorGraph seam (reference include/factorgraph/graph.h:70-121) on the MI355X."""

    def __init__(self, params: Params | None = None, handle=None):
        self.L = lib()
        self.own = handle is None
        if handle is None:
            h = self.L.slide_graph_create(C.byref(params) if params is not None else None)
            if not h:
                raise SlideError(f"slide_graph_create failed: {last_error()}")
            handle = C.c_void_p(h)
        self.h = handle

    def __del__(self):
        if getattr(self, "own", False) and getattr(self, "h", None):
            self.L.slide_graph_destroy(self.h)
            self.h = None

    def set_prior(self, robot, pose7):
        _check(self.L.slide_graph_set_prior(self.h, C.c_int(robot), _p(_d(pose7))))

    def add_keypose_between(self, robot, frm, to, rel7, est7):
        _check(self.L.slide_graph_add_keypose_between(self.h, C.c_int(robot), C.c_uint64(frm), C.c_uint64(to), _p(_d(rel7)),
                                                      _p(_d(est7))))

    def add_loop_closure(self, rel7, i1, r1, i2, r2):
        _check(self.L.slide_graph_add_loop_closure(self.h, _p(_d(rel7)), C.c_uint64(i1), C.c_int(r1), C.c_uint64(i2),
                                                   C.c_int(r2)))

    def add_relative_meas(self, rel7, i1, r1, i2, r2):
        _check(self.L.slide_graph_add_relative_meas(self.h, _p(_d(rel7)), C.c_uint64(i1), C.c_int(r1), C.c_uint64(i2),
                                                    C.c_int(r2)))

    def add_point_landmark(self, idx, xyz):
        _check(self.L.slide_graph_add_point_landmark(self.h, C.c_uint64(idx), _p(_d(xyz))))

    def add_range_bearing(self, robot, pose_idx, lm_idx, bearing, rng):
        _check(self.L.slide_graph_add_range_bearing(self.h, C.c_int(robot), C.c_uint64(pose_idx), C.c_uint64(lm_idx),
                                                    _p(_d(bearing)), C.c_double(rng)))

    def add_cube(self, robot, pose_idx, cube_idx, pose7, cube7, scale, exists):
        _check(self.L.slide_graph_add_cube(self.h, C.c_int(robot), C.c_uint64(pose_idx), C.c_uint64(cube_idx), _p(_d(pose7)),
                                           _p(_d(cube7)), _p(_d(scale)), C.c_int(int(exists))))

    def add_cylinder(self, robot, pose_idx, cyl_idx, pose7, root, ray, radius, exists):
        _check(self.L.slide_graph_add_cylinder(self.h, C.c_int(robot), C.c_uint64(pose_idx), C.c_uint64(cyl_idx),
                                               _p(_d(pose7)), _p(_d(root)), _p(_d(ray)), C.c_double(radius),
                                               C.c_int(int(exists))))

    def solve(self):
        return _check(self.L.slide_graph_solve(self.h))

    def gauss_newton(self, iterations=1):
        return _check(self.L.slide_graph_gauss_newton(self.h, C.c_int(iterations)))

    def get_pose(self, robot, idx):
        out = np.zeros(7)
        st = _check(self.L.slide_graph_get_pose(self.h, C.c_int(robot), C.c_uint64(idx), _p(out)), True)
        return st, out

    def get_pose12(self, robot, idx):
        out = np.zeros(12)
        st = _check(self.L.slide_graph_get_pose12(self.h, C.c_int(robot), C.c_uint64(idx), _p(out)), True)
        return st, out

    def get_all_poses(self, robot, cap):
        out = np.zeros((cap, 7))
        n = C.c_uint64(0)
        _check(self.L.slide_graph_get_all_poses(self.h, C.c_int(robot), _p(out), C.c_uint64(cap), C.byref(n)))
        return out[: n.value]

    def get_landmark(self, cls, idx):
        out = np.zeros(15)
        st = _check(self.L.slide_graph_get_landmark(self.h, C.c_int(cls), C.c_uint64(idx), _p(out)), True)
        return st, out[: (7, 15, 3)[cls]]

    def stats(self):
        out = np.zeros(5, np.int64)
        _check(self.L.slide_graph_stats(self.h, _p(out)))
        return dict(n_pose=int(out[0]), n_lm=int(out[1]), n_factors=int(out[2]), n_relin=int(out[3]), chol_dim=int(out[4]))

    def rejected_count(self):
        """Entries (factors on unknown keys, values inserted twice) refused since creation; the reference's isam->update throws there."""
        return int(self.L.slide_graph_rejected_count(self.h))

    def set_shared(self, cls, idx, owner):
        cls, owner = _i(cls), _i(owner)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        _check(self.L.slide_graph_set_shared(self.h, _p(cls), _p(idx), _p(owner), C.c_int(len(cls))))

    def get_pose_covariance(self, robot, idx):
        """getPoseCovariance (graph.cpp:314-323).  Returns (status, 6x6)."""
        out = np.zeros(36)
        st = self.L.slide_graph_get_pose_covariance(self.h, C.c_int(robot), C.c_uint64(idx), _p(out))
        if st < 0:
            _check(st)
        return st, out.reshape(6, 6)

    def set_ghosts(self, own_robot, own_idx):
        r = _i(own_robot)
        i = np.ascontiguousarray(own_idx, dtype=np.int64)
        _check(self.L.slide_graph_set_ghosts(self.h, _p(r), _p(i), C.c_int(len(r))))

    def add_relative_meas_ghost(self, rel7, idx, robot, slot, local_first):
        _check(self.L.slide_graph_add_relative_meas_ghost(self.h, _p(_d(rel7)), C.c_uint64(idx), C.c_int(robot), C.c_int(slot),
                                                          C.c_int(int(local_first))))

    def join_chol_batch(self, batch, slot=0):
        """Share the dense factor + solve of phase 1 with the other graphs of `batch` (CholBatch; None leaves it)."""
        _check(self.L.slide_graph_join_chol_batch(self.h, C.c_void_p(batch.h if batch is not None else None), C.c_int(slot)))

    def set_pcg(self, iterations, tol=0.0):
        """Un-batched passes: PCG iterations of the joint solve (dist_phase 31 / 32 / 33 between phases 1 and 2); 0 = block solves only.
        tol > 0: iterations past a relative reduction of sqrt(r^T M^-1 r) by tol are no-ops."""
        _check(self.L.slide_graph_set_pcg_tolerance(self.h, C.c_double(float(tol))))
        _check(self.L.slide_graph_set_pcg(self.h, C.c_int(iterations)))

    def set_ghost_ids(self, ids, n_total):
        """Exact joint step: ids[i] = index of this graph's i-th ghost factor in the job's list of n_total relative-pose measurements."""
        a = np.ascontiguousarray(ids, dtype=np.int32)
        _check(self.L.slide_graph_set_ghost_ids(self.h, _p(a), C.c_int(len(a)), C.c_int(int(n_total))))

    def set_separator(self, offsets):
        """Exact joint step: offsets of the shared slots' tangent coordinates in the separator system (n_slots + 1 ints)."""
        off = np.ascontiguousarray(offsets, dtype=np.int32)
        _check(self.L.slide_graph_set_separator(self.h, _p(off), C.c_int(len(off))))

    def chi2(self):
        """Sum of squared whitened residuals at the current estimate: dict(total, prior, between, landmark)."""
        out = np.zeros(4)
        _check(self.L.slide_graph_chi2(self.h, _p(out)))
        return dict(total=out[0], prior=out[1], between=out[2], landmark=out[3])

    def tile_profile(self):
        """Tile-level profile of the reduced pose system: prof[c] = last tile row of block column c inside it (64 x 64 tiles)."""
        T = self.L.slide_graph_get_tile_profile(self.h, None, C.c_int(0))
        if T < 0:
            _check(T)
        out = np.zeros(max(T, 1), np.int32)
        T = self.L.slide_graph_get_tile_profile(self.h, _p(out), C.c_int(T))
        if T < 0:
            _check(T)
        return out[:T]

    def set_incremental(self, on=True):
        """False: every update re-factors all block columns (the reference behaviour of round 2; same result to rounding)."""
        _check(self.L.slide_graph_set_incremental(self.h, C.c_int(int(on))))

    def set_wildfire(self, threshold):
        """iSAM2's wildfire threshold on the back-substitution of incremental updates (0: off, the default; the reference's GTSAM: 1e-3)."""
        _check(self.L.slide_graph_set_wildfire(self.h, C.c_double(threshold)))

    def wildfire_stats(self):
        out = np.zeros(2, np.int64)
        _check(self.L.slide_graph_get_wildfire_stats(self.h, _p(out)))
        return dict(kept_total=int(out[0]), kept_last=int(out[1]))

    def incremental_stats(self):
        """Updates that re-factored a suffix of the block columns only / everything, first re-factored column of the last, block columns."""
        out = np.zeros(4, np.int64)
        _check(self.L.slide_graph_get_incremental_stats(self.h, _p(out)))
        return dict(incremental=int(out[0]), full=int(out[1]), last_first_column=int(out[2]), block_columns=int(out[3]))

    def segments(self):
        """Exact joint passes: (tile ranges of the band's segments, separator poses); ([], 0) when the band is not cut."""
        out = np.zeros(32, np.int32)
        n = self.L.slide_graph_get_segments(self.h, _p(out), C.c_int(32))
        if n < 0:
            _check(-n)
        if n <= 1:
            return [], 0
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)], int(out[2 * n])

    def segment_table(self):
        """Exact joint passes over a cut band: (ends, first) — ends[s] = last block column + 1 of segment s, first[s][i] = first block
        column of segment s in which border tile row i is non-zero (i = nbr: the right-hand side; 1 << 30: never) — or None."""
        n = self.L.slide_graph_get_segment_table(self.h, None, C.c_int(0))
        if n < 0:
            _check(-n)
        if n == 0:
            return None
        out = np.zeros(n, np.int32)
        self.L.slide_graph_get_segment_table(self.h, _p(out), C.c_int(n))
        ns = int(out[0])
        w = (n - 1 - ns) // ns
        return [int(v) for v in out[1:1 + ns]], out[1 + ns:].reshape(ns, w).astype(np.int64)

    def border_profile(self):
        """Exact joint step: first[i] = first block column of the band in which border tile row i can be non-zero (len = border row tiles)."""
        n = self.L.slide_graph_get_border_profile(self.h, None, C.c_int(0))
        if n < 0:
            _check(-n)
        out = np.zeros(max(n, 1), np.int32)
        n = self.L.slide_graph_get_border_profile(self.h, _p(out), C.c_int(n))
        if n < 0:
            _check(-n)
        return out[:n]

    def set_dense_profile(self, on=True):
        """Measurement aid: make the solver ignore the structure (every tile of the lower triangle)."""
        _check(self.L.slide_graph_set_dense_profile(self.h, C.c_int(1 if on else 0)))

    def pcg_stats(self):
        out = np.zeros(8)
        _check(self.L.slide_graph_get_pcg_stats(self.h, _p(out)))
        return dict(alpha=out[2], beta=out[3], gamma_first=out[4], gamma_last=out[5])

    def dist_pass_local(self, d_buf_ptr):
        """One distributed pass with device-side exchanges: every robot of the job must be in this graph's CholBatch."""
        return _check(self.L.slide_graph_dist_pass_local(self.h, C.c_void_p(d_buf_ptr)))

    def dist_phase(self, phase, d_buf_ptr):
        """d_buf_ptr: integer DEVICE address of the exchange buffer (e.g. torch_tensor.data_ptr())."""
        return _check(self.L.slide_graph_dist_phase(self.h, C.c_int(phase), C.c_void_p(d_buf_ptr)))

    def set_profiling(self, on=True):
        _check(self.L.slide_graph_set_profiling(self.h, C.c_int(int(on))))

    def get_profile(self):
        names = C.create_string_buffer(32 * 32)
        ms = np.zeros(32)
        cnt = np.zeros(32, np.int64)
        n = self.L.slide_graph_get_profile(self.h, names, _p(ms), _p(cnt), C.c_int(32))
        out = {}
        for i in range(n):
            nm = names.raw[32 * i: 32 * i + 32].split(b"\0")[0].decode()
            out[nm] = dict(ms=float(ms[i]), launches=int(cnt[i]))
        return out


class CholBatch:
    """slide_chol_batch_t: the graphs of several robots on one GPU factor and solve their pose systems in one launch sequence."""

    def __init__(self, n_graphs):
        self.L = lib()
        self.h = self.L.slide_chol_batch_create(C.c_int(n_graphs))
        if not self.h:
            raise ValueError("CholBatch: 1 .. 8 graphs")

    def pass_all(self, buf_ptrs):
        """One distributed pass of all joined graphs from this thread; buf_ptrs[i] = device address of slot i's exchange buffer."""
        arr = (C.c_void_p * len(buf_ptrs))(*[int(p) for p in buf_ptrs])
        return _check(self.L.slide_chol_batch_pass(C.c_void_p(self.h), arr))

    def set_pcg(self, iterations, tol=0.0):
        """PCG iterations of the joint solve after the factorisations (0 = every robot's own block solve only); tol > 0: iterations
        past a relative reduction of sqrt(r^T M^-1 r) by tol are no-ops."""
        _check(self.L.slide_chol_batch_set_pcg_tolerance(C.c_void_p(self.h), C.c_double(float(tol))))
        _check(self.L.slide_chol_batch_set_pcg(C.c_void_p(self.h), C.c_int(iterations)))

    def set_exact_joint(self, on=True, sep_ptr=0, sep_len=0):
        """Passes take the EXACT joint Gauss-Newton step (shared landmarks as the separator of the joint graph, slide_gpu.h);
        sep_ptr / sep_len: the caller's device buffer for the separator system (sep_buffer_len(m) doubles) or 0."""
        _check(self.L.slide_chol_batch_set_exact_joint(C.c_void_p(self.h), C.c_int(int(on)), C.c_void_p(int(sep_ptr) or None), C.c_longlong(int(sep_len))))

    def set_segments(self, n_seg):
        """Exact joint passes: cut every robot's pose chain into n_seg segments factored side by side (nested dissection; 1 = off)."""
        _check(self.L.slide_chol_batch_set_segments(C.c_void_p(self.h), C.c_int(int(n_seg))))

    def set_separator_profile(self, prof):
        """Tile profile of the separator system's landmark part (distributed.separator_offsets)."""
        a = np.ascontiguousarray(prof, dtype=np.int32)
        _check(self.L.slide_chol_batch_set_separator_profile(C.c_void_p(self.h), _p(a), C.c_int(len(a))))

    def set_separator_blocks(self, Ta, Tb, used_a, used_b):
        """Nested dissection of the separator system (distributed.separator_offsets: two leaf blocks + the top block); zeros: off."""
        _check(self.L.slide_chol_batch_set_separator_blocks(C.c_void_p(self.h), C.c_int(int(Ta)), C.c_int(int(Tb)), C.c_int(int(used_a)), C.c_int(int(used_b))))

    @staticmethod
    def sep_buffer_len(m, n_relmeas=0):
        L = lib()
        L.slide_chol_batch_sep_buffer_len.restype = C.c_longlong
        return int(L.slide_chol_batch_sep_buffer_len(C.c_int(int(m)), C.c_int(int(n_relmeas))))

    def set_separator_owner(self, leaf, leader=True):
        """Cut passes of a job whose ranks split in two halves along the separator's dissection: this rank factors leaf `leaf` only
        (-1: off); `leader`: the one rank of its half that adds the leaf's Schur complement to the top block's sum."""
        _check(self.L.slide_chol_batch_set_separator_owner(C.c_void_p(self.h), C.c_int(int(leaf)), C.c_int(1 if leader else 0)))

    @staticmethod
    def sep_segment(m, n_relmeas, Ta, Tb, which):
        """(offset, length) in doubles of leaf a (0), leaf b (1) or the top block + lambdas (2) inside the packed exchange buffer."""
        out = (C.c_longlong * 2)()
        _check(lib().slide_chol_batch_sep_segment(C.c_int(int(m)), C.c_int(int(n_relmeas)), C.c_int(int(Ta)), C.c_int(int(Tb)), C.c_int(int(which)), out))
        return int(out[0]), int(out[1])

    @staticmethod
    def sep_exchange_len(m, n_relmeas=0, Ta=0, Tb=0):
        L = lib()
        L.slide_chol_batch_sep_exchange_len.restype = C.c_longlong
        return int(L.slide_chol_batch_sep_exchange_len(C.c_int(int(m)), C.c_int(int(n_relmeas)), C.c_int(int(Ta)), C.c_int(int(Tb))))

    def pass_part(self, buf_ptrs, part):
        """Part 0 / 1 / 2 of the pass cut at its two exchanges (jobs that span GPUs): the caller's all-reduce of buffer 0 goes onto
        stream() between the parts; only part 2 synchronises with the host."""
        arr = (C.c_void_p * len(buf_ptrs))(*[int(p) for p in buf_ptrs])
        return _check(self.L.slide_chol_batch_pass_part(C.c_void_p(self.h), arr, C.c_int(part)))

    def stream(self):
        """hipStream_t (integer) the passes run on."""
        return int(self.L.slide_chol_batch_stream(C.c_void_p(self.h)) or 0)

    def profile(self, buf_ptrs):
        """(ms of the batched step kernels of one un-captured pass, number of step launches)."""
        arr = (C.c_void_p * len(buf_ptrs))(*[int(p) for p in buf_ptrs])
        ms, nl = C.c_double(0), C.c_int(0)
        _check(self.L.slide_chol_batch_profile(C.c_void_p(self.h), arr, C.byref(ms), C.byref(nl)))
        return ms.value, nl.value

    def profile_exact_joint(self, buf_ptrs):
        """Stage times (ms) of one un-captured exact joint pass + block columns of the separator system."""
        arr = (C.c_void_p * len(buf_ptrs))(*[int(p) for p in buf_ptrs])
        out = np.zeros(6)
        ns = C.c_int(0)
        _check(self.L.slide_chol_batch_profile_exact_joint(C.c_void_p(self.h), arr, _p(out), C.byref(ns)))
        names = ("assembly", "band_factorisations", "border_products", "separator_gather", "separator_solve", "back_substitution")
        return dict(zip(names, [float(v) for v in out])), ns.value

    def close(self):
        if self.h:
            self.L.slide_chol_batch_destroy(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        self.close()


class SlideBackend:
    """runSLOAMNode seam (reference src/core/sloamNode.cpp:762-1036) on the MI355X."""

    def __init__(self, params: Params | None = None, num_robots: int = 1):
        self.L = lib()
        self.n_robots = num_robots
        h = self.L.slide_backend_create(C.byref(params) if params is not None else None)
        if not h:
            raise SlideError(f"slide_backend_create failed: {last_error()}")
        self.h = C.c_void_p(h)
        self.graph = SlideGraph(handle=C.c_void_p(self.L.slide_backend_graph(self.h)))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.slide_backend_destroy(self.h)
            self.h = None

    def process_frame(self, robot, rel7, prev7, det, mode=FRAME_HOST):
        nc, nb, ne = len(det["cyl_label"]), len(det["cube_label"]), len(det["ell_label"])
        a = [_d(det["cyl_root"]), _d(det["cyl_ray"]), _d(det["cyl_radius"]), _i(det["cyl_label"]),
             _d(det["cube_pose7"]), _d(det["cube_scale"]), _i(det["cube_label"]),
             _d(det["ell_pose7"]), _d(det["ell_scale"]), _i(det["ell_label"])]
        D = Detections(nc, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), nb, _p(a[4]), _p(a[5]), _p(a[6]), ne, _p(a[7]),
                       _p(a[8]), _p(a[9]))
        cm, bm, em = np.full(nc, -1, np.int32), np.full(nb, -1, np.int32), np.full(ne, -1, np.int32)
        cid, bid, eid = np.full(nc, -1, np.int32), np.full(nb, -1, np.int32), np.full(ne, -1, np.int32)
        R = FrameResult()
        R.cyl_match, R.cube_match, R.ell_match = _p(cm), _p(bm), _p(em)
        R.cyl_id, R.cube_id, R.ell_id = _p(cid), _p(bid), _p(eid)
        rc = self.L.slide_backend_process_frame(self.h, C.c_int(mode), C.c_int(robot), _p(_d(rel7)), _p(_d(prev7)),
                                                C.byref(D), C.byref(R))
        if rc not in (SLIDE_OK, -2):
            _check(rc)
        return dict(status=int(rc), pose7=np.array(R.out_pose7[:]), cyl_match=cm, cube_match=bm, ell_match=em,
                    cyl_id=cid, cube_id=bid, ell_id=eid, t_assoc=R.ms_association * 1e-3, t_graph=R.ms_graph * 1e-3)

    def ingest_solve(self):
        return self.L.slide_backend_ingest_solve(self.h)

    def end_frame(self, robot):
        out = np.zeros(7)
        st = self.L.slide_backend_end_frame(self.h, C.c_int(robot), _p(out))
        return int(st), out

    def counts(self):
        out = np.zeros(4, np.uint64)
        pc = np.zeros(13, np.uint64)
        _check(self.L.slide_backend_counts(self.h, _p(out), _p(pc), C.c_int(13)))
        return dict(cyl=int(out[0]), cube=int(out[1]), point=int(out[2]), factors=int(out[3]),
                    poses=pc[: self.n_robots].astype(np.int64))

    def landmark_table(self, cls):
        n = self.counts()[("cyl", "cube", "point")[cls]]
        xyz = np.zeros((max(n, 1), 3))
        lab = np.zeros(max(n, 1), np.int32)
        k = self.L.slide_backend_landmark_table(self.h, C.c_int(cls), _p(xyz), _p(lab), C.c_int(n))
        if k < 0:
            _check(k)
        return xyz[:k], lab[:k]

    def map_model(self, cls, idx):
        out = np.zeros(7)
        hits, label = C.c_int(0), C.c_int(0)
        st = _check(self.L.slide_backend_map_model(self.h, C.c_int(cls), C.c_int(idx), _p(out), C.byref(hits), C.byref(label)),
                    True)
        return int(st), out[: 7 if cls == 0 else 6], hits.value, label.value


def pair_timeouts():
    """Flag waits of the pair kernel (two block columns per launch) that gave up since the library was loaded; 0 on a healthy run."""
    f = lib().slide_debug_pair_timeouts
    f.restype = C.c_int
    return int(f())


def dense_spd_solve(A, b, repeats=1, method=0):
    """x = A^-1 b on the GPU (blocked FP64-MFMA Cholesky); returns (x, device ms over `repeats` passes).  method 0: one step launch
    per 64-column block; 1: the left-looking persistent factorisation (one launch, flags between workgroups); 2: two block columns
    per launch (k_chol_pair_batched)."""
    A = np.asfortranarray(np.asarray(A, dtype=np.float64))
    n = A.shape[0]
    x = np.zeros(n)
    ms = C.c_double(0)
    _check(lib().slide_dense_spd_solve_ex(A.ctypes.data_as(C.c_void_p), C.c_int(n), _p(_d(b)), _p(x), C.c_int(repeats),
                                          C.byref(ms), C.c_int(method)))
    return x, ms.value


def debug_chol_bordered(S, ld, T, nbr, prof=None, bfirst=None, ord=None, b0=0, kofs=0, n_copies=1, method=0):
    """slide_debug_chol_bordered: factor one bordered system with a profile (S: flat column-major ld x T*64) n_copies times side by side.
    Returns dict(S, Ld, Winv, status, copy_diff)."""
    S = np.ascontiguousarray(S, dtype=np.float64).ravel()
    assert S.size == ld * T * 64
    So = np.zeros_like(S)
    Ld = np.zeros(T * 64 * 64)
    Wi = np.zeros(T * 1024)
    st = np.zeros(8, np.int32)
    diff = C.c_double(0)
    ia = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
    pr, bf, od = ia(prof), ia(bfirst), ia(ord)
    pp = lambda a: None if a is None else _p(a)
    _check(lib().slide_debug_chol_bordered(_p(S), C.c_int(ld), C.c_int(T), C.c_int(nbr), pp(pr), pp(bf), pp(od), C.c_int(b0), C.c_int(kofs),
                                           C.c_int(n_copies), C.c_int(method), _p(So), _p(Ld), _p(Wi), _p(st), C.byref(diff)))
    return dict(S=So, Ld=Ld, Winv=Wi, status=st, copy_diff=diff.value)


# ---- stand-alone association / place recognition ---------------------------------------------------------
def submap_knn(cloud_xyz_f32, query_xyz, K):
    cloud = np.ascontiguousarray(cloud_xyz_f32, dtype=np.float32)
    n = cloud.shape[0]
    out = np.zeros(max(min(K, n), 1), np.int32)
    k = C.c_int(0)
    _check(lib().slide_submap_knn(_p(cloud), C.c_uint64(n), _p(_d(query_xyz)), C.c_int(K), _p(out), C.byref(k)))
    return out[: k.value]


def match_cylinders(root, ray, label, map_root, map_ray, map_label, thresh):
    n, m = len(label), len(map_label)
    out = np.full(max(n, 1), -1, np.int32)
    _check(lib().slide_assoc_match_cylinders(C.c_int(n), _p(_d(root)), _p(_d(ray)), _p(_i(label)), C.c_int(m), _p(_d(map_root)),
                                             _p(_d(map_ray)), _p(_i(map_label)), C.c_double(thresh), _p(out)))
    return out[:n]


def match_boxes(cls, xyz, label, map_xyz, map_label, thresh):
    n, m = len(label), len(map_label)
    out = np.full(max(n, 1), -1, np.int32)
    _check(lib().slide_assoc_match_boxes(C.c_int(cls), C.c_int(n), _p(_d(xyz)), _p(_i(label)), C.c_int(m), _p(_d(map_xyz)),
                                         _p(_i(map_label)), C.c_double(thresh), _p(out)))
    return out[:n]


def assoc_sweep_batch(cloud_xyz_f32, model_xyz, label, query_pos, obs_xyz, obs_label, K, thresh, repeats=1):
    """Batched association sweep (getSubmap K-NN gate + matchEllipsoidModels) of n_query frames against one resident map.
    obs_xyz: (n_query, n_obs, 3), obs_label: (n_query, n_obs).  Returns (map index or -1 per observation, device ms of `repeats`
    launches on resident inputs)."""
    cloud = np.ascontiguousarray(cloud_xyz_f32, dtype=np.float32).reshape(-1, 3)
    model = _d(model_xyz).reshape(-1, 3)
    lab = _i(label)
    qp = _d(query_pos).reshape(-1, 3)
    ox = _d(obs_xyz)
    ol = _i(obs_label)
    nq = qp.shape[0]
    n_obs = ol.shape[1] if ol.ndim == 2 else 0
    out = np.full((nq, max(n_obs, 1)), -1, np.int32)
    ms = C.c_double(0)
    _check(lib().slide_assoc_sweep_batch(_p(cloud), _p(model), _p(lab), C.c_int(len(lab)), _p(qp), _p(ox), _p(ol), C.c_int(nq),
                                         C.c_int(n_obs), C.c_int(K), C.c_double(thresh), _p(out), C.c_int(repeats), C.byref(ms)))
    return out[:, :n_obs], ms.value


def place_default_params(**kw) -> PlaceParams:
    p = PlaceParams()
    lib().slide_place_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def match_maps(ref7, qry7, params: PlaceParams):
    ref7, qry7 = _d(ref7), _d(qry7)
    nr, nq = ref7.shape[0], qry7.shape[0]
    best = np.zeros(3)
    pr, pq = np.full(max(nq, 1), -1, np.int32), np.full(max(nq, 1), -1, np.int32)
    nc = C.c_int64(0)
    inl = lib().slide_match_maps(_p(ref7), C.c_int(nr), _p(qry7), C.c_int(nq), C.byref(params), _p(best), _p(pr), _p(pq),
                                 C.byref(nc))
    if inl < 0 and inl != -10000:
        _check(inl)
    k = max(inl, 0)
    return dict(inliers=int(inl), xyyaw=best, ref_idx=pr[:k], qry_idx=pq[:k], candidates=int(nc.value))


def find_inter_loop_closure(ref7, qry7, params: PlaceParams):
    ref7, qry7 = _d(ref7), _d(qry7)
    tf = np.zeros(16)
    inl = C.c_int(0)
    xyzyaw = np.zeros(4)
    rc = lib().slide_find_inter_loop_closure(_p(ref7), C.c_int(ref7.shape[0]), _p(qry7), C.c_int(qry7.shape[0]),
                                             C.byref(params), _p(tf), C.byref(inl), _p(xyzyaw))
    if rc < 0:
        _check(rc)
    return dict(found=bool(rc), tf=tf.reshape(4, 4), inliers=inl.value, xyzyaw=xyzyaw)


def find_intra_loop_closure(meas7, submap7, query_pose7, candidate_pose7, params: PlaceParams, x_half=5.0, y_half=5.0,
                            yaw_half=10.0 * np.pi / 180.0):
    """PlaceRecognition::findIntraLoopClosure (place_recognition.cpp:389-496); intra half ranges default as :53-63."""
    m, sm = _d(meas7).reshape(-1, 7), _d(submap7).reshape(-1, 7)
    tf = np.zeros(16)
    inl = C.c_int(0)
    xyzyaw = np.zeros(4)
    rc = lib().slide_find_intra_loop_closure(_p(m), C.c_int(len(m)), _p(sm), C.c_int(len(sm)), _p(_d(query_pose7)),
                                             _p(_d(candidate_pose7)), C.byref(params), C.c_double(x_half), C.c_double(y_half),
                                             C.c_double(yaw_half), _p(tf), C.byref(inl), _p(xyzyaw))
    if rc < 0:
        _check(rc)
    return dict(found=bool(rc), tf=tf.reshape(4, 4), inliers=inl.value, xyzyaw=xyzyaw)


def loop_candidate_idx(cloud_xyz, max_dist, pose_idx, at_least_num_of_poses_old):
    """CylinderMapManager::getLoopCandidateIdx (cylinderMapManager.cpp:160-184).  Returns the candidate index or None."""
    cloud = np.ascontiguousarray(cloud_xyz, dtype=np.float32).reshape(-1, 3)
    cand, found = C.c_uint64(0), C.c_int(0)
    _check(lib().slide_loop_candidate_idx(_p(cloud), C.c_int(len(cloud)), C.c_double(max_dist), C.c_uint64(pose_idx),
                                          C.c_uint64(at_least_num_of_poses_old), C.byref(cand), C.byref(found)))
    return int(cand.value) if found.value else None


def clipper_affinity(D1, D2, A, sigma=0.01, epsilon=0.06, mindist=0.0, affinityeps=1e-4):
    D1, D2 = _d(D1), _d(D2)
    A = _i(A)
    m = A.shape[0]
    M = np.zeros((m, m))
    _check(lib().slide_clipper_affinity(_p(D1), C.c_int(D1.shape[0]), _p(D2), C.c_int(D2.shape[0]), C.c_int(D1.shape[1]), _p(A),
                                        C.c_int(m), C.c_double(sigma), C.c_double(epsilon), C.c_double(mindist),
                                        C.c_double(affinityeps), _p(M)))
    return M


class ClipperParams(C.Structure):
    """slide_clipper_params_t (clipper.h:27-60, euclidean_distance.h:24-29)."""
    _fields_ = [("tol_u", C.c_double), ("tol_F", C.c_double), ("maxiniters", C.c_int), ("maxoliters", C.c_int),
                ("beta", C.c_double), ("maxlsiters", C.c_int), ("eps", C.c_double), ("affinityeps", C.c_double),
                ("rescale_u0", C.c_int), ("sigma", C.c_double), ("epsilon", C.c_double), ("mindist", C.c_double)]


def clipper_params(**kw):
    p = ClipperParams()
    lib().slide_clipper_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def clipper_dense_clique(M_upper, u0=None, params=None):
    """CLIPPER::findDenseClique (clipper.cpp:172-323, DSD_HEU).  Returns (nodes, u, score)."""
    M = _d(M_upper)
    n = M.shape[0]
    p = params or clipper_params()
    nodes = np.zeros(max(n, 1), np.int32)
    u = np.zeros(max(n, 1))
    nn, sc = C.c_int(0), C.c_double(0)
    u0a = _d(u0) if u0 is not None else None
    _check(lib().slide_clipper_dense_clique(_p(M), C.c_int(n), _p(u0a) if u0a is not None else None, C.byref(p), _p(nodes),
                                            C.byref(nn), _p(u), C.byref(sc)))
    return nodes[:nn.value].copy(), u[:n].copy(), sc.value


def clipper_last_solve_info():
    """(workgroups, gradient evaluations) of this process's last clipper_dense_clique call: > 1 workgroup = the cooperative solve of
    one large problem (n >= 1024, or SLIDE_CLIPPER_WGS)."""
    w, e = C.c_int(0), C.c_double(0)
    lib().slide_clipper_last_solve_info(C.byref(w), C.byref(e))
    return w.value, e.value


MS_PLACE_SWEEP, MS_TRI_MATCH, MS_CLQ_CSR, MS_CLQ_SOLVE, MS_AFFINITY, MS_CLQ_NNZ, MS_PLACE_PAIR_TESTS, MS_TRI_PAIRS, MS_PLACE_DIST_TESTS = range(9)


def last_device_ms(what):
    """slide_last_device_ms: kernel time (ms) or work count of the last stand-alone SlideMatch / SlideGraph / CLIPPER call."""
    v = C.c_double(0)
    _check(lib().slide_last_device_ms(C.c_int(what), C.byref(v)))
    return v.value


def clipper_dense_clique_batch(Ms, u0s=None, params=None):
    """Several CLIPPER dense-clique problems in one launch (one persistent workgroup per problem).  Ms: list of (n_j, n_j) affinity
    matrices; u0s: list of start vectors or None.  Returns [(nodes, u, score), ...] — what clipper_dense_clique gives for each."""
    J = len(Ms)
    Ma = [_d(M) for M in Ms]
    ns = np.array([M.shape[0] for M in Ma], np.int32)
    p = params or clipper_params()
    nodes = [np.zeros(max(int(k), 1), np.int32) for k in ns]
    us = [np.zeros(max(int(k), 1)) for k in ns]
    u0a = [(_d(u) if u is not None else None) for u in (u0s or [None] * J)]
    PP = C.POINTER(C.c_double)
    Mp = (C.c_void_p * J)(*[M.ctypes.data for M in Ma])
    Up = (C.c_void_p * J)(*[(u.ctypes.data if u is not None else None) for u in u0a])
    Np = (C.c_void_p * J)(*[x.ctypes.data for x in nodes])
    Op = (C.c_void_p * J)(*[x.ctypes.data for x in us])
    nn = np.zeros(max(J, 1), np.int32)
    sc = np.zeros(max(J, 1))
    _check(lib().slide_clipper_dense_clique_batch(C.c_int(J), Mp, _p(ns), Up, C.byref(p), Np, _p(nn), Op, _p(sc)))
    return [(nodes[j][:nn[j]].copy(), us[j][:ns[j]].copy(), float(sc[j])) for j in range(J)]


def match_triangles(tri_model, tri_data, threshold=0.1):
    """semantic_clipper::match_triangles (semantic_clipper.cpp:49-118).  tri_*: (n, 3, 2).  Returns (pts (n_pairs, 3, 4) with
    rows [model x, model y, data x, data y], diffs (n_pairs,))."""
    tm, td = _d(tri_model).reshape(-1, 6), _d(tri_data).reshape(-1, 6)
    n = C.c_int(0)
    _check(lib().slide_match_triangles(_p(tm), C.c_int(len(tm)), _p(td), C.c_int(len(td)), C.c_double(threshold), None, None,
                                       C.c_int(0), C.byref(n)))
    pts = np.zeros((max(n.value, 1), 3, 4))
    diffs = np.zeros(max(n.value, 1))
    _check(lib().slide_match_triangles(_p(tm), C.c_int(len(tm)), _p(td), C.c_int(len(td)), C.c_double(threshold), _p(pts), _p(diffs),
                                       C.c_int(n.value), C.byref(n)))
    return pts[:n.value], diffs[:n.value]


def estimate_tf2d(a_xy, b_xy):
    a, b = _d(a_xy), _d(b_xy)
    tf = np.zeros(9)
    _check(lib().slide_estimate_tf2d(_p(a), _p(b), C.c_int(len(a)), _p(tf)))
    return tf.reshape(3, 3)


def semantic_clipper(tri_model, tri_data, params=None, min_num_pairs=4, matching_threshold=0.1, u0=None):
    """semantic_clipper::run_semantic_clipper (semantic_clipper.cpp:140-274) from the triangle lists on.
    Returns dict(found, tf (4x4 query -> reference), n_putative, n_inliers, inliers)."""
    tm, td = _d(tri_model).reshape(-1, 6), _d(tri_data).reshape(-1, 6)
    p = params or clipper_params()
    tf = np.zeros(16)
    counts = np.zeros(2, np.int32)
    cap = 3 * len(tm) * max(len(td), 1)
    cap = min(cap, 1 << 22)
    inl = np.zeros(max(cap, 1), np.int32)
    found = C.c_int(0)
    u0a = _d(u0) if u0 is not None else None
    _check(lib().slide_semantic_clipper(_p(tm), C.c_int(len(tm)), _p(td), C.c_int(len(td)), C.byref(p), C.c_int(min_num_pairs),
                                        C.c_double(matching_threshold), _p(u0a) if u0a is not None else None,
                                        C.c_int(len(u0a) if u0a is not None else 0), _p(tf), _p(counts), _p(inl), C.c_int(len(inl)),
                                        C.byref(found)))
    return dict(found=bool(found.value), tf=tf.reshape(4, 4), n_putative=int(counts[0]), n_inliers=int(counts[1]),
                inliers=inl[:counts[1]].copy())


def closest_stamp(sec, nsec, qsec, qnsec):
    sec = np.ascontiguousarray(sec, dtype=np.int64)
    nsec = np.ascontiguousarray(nsec, dtype=np.int64)
    idx, diff = C.c_int(0), C.c_double(0)
    lib().slide_closest_stamp(_p(sec), _p(nsec), C.c_int(len(sec)), C.c_int64(qsec), C.c_int64(qnsec), C.byref(idx),
                              C.byref(diff))
    return idx.value, diff.value


def find_relative_meas_match(packets, counters, host, pending):
    """sloam::FindRelativeMeasurementMatch (sloam.cpp:321-412).  packets: per robot list of (sec, nsec); pending: list of
    ((sec, nsec), robot, only_use_odom).  Returns (matches (n, 4) = [tag, index, host idx, other idx], tags of the
    measurements still pending); raises SlideError where the reference throws."""
    sec, ns, off = [], [], [0]
    for pk in packets:
        for (a, b) in pk:
            sec.append(a); ns.append(b)
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
    nm = C.c_int(0)
    _check(lib().slide_find_relative_meas_match(C.c_int(len(packets)), _p(sec), _p(ns), _p(off), _p(pc), C.c_int(host), C.byref(npend),
                                                _p(m_sec), _p(m_ns), _p(m_rob), _p(m_odo), _p(m_tag), _p(out), C.byref(nm)))
    return out.reshape(-1, 4)[:nm.value].copy(), m_tag[:npend.value].copy()


def delaunay_2d(points_xy):
    """Triangle index triples (ascending ids, lexicographic order) of the 2-D Delaunay triangulation (host code)."""
    xy = _d(points_xy)
    n = len(xy)
    cap = max(2 * n, 1)
    tri = np.zeros((cap, 3), np.int32)
    nt = C.c_int(0)
    _check(lib().slide_delaunay_2d(_p(xy), C.c_int(n), _p(tri), C.c_int(cap), C.byref(nt)))
    return tri[:nt.value].copy()


def run_semantic_clipper(ref7, qry7, sigma=0.01, epsilon=0.06, min_num_pairs=4, matching_threshold=0.1, u0=None):
    """semantic_clipper::run_semantic_clipper (semantic_clipper.cpp:140-274) on two object maps (rows [label, x, y, z, d1, d2, d3])."""
    r, q = _d(ref7), _d(qry7)
    tf = np.zeros(16)
    counts = np.zeros(2, np.int32)
    found = C.c_int(0)
    u0a = _d(u0) if u0 is not None else None
    _check(lib().slide_run_semantic_clipper(_p(r), C.c_int(len(r)), _p(q), C.c_int(len(q)), C.c_double(sigma), C.c_double(epsilon),
                                            C.c_int(min_num_pairs), C.c_double(matching_threshold),
                                            _p(u0a) if u0a is not None else None, C.c_int(len(u0a) if u0a is not None else 0),
                                            _p(tf), _p(counts), C.byref(found)))
    return dict(found=bool(found.value), tf=tf.reshape(4, 4), n_putative=int(counts[0]), n_inliers=int(counts[1]))


def pick_next_measurement(odom, obs, rel, latest, current_time, msg_delay_tolerance, min_odom_distance):
    """Input::PickNextMeasurementToAdd (input.cpp:26-108).  odom: list of ((sec, nsec), pose7); obs / rel: lists of (sec, nsec);
    latest: ((sec, nsec), pose7).  Returns (meas_to_add, pop_odom, pop_obs, pop_rel)."""
    def stamps(lst):
        a = np.array([[x[0], x[1]] for x in lst] + [[0, 0]], np.int64)
        return np.ascontiguousarray(a[:, 0]), np.ascontiguousarray(a[:, 1])
    os_, on_ = stamps([o[0] for o in odom])
    op = _d(np.array([o[1] for o in odom] + [[0, 0, 0, 0, 0, 0, 1.0]]))
    bs, bn = stamps(obs)
    rs, rn = stamps(rel)
    out = np.zeros(4, np.int32)
    _check(lib().slide_pick_next_measurement(_p(os_), _p(on_), _p(op), C.c_int(len(odom)), _p(bs), _p(bn), C.c_int(len(obs)), _p(rs), _p(rn),
                                             C.c_int(len(rel)), C.c_int64(latest[0][0]), C.c_int64(latest[0][1]), _p(_d(latest[1])),
                                             C.c_double(current_time), C.c_double(msg_delay_tolerance), C.c_float(min_odom_distance),
                                             _p(out)))
    return tuple(int(v) for v in out)


def in_loop_closure_region(cloud_xyz, pose_xyz, max_dist_xy=10.0, max_dist_z=2.0, at_least_num_of_poses_old=30):
    cloud = np.ascontiguousarray(cloud_xyz, dtype=np.float32).reshape(-1, 3)
    inside = C.c_int(0)
    _check(lib().slide_in_loop_closure_region(_p(cloud), C.c_int(len(cloud)), _p(_d(pose_xyz)), C.c_double(max_dist_xy),
                                              C.c_double(max_dist_z), C.c_uint64(at_least_num_of_poses_old), C.byref(inside)))
    return bool(inside.value)
