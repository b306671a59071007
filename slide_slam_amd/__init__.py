"""slide_slam_amd — MI355X-native (gfx950, HIP) drop-in for the SlideSLAM `backend/sloam` hot path:
per-frame semantic data association + factor-graph linearise-and-solve, behind the C-ABI of
include/slide_gpu.h.  See DESIGN.md.  There is no CPU fallback."""
from .api import (assoc_sweep_batch, CHART_CAYLEY, CHART_EXPMAP, CLS_CUBE, CLS_CYLINDER, CLS_ELLIPSOID, FRAME_FOREIGN, FRAME_HOST,  # noqa: F401
                  FRAME_HOST_DEFERRED, LIB_PATH, CholBatch, ClipperParams, Params, PlaceParams, SlideBackend, SlideError, SlideGraph,
                  clipper_affinity, clipper_dense_clique, clipper_dense_clique_batch, clipper_last_solve_info, clipper_params, closest_stamp, delaunay_2d, estimate_tf2d, find_relative_meas_match, in_loop_closure_region, match_triangles,
                  pick_next_measurement, run_semantic_clipper,
                  semantic_clipper, default_params, dense_spd_solve, pair_timeouts, device_check, find_inter_loop_closure, find_intra_loop_closure, lib, loop_candidate_idx, match_boxes,
                  match_cylinders, match_maps, place_default_params, submap_knn)
from . import api  # noqa: E402,F401
