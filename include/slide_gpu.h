/*
 * slide_gpu.h — C-ABI boundary of the MI355X-native SlideSLAM backend hot path.
 *
 * The reference (lunarlab-gatech/SLIDE_SLAM, backend/sloam) has no plugin / FFI interface for this
 * path: the seam is a set of in-process C++ class methods called from SLOAMNode
 * (backend/sloam/src/core/sloamNode.cpp).  Every entry point below names the reference method it
 * replaces (paths relative to backend/sloam/).  All types are POD, all matrices row-major, poses are
 *     pose7 = tx, ty, tz, qx, qy, qz, qw      (geometry_msgs/Pose order; the trajectory file order
 *                                              of sloamNode.cpp:318-337)
 * and always mean T_world<-sensor ("tf_sensor_to_map", include/factorgraph/graph.h:44).
 * Functions return an int status, never throw across the ABI, and every handle serialises its own
 * calls with an internal mutex (the reference leaves add_* vs solve() unsynchronised).
 *
 * There is NO CPU fallback behind this header: every compute entry point runs hand-written HIP
 * kernels on a gfx950 device and fails with SLIDE_ERR_HIP when none is usable.
 */
#ifndef SLIDE_GPU_H_
#define SLIDE_GPU_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLIDE_OK 0
#define SLIDE_MISSING 1          /* key absent: getPose() == false semantics (graph.cpp:297-311) */
#define SLIDE_ERR_INVALID (-1)   /* bad argument / robot id outside [0, SLIDE_MAX_ROBOTS) */
#define SLIDE_ERR_NOT_SPD (-2)   /* a landmark block or the reduced pose system is not positive definite */
#define SLIDE_ERR_CAPACITY (-3)  /* a fixed on-chip capacity was exceeded (e.g. K > 16384 neighbours in the K-NN gate; the map itself is unbounded) */
#define SLIDE_ERR_HIP (-4)       /* HIP runtime error or no gfx950 device */
#define SLIDE_ERR_RUNTIME (-5)   /* the reference would throw std::runtime_error here (sloam.cpp:349,355); also: a device-side wait gave up (scheduling stall) */

#define SLIDE_MAX_ROBOTS 13      /* include/factorgraph/graph.h:11 */

#define SLIDE_CHART_CAYLEY 0     /* GTSAM 4.0.3 default Pose3 chart (cubeFactor.h:96-97) */
#define SLIDE_CHART_EXPMAP 1     /* GTSAM_POSE3_EXPMAP builds */

#define SLIDE_CLS_CYLINDER 0
#define SLIDE_CLS_CUBE 1
#define SLIDE_CLS_ELLIPSOID 2    /* ellipsoids become Point3 landmarks in the graph (graphWrapper.cpp:157-202) */

/* Parameter names/defaults mirror the rosparam reads of the reference:
 * graphWrapper.cpp:31-34,55,60,63-64 ; graph.cpp:15-17 ; graph.h:125 ; sloamNode.cpp:151-153 ;
 * {cylinder,cube,ellipsoid}MapManager.cpp K = 50 / 30 / 1000. */
typedef struct slide_params_t {
  int pose_chart;                 /* SLIDE_CHART_* */
  double relinearize_threshold;   /* 0.1 */
  double noise_floor;             /* 0.01 */
  double noise_model_prior_first_pose_vec[6]; /* 1e-6 */
  double noise_model_odom_vec[6];             /* 0.1  */
  double noise_model_cube_vec[9];             /* 0.1  */
  double noise_model_rel_meas_vec[6];         /* 0.1  */
  double cylinder_sigma;          /* 400 */
  double bearing_range_sigma;     /* 1   */
  double numdiff_delta;           /* 1e-6 (cubeFactor.cpp:43,48) */
  double cylinder_match_thresh;   /* 2.0 */
  double cuboid_match_thresh;     /* 2.0 */
  double ellipsoid_match_thresh;  /* 0.75 */
  int knn_cylinder, knn_cube, knn_ellipsoid;  /* 50, 30, 1000 */
  int number_of_robots;           /* <= SLIDE_MAX_ROBOTS */
  int device;                     /* HIP device ordinal, -1 = current */
} slide_params_t;

void slide_default_params(slide_params_t* p);
/* 0 when a gfx950 device is present and usable, SLIDE_ERR_HIP otherwise (message via slide_last_error). */
int slide_device_check(int device);
const char* slide_last_error(void);
const char* slide_version(void);

/* ------------------------------------------------------------------------------------------------
 * S1 — SemanticFactorGraph (include/factorgraph/graph.h:70-121, src/factorgraph/graph.cpp)
 * ---------------------------------------------------------------------------------------------- */
typedef struct slide_graph slide_graph_t;
typedef struct slide_backend slide_backend_t;

slide_graph_t* slide_graph_create(const slide_params_t* p);          /* SemanticFactorGraph() graph.cpp:14-22 */
void slide_graph_destroy(slide_graph_t* g);

int slide_graph_set_prior(slide_graph_t* g, int robot, const double pose7[7]);              /* setPriors graph.cpp:24-42 */
int slide_graph_add_keypose_between(slide_graph_t* g, int robot, uint64_t from_idx, uint64_t to_idx,
                                    const double rel7[7], const double est7[7]);            /* addKeyPoseAndBetween graph.cpp:44-151 */
int slide_graph_add_loop_closure(slide_graph_t* g, const double rel7[7], uint64_t from_idx, int from_robot,
                                 uint64_t to_idx, int to_robot);                            /* addLoopClosureFactor graph.cpp:233-245 */
int slide_graph_add_relative_meas(slide_graph_t* g, const double rel7[7], uint64_t from_idx, int from_robot,
                                  uint64_t to_idx, int to_robot);                           /* addRelativeMeasFactor graph.cpp:247-258 */
int slide_graph_add_point_landmark(slide_graph_t* g, uint64_t lm_idx, const double xyz[3]); /* addPointLandmarkKey graph.cpp:153-156 */
int slide_graph_add_range_bearing(slide_graph_t* g, int robot, uint64_t pose_idx, uint64_t lm_idx,
                                  const double bearing[3], double range);                   /* addRangeBearingFactor graph.cpp:158-180 */
int slide_graph_add_cube(slide_graph_t* g, int robot, uint64_t pose_idx, uint64_t cube_idx, const double pose_world7[7],
                         const double cube_world7[7], const double scale[3], int already_exists); /* addCubeFactor graph.cpp:198-231 */
int slide_graph_add_cylinder(slide_graph_t* g, int robot, uint64_t pose_idx, uint64_t cyl_idx,
                             const double pose_world7[7], const double root[3], const double ray[3], double radius,
                             int already_exists);                                           /* addCylinderFactor graph.cpp:182-196 */
/* One iSAM2-equivalent update: isam->update(fgraph, fvalues); currEstimate = calculateEstimate()  (graph.cpp:260-272). */
int slide_graph_solve(slide_graph_t* g);
/* n batch Gauss-Newton iterations over the whole graph (relinearise every variable each time): the
 * "ms per Gauss-Newton iteration" unit of BASELINE.json.  Equivalent to solve() with threshold 0. */
int slide_graph_gauss_newton(slide_graph_t* g, int iterations);

int slide_graph_get_pose(slide_graph_t* g, int robot, uint64_t idx, double out7[7]);        /* getPose graph.cpp:290-312: SLIDE_MISSING + identity */
int slide_graph_get_pose12(slide_graph_t* g, int robot, uint64_t idx, double out12[12]);    /* same, R row-major (9) + t (3) */
int slide_graph_get_all_poses(slide_graph_t* g, int robot, double* out7N, uint64_t cap, uint64_t* n_out); /* getAllPoses graphWrapper.cpp:313-338 */
/* cls = SLIDE_CLS_*: cylinder -> 7 (root, ray, radius), cube -> 15 (R, t, scale), ellipsoid/point -> 3.
 * getCylinder / getCube / getCentroidLandmark graph.cpp:274-288 (absent key: zeros + SLIDE_MISSING). */
int slide_graph_get_landmark(slide_graph_t* g, int cls, uint64_t idx, double* out);
/* getPoseCovariance graph.cpp:314-323 (isam->marginalCovariance(X(idx))): 6x6 row-major, tangent order [rot, trans], at the
 * linearisation point of the last solve (a forward substitution with six right-hand sides on the resident Cholesky factor).
 * SLIDE_MISSING for an unknown pose, SLIDE_ERR_INVALID before the first solve. */
int slide_graph_get_pose_covariance(slide_graph_t* g, int robot, uint64_t idx, double cov36[36]);
/* counts: [poses, landmarks, factors, relinearised vars in the last solve, chol dim] */
int slide_graph_stats(slide_graph_t* g, int64_t out5[5]);
/* Sum of squared whitened residuals of every factor at the current estimate (= 2 x gtsam::NonlinearFactorGraph::error of the graph
 * ISAM2 holds): out4 = {total, prior factors, Between factors, landmark factors}.  Commits delta into the linearisation point and
 * relinearises (the estimate itself does not move); pending factors are merged first. */
int slide_graph_chi2(slide_graph_t* g, double out4[4]);
/* isam->update(fgraph, fvalues) (graph.cpp:262) throws when a factor names a key that is in neither the graph nor fvalues, and when a
 * value is inserted under a key that exists already.  Here such an entry is refused, the rest of the update is merged, and the call
 * that consumed it (solve, gauss_newton, dist_phase 0 / 20, set_shared, set_ghosts, chol_batch_pass) returns SLIDE_ERR_INVALID with the
 * first offending key in slide_last_error().  This returns the number of entries refused since the graph was created (0 in a healthy run). */
int64_t slide_graph_rejected_count(slide_graph_t* g);
/* Per-kernel device timings (HIP events on the launch stream) of the solves since the last reset.
 * names: caller buffer of n_max * 32 chars; ms_total / launches: n_max entries.  Returns #entries. */
int slide_graph_set_profiling(slide_graph_t* g, int on);
int slide_graph_get_profile(slide_graph_t* g, char* names, double* ms_total, int64_t* launches, int n_max);

/* ---- one robot per GPU (the reference instead keeps a full replica of every robot's graph in
 * every sloam_node and gossips packets over ROS topics, databaseManager.cpp:219-279) -------------------------
 * Slots are a global, rank-independent enumeration of the landmarks observed by more than one robot.
 * slide_graph_set_shared: slot i is this rank's landmark (cls[i], idx[i]) or none (cls[i] < 0); owner[i] != 0 on
 * exactly one rank per slot (the rank whose value every replica adopts).
 * slide_graph_dist_phase: one distributed Gauss-Newton pass is
 *     phase 0 ; all-reduce(sum) of d_buf[54 * n_slots] ; phase 1 ; all-reduce(sum) of d_buf[9 * n_slots] ; phase 2
 * and a value broadcast is  phase 10 ; all-reduce(sum) of d_buf[15 * n_slots] ; phase 11.
 * d_buf is a DEVICE buffer owned by the caller (the collective itself — RCCL over xGMI — runs outside this
 * library); each call returns after its kernels have completed. */
int slide_graph_set_shared(slide_graph_t* g, const int32_t* cls, const int64_t* idx, const int32_t* owner, int n_slots);
int slide_graph_dist_phase(slide_graph_t* g, int phase, double* d_buf);

/* Several robot graphs on ONE GPU (the 8 / N-robots-per-GPU layout; no counterpart in the reference, whose replica solves one joint
 * graph): the dense factor + solve of phase 1 of all joined graphs runs as one launch sequence, one launch per block column for
 * all of them.  Every joined graph must then run its passes in lockstep from its own host thread (slide_graph_dist_phase(g, 1, ..)
 * returns when all have arrived; SLIDE_ERR_RUNTIME after 60 s).  slide_graph_join_chol_batch(g, NULL, 0) leaves the batch. */
/* Ownership: a batch does not own its graphs, a graph does not own its batch.  slide_graph_destroy / slide_backend_destroy leave the
 * batch first; slide_chol_batch_destroy sends every joined graph back to its own launches.  Neither may run while a pass of the batch
 * is executing on another thread. */
typedef struct slide_chol_batch slide_chol_batch_t;
slide_chol_batch_t* slide_chol_batch_create(int n_graphs);      /* 1 .. 8 */
void slide_chol_batch_destroy(slide_chol_batch_t* b);
int slide_graph_join_chol_batch(slide_graph_t* g, slide_chol_batch_t* b, int slot);
/* One distributed pass (phases 0, 1, 2 of slide_graph_dist_phase) when the batch holds every robot of the job: the two exchanges are
 * device-side sums between the joined graphs' buffers (d_buf: this graph's exchange buffer, 54 doubles per shared slot), no host
 * synchronisation inside the pass.  Called from every joined graph's thread in lockstep. */
int slide_graph_dist_pass_local(slide_graph_t* g, double* d_buf);
/* The same pass for ALL graphs of the batch from ONE host thread: every robot's phases on its own stream, forked from and joined to
 * the batch's stream around the exchanges and the batched factor + solve, captured once and replayed as one hipGraph per pass.
 * d_bufs[i] = exchange buffer (device) of the graph in slot i. */
int slide_chol_batch_pass(slide_chol_batch_t* b, double* const* d_bufs);
/* The same pass for a job that spans GPUs, cut at its two exchanges (8 / N robots on each of N GPUs): every part is a captured
 * hipGraph replayed on the batch's stream.
 *   part 0: phase 0 of every robot + the local sum -> every local buffer holds this GPU's sum of the 54-doubles-per-slot blocks;
 *           the caller all-reduces d_bufs[0][0 .. 54 n_slots) across the ranks ON slide_chol_batch_stream() (RCCL), no host sync;
 *   part 1: d_bufs[0] back to the other local buffers, Schur + batched factor + solve + t_l, local sum of 9 doubles per slot;
 *           the caller all-reduces d_bufs[0][0 .. 9 n_slots);
 *   part 2: d_bufs[0] back to the others, landmark back-substitution, retract; ONE host synchronisation, status decoded.
 * slide_chol_batch_stream: the hipStream_t (as void*) the parts run on. */
int slide_chol_batch_pass_part(slide_chol_batch_t* b, double* const* d_bufs, int part);
void* slide_chol_batch_stream(slide_chol_batch_t* b);
/* Joint Gauss-Newton step over the robots (what the reference's replica computes: solve() on ONE graph holding every robot,
 * graph.cpp:260-272 + sloamNode.cpp:912-1002).  With iterations = 0 a pass solves every robot's own reduced pose system only — block
 * Jacobi over robots, which stops converging once the robots share more than a few landmarks.  With iterations > 0 the pass runs
 * that many preconditioned conjugate-gradient iterations on the GLOBAL reduced system after the factorisations (the robots' factors
 * are the preconditioner; the coupling through the shared landmarks is applied matrix-free), two all-reduces per iteration
 * (9 doubles per shared slot, then 2 scalars).  A cut pass then reads
 *     part 0 | AR 54 n | part 1 | { AR 9 n | part 10 | AR 2 | part 11 (12 on the last iteration) } x iterations | AR 9 n | part 2
 * and the un-batched path (slide_graph_set_pcg on every graph)
 *     phase 0 | AR 54 n | phase 1 | { AR 9 n | phase 31 | AR 2 | phase 32 (33 last) } x iterations | AR 9 n | phase 2
 * with every all-reduce on the first max(1, ..) doubles of the exchange buffer. */
int slide_chol_batch_set_pcg(slide_chol_batch_t* b, int iterations);
int slide_graph_set_pcg(slide_graph_t* g, int iterations);
/* Residual-based end of that iteration: `iterations` stays the upper bound per pass; with tol > 0 the solve counts as converged once
 * gamma = r^T M^-1 r has fallen to tol^2 times its value in the first iteration (never later than at eps^2) — the remaining iterations
 * of the pass are no-ops (alpha = beta = 0), so captured launch sequences keep their shape and every rank takes the same decision from
 * the same all-reduced scalars.  Call before slide_*_set_pcg or repeat that call. */
int slide_chol_batch_set_pcg_tolerance(slide_chol_batch_t* b, double tol);
int slide_graph_set_pcg_tolerance(slide_graph_t* g, double tol);
/* EXACT joint Gauss-Newton step over the robots ("arrow" solve) — what the reference's replica computes with ONE solve() on a graph
 * holding every robot (graph.cpp:260-272 on the replica of sloamNode.cpp:912-1002), without an inner iteration.  The landmarks of the
 * shared slots are not eliminated into the robots' pose systems: they stay as the SEPARATOR of the joint graph.  Every robot eliminates
 * its private landmarks, factors its banded pose system with the separator's coupling rows riding below the band (the step kernels),
 * and forms its Schur complement onto the separator with one FP64-MFMA product; the sum over the robots — ONE all-reduce(sum) of the
 * separator buffer per pass for a job that spans GPUs — is the Schur complement of the JOINT graph onto the shared landmarks; every rank
 * factors it (dense, the same step kernels), substitutes back through its bands and retracts.
 *   slide_graph_set_separator: off[i] = offset of shared slot i's tangent coordinates (cylinder 7, cube 9, point 3) in the separator
 *     system — any non-overlapping layout —, off[n_slots] = its dimension; n = n_slots + 1 entries, identical on every rank; after
 *     slide_graph_set_shared.
 *   slide_chol_batch_set_exact_joint(b, 1, sep_buf, len): passes of the batch take the exact joint step (batched passes only; the PCG
 *     setting is ignored).  sep_buf: the caller's device buffer of slide_chol_batch_sep_buffer_len(m) doubles, m = off[n_slots], in which
 *     part 0 of a cut pass leaves this GPU's partial sum of the separator system (packed: the lower tile columns only) and from which
 *     part 2 takes the all-reduced sum; NULL when the job is this process alone (whole passes only).
 * A cut pass then reads   part 0 | all-reduce(sum) of sep_buf[0 .. len) on slide_chol_batch_stream() | part 2.
 * With inter-robot relative-pose factors (slide_graph_set_ghosts) every pass, cut or whole, PCG or exact, opens with the refresh of the
 * ghost poses; a cut pass reads   part 20 | all-reduce(sum) of d_bufs[0][0 .. 12 n_ghost_slots) | part 0 | ...  (the exchange buffers
 * must hold 12 doubles per ghost slot).  The cross block of such a factor is left out of the step (gradient exact). */
int slide_graph_set_separator(slide_graph_t* g, const int32_t* off, int n);
int slide_chol_batch_set_exact_joint(slide_chol_batch_t* b, int on, double* sep_buf, long long len);
/* Tile-level profile of the separator system (64-coordinate tiles, landmark part): prof[c] = last tile row of tile column c that can be
 * non-zero, monotone, c <= prof[c] < n.  Two shared landmarks couple there only if some robot observes both, so with the slots'
 * coordinates laid out along the robots' adjacency (slide_graph_set_separator takes any layout) the system is block-banded; the caller
 * knows every robot's observer set (the cross-robot association), the batch only its own robots'.  Not set: dense. */
int slide_chol_batch_set_separator_profile(slide_chol_batch_t* b, const int32_t* prof, int n);
/* Nested dissection of the separator system itself.  The robots split into two sets; a shared landmark seen only by robots of one set
 * couples with none seen only by robots of the other, so the layout puts those two "leaf" blocks first (Ta, Tb tile columns; each
 * starts at a tile boundary, used_a / used_b coordinates of it carry slots, the rest of its last tile is padding and gets a unit
 * diagonal) and the landmarks seen from both sets ("top" block) behind them.  The leaves are factored side by side — half the serial
 * chain of block columns, and no arithmetic on the structural zeros between them — with the top block's rows riding as their border
 * (as the lambda rows do), then the top block.  The profile (above) must end each leaf at its own last tile row.  Same step: only the
 * elimination order changes.  (0, 0, 0, 0): not dissected (default).  The reference has no counterpart: its replica solves the joint
 * system inside GTSAM (graph.cpp:260-272), whose elimination order is COLAMD's. */
int slide_chol_batch_set_separator_blocks(slide_chol_batch_t* b, int Ta, int Tb, int used_a, int used_b);
/* A job that spans GPUs whose ranks split in two halves ALONG the dissection (the robots of the first half of the ranks see leaf a and
 * the top block only, those of the second half leaf b and the top block): this rank owns leaf `leaf` (0 / 1; -1: none, the default).  It
 * factors that leaf only, and only the top block of the separator system crosses between the halves.  A cut pass then runs
 *   [20 | AR ghosts |] part 0 | all-reduce of the OWN leaf's segment of sep_buf within the own half (nothing when the half is one rank) |
 *   part 1 | all-reduce of the top segment over all ranks | part 2
 * (slide_chol_batch_sep_segment gives the segments' offsets and lengths in doubles: which = 0 leaf a, 1 leaf b, 2 top block + lambdas).
 * leader: non-zero on ONE rank of each half — the one that adds the leaf's Schur complement to the top block's sum.  C4 on two GPUs:
 * 5 MB cross the link per pass instead of 45. */
int slide_chol_batch_set_separator_owner(slide_chol_batch_t* b, int leaf, int leader);
int slide_chol_batch_sep_segment(int m, int n_relmeas, int Ta, int Tb, int which, long long out2[2]);
/* Nested dissection of every robot's own pose chain inside an exact joint pass: the banded pose system of a robot is a serial chain of
 * block columns (one launch each); cut into n_seg segments at windows of poses as wide as the band is (every coupling across a window
 * passes through it), the segments are factored side by side as systems of their own, and the windows' poses — moved into the border
 * next to the shared landmarks — are eliminated at a second level before the robot's Schur complement joins the separator system.
 * Same step (the elimination order changes, not the system); n_seg = 1 (default) factors every band as one chain; at most 8. */
int slide_chol_batch_set_segments(slide_chol_batch_t* b, int n_seg);
/* The cut of this graph's band: returns the number of segments (1: not cut — the batch does not ask for it, or the chain is too short);
 * out[2 i], out[2 i + 1] = tile columns [t0, t1) of segment i, out[2 n] = number of separator poses (cap >= 2 n + 1 ints). */
int slide_graph_get_segments(slide_graph_t* g, int* out, int cap);
/* Which border tile rows are non-zero in which segment of the cut band, and from which block column on (measurement aid: the flop count
 * of the steps and of the border product follows from it).  Returns the table's length (0: the band is not cut) and fills out[0 .. cap):
 * nseg, the segments' last block columns + 1, then per segment nbr + 1 ints = the first block column of every border tile row and of the
 * right-hand side (1 << 30: the row is all-zero in that segment). */
int slide_graph_get_segment_table(slide_graph_t* g, int* out, int cap);
long long slide_chol_batch_sep_buffer_len(int m, int n_relmeas);
/* Doubles at the head of that buffer a cut pass has to all-reduce: all of it, less the block between the two leaves of a dissected
 * layout (slide_chol_batch_set_separator_blocks: Ta x Tb tiles that are structurally zero and left out of the packed layout). */
long long slide_chol_batch_sep_exchange_len(int m, int n_relmeas, int Ta, int Tb);
/* Inter-robot relative-pose factors in an exact joint pass: the factor between pose a of robot A and pose b of robot B is the rank-6
 * term U U^T, U = [J_a^T; J_b^T], of the joint normal equations; it is carried as six further separator coordinates "lambda" (the
 * factor's linearised residual) of the bordered system [H_rest U; U^T -I] [delta; lambda] = [b; -r]: every robot couples to them through
 * its OWN Jacobian only, and the step is exactly the joint replica's.  ids[i] = index, in the job's list of n_total measurements
 * (identical on every rank), of this graph's i-th ghost factor (slide_graph_add_relative_meas_ghost order); the ghost poses are still
 * refreshed at the start of every pass: they are where the factor is linearised. */
int slide_graph_set_ghost_ids(slide_graph_t* g, const int32_t* ids, int n, int n_total);
/* Scalars of the last joint solve this graph took part in: out8 = {gamma of the last but one iteration, alpha of it, alpha, beta of the
 * last iteration, gamma of the FIRST iteration, gamma of the last; 0, 0} with gamma = r^T M^-1 r (M = the robots' own factors), summed
 * over all robots: gamma_last / gamma_first is the squared reduction of the preconditioned residual. */
int slide_graph_get_pcg_stats(slide_graph_t* g, double out8[8]);
/* Tile-level profile (envelope) of the reduced pose system the solver works in (64 x 64 tiles; the reference's sparse elimination,
 * ISAM2 graph.cpp:260-272, exploits the same structure): returns T, the number of block columns, and writes prof[c] = the last tile row
 * of block column c that can be non-zero in the factor (c <= prof[c] < T, monotone) for c < min(T, cap).  Pending additions are
 * merged first.  slide_graph_set_dense_profile(g, 1) makes the solver ignore the structure (every tile of the lower triangle: the
 * GEMM-shaped extreme, a measurement aid); results are the same either way. */
int slide_graph_get_tile_profile(slide_graph_t* g, int* prof, int cap);
/* Incremental re-factorisation of slide_graph_solve (ISAM2::update re-eliminates only the part of the Bayes tree the new factors and the
 * relinearised variables touch, graph.cpp:260-272; here: the block columns of the banded reduced system from the first dirty one on —
 * the columns before it keep the factor of the last solve, the re-assembled trailing tiles catch up with their panels in one product,
 * the step kernels run on the trailing sub-matrix).  out4 = {updates that re-factored a suffix only, updates that re-factored
 * everything, first re-factored block column of the last update, block columns}.  slide_graph_set_incremental(g, 0) (or
 * SLIDE_NO_INCREMENTAL=1 for the whole process) re-factors everything at every update: the same result up to the summation order of
 * the kept columns' panels. */
int slide_graph_get_incremental_stats(slide_graph_t* g, int64_t out4[4]);
/* iSAM2's bounded back-substitution (ISAM2GaussNewtonParams::wildfireThreshold — 1e-3 in the GTSAM 4.0.3 the reference links, used by
 * every ISAM2::update + calculateEstimate() of SemanticFactorGraph::solve, graph.cpp:15-18, 260-272): on an incremental update a block
 * of the reduced system below the first re-factored block column keeps the last solve's solution when every block it depends on moved
 * by less than `threshold` (infinity norm), and so does everything below it; the chained substitution then ends there.  threshold 0 (the
 * default): off — every update solves its linear system exactly, which is what the parity tests compare.  out2 = {blocks kept over all
 * updates, blocks kept by the last update}. */
int slide_graph_set_wildfire(slide_graph_t* g, double threshold);
int slide_graph_get_wildfire_stats(slide_graph_t* g, int64_t out2[2]);
int slide_graph_set_incremental(slide_graph_t* g, int on);
int slide_graph_set_dense_profile(slide_graph_t* g, int on);
/* Exact joint step: the border of this graph's reduced system — returns the number of border row tiles (64 separator coordinates each;
 * 0 when the graph's batch does not run exact joint passes) and writes first[i] = the first block column of the band in which border
 * tile row i can be non-zero (the border product skips the columns before it) for i < min(n, cap). */
int slide_graph_get_border_profile(slide_graph_t* g, int* first, int cap);
/* Measurement aid: the same pass issued without the graph, HIP events around the batched step kernels; *ms_steps = their device time
 * (launch gaps included), *n_launches = their number. */
int slide_chol_batch_profile(slide_chol_batch_t* b, double* const* d_bufs, double* ms_steps, int* n_launches);
/* The same for an exact joint pass, stage by stage (HIP events on the pass's stream): out6 = ms of {assembly (relinearisation ..
 * borders), the bands' factorisations, the border products (k_border_syrk), the separator gather, the separator's factorisation +
 * solve, the back-substitutions}; *n_sep_steps = block columns of the separator system. */
int slide_chol_batch_profile_exact_joint(slide_chol_batch_t* b, double* const* d_bufs, double out6[6], int* n_sep_steps);
/* Sharded mode, inter-robot relative-pose factors (addRelativeMeasFactor graph.cpp:247-258 between poses of two ranks).
 * Ghost slots enumerate, identically on every rank, the poses such factors touch; slot i is this rank's pose
 * (own_robot[i], own_idx[i]) or belongs to another rank (own_robot[i] < 0).  A factor is added on BOTH ranks, each with its
 * own pose as the variable and the other pose as ghost slot; local_first = 1 when the local pose is the Between's first key.
 * Per pass, before phase 0: dist_phase 20 (pack 12 doubles per slot), all-reduce(sum), dist_phase 21 (adopt). */
int slide_graph_set_ghosts(slide_graph_t* g, const int32_t* own_robot, const int64_t* own_idx, int n_slots);
int slide_graph_add_relative_meas_ghost(slide_graph_t* g, const double rel7[7], uint64_t idx, int robot, int ghost_slot, int local_first);
/* Landmark table of this rank (input of the cross-robot association): for class cls, positions (xyz of the
 * current estimate; cylinders: root) and labels of landmarks [0, n).  Returns n (<= cap). */
int slide_backend_landmark_table(slide_backend_t* b, int cls, double* xyz, int32_t* label, int cap);

/* The dense kernel behind solve(): x = A^-1 b for a symmetric positive definite A (n x n, row- or
 * column-major: only the lower triangle in column-major sense, A[i + j*n] with i >= j, is read) by
 * the blocked FP64-MFMA Cholesky the reduced pose system uses (the reference delegates this to
 * GTSAM's CHOLESKY factorisation, graph.cpp:15).  ms_out (may be NULL): device time of `repeats`
 * factor+solve passes measured with HIP events on the launch stream. */
int slide_dense_spd_solve(const double* A, int n, const double* b, double* x, int repeats, double* ms_out);
/* The same solve by an explicit schedule of the factorisation: method 0 = one step launch per 64-column block (what
 * slide_dense_spd_solve runs), 1 = the left-looking persistent factorisation (ONE launch for all block columns, flags between
 * workgroups instead of kernel boundaries; opt-in for the exact joint passes, SLIDE_CHOL_LL), 2 = TWO block columns per launch
 * (k_chol_pair_batched: every type-A workgroup carries the sub-diagonal tile and the next diagonal block as well, so that the second
 * column's chain starts inside the launch — what the exact joint passes use; ISAM2Params::CHOLESKY of graph.cpp:15 all the same). */
int slide_dense_spd_solve_ex(const double* A, int n, const double* b, double* x, int repeats, double* ms_out, int method);
/* Unit-test hook of the same kernels on a BORDERED system with a profile — one system exactly as an exact joint pass hands it to the
 * factorisation (a robot's band segment with the separator's coupling rows riding below it, a leaf of the separator system: DESIGN.md 3):
 * S_in (ld x T*64 doubles, column-major): the band's lower tiles inside the profile, nbr border row tiles from tile row b0 on (0: right
 * behind the band), the right-hand-side row tile behind them.  prof: T ints (last tile row of every block column inside the profile)
 * or NULL = dense; bfirst: nbr non-decreasing ints = first block column (+ kofs) of the j-th border row in the order `ord` lists them
 * (1 << 30: never), or NULL = every row from column 0; ord: nbr ints (the j-th active row is tile row b0 + ord[j]) or NULL.  The system
 * is factored n_copies (<= 32) times side by side in one launch sequence; S_out / Ld_out (T x 64 x 64) / Winv_out (T x 1024) /
 * status_out (8) receive copy 0, *max_copy_diff the largest difference of any other copy from it.  method 0: one step launch per block
 * column, 2: two block columns per launch.  After the call the band tiles hold L, the border rows W = B L^-T, the RHS row y = L^-1 b. */
int slide_debug_chol_bordered(const double* S_in, int ld, int T, int nbr, const int* prof, const int* bfirst, const int* ord, int b0, int kofs,
                              int n_copies, int method, double* S_out, double* Ld_out, double* Winv_out, int* status_out, double* max_copy_diff);
/* LDS flag waits of the pair kernel (method 2) that gave up since the library was loaded: 0 unless the kernel is broken (its waits are
 * bounded so that every wave reaches the end of its launch); the tests assert 0. */
int slide_debug_pair_timeouts(void);

/* ------------------------------------------------------------------------------------------------
 * S3 — association (include/core/sloam.h:88-108, src/core/sloam.cpp:73-306; *MapManager::getSubmap)
 * Stand-alone entry points on caller-provided host buffers (copied to HBM, kernels run, results copied back).
 * ---------------------------------------------------------------------------------------------- */
/* *MapManager::getSubmap candidate gate (cubeMapManager.cpp:36-75 etc.): exact float32 K-NN of the
 * first-seen cloud to the query position; out_idx = map indices nearest first, *out_k = min(K, n). */
int slide_submap_knn(const float* cloud_xyz, uint64_t n, const double query_xyz[3], int K, int32_t* out_idx, int* out_k);
/* sloam::matchModels sloam.cpp:73-111 (cylinders; objects in the world frame). out = submap index or -1. */
int slide_assoc_match_cylinders(int n_cur, const double* root, const double* ray, const int32_t* label, int n_map,
                                const double* map_root, const double* map_ray, const int32_t* map_label, double thresh,
                                int32_t* out_idx);
/* sloam::matchCubeModels :113-156 (cls = SLIDE_CLS_CUBE, labels ignored) and
 * sloam::matchEllipsoidModels :158-203 (cls = SLIDE_CLS_ELLIPSOID, label-gated). */
int slide_assoc_match_boxes(int cls, int n_cur, const double* xyz, const int32_t* label, int n_map, const double* map_xyz,
                            const int32_t* map_label, double thresh, int32_t* out_idx);
/* Batched association sweep (roofline leg): n_query independent frames, each n_obs ellipsoid detections
 * (xyz + label) against ONE resident map of n_map landmarks: K-NN gate to each query's robot position
 * (ellipsoidMapManager.cpp:40-80), then label-gated nearest neighbour (sloam.cpp:158-203).  Inputs are DEVICE pointers (already
 * resident in HBM); the first-seen cloud is SoA (x[], y[], z[]: three coalesced float32 streams).  Runs on `stream` (a hipStream_t
 * cast to void*, NULL = default stream).  out_map_idx: n_query * n_obs map indices or -1.  The map may be of any size;
 * SLIDE_ERR_CAPACITY when min(K, n_map) exceeds the on-chip sort buffer (16384). */
int slide_assoc_sweep_batch_device(const float* d_cloud_x, const float* d_cloud_y, const float* d_cloud_z, const double* d_model_xyz,
                                   const int32_t* d_label, int n_map, const double* d_query_pos, const double* d_obs_xyz,
                                   const int32_t* d_obs_label, int n_query, int n_obs, int K, double thresh, int32_t* d_out_map_idx,
                                   void* stream);

/* The same sweep on caller-provided HOST buffers (cloud_xyz: AoS float32 as for slide_submap_knn): inputs are uploaded once, then
 * `repeats` launches run on the resident inputs; *ms_out (may be NULL) = their device time measured with HIP events on the launch
 * stream (uploads and the result copy excluded). */
int slide_assoc_sweep_batch(const float* cloud_xyz, const double* model_xyz, const int32_t* label, int n_map, const double* query_pos,
                            const double* obs_xyz, const int32_t* obs_label, int n_query, int n_obs, int K, double thresh,
                            int32_t* out_map_idx, int repeats, double* ms_out);

/* ------------------------------------------------------------------------------------------------
 * S2 + runSLOAMNode — the per-key-frame update (src/core/sloamNode.cpp:762-1036 without ROS):
 * submap gate -> projectModels -> match -> updateMap -> addSLOAMObservation -> solve ->
 * updateFactorGraphMap -> getCurrPose.
 * ---------------------------------------------------------------------------------------------- */
/* Body-frame detections of one key frame: field order/precision of sloam_msgs/SemanticMeasSyncOdom
 * (ROSCylinder/ROSCube/ROSEllipsoid; float32 fields already widened to double by the caller). */
typedef struct slide_detections_t {
  int n_cyl;
  const double* cyl_root;    /* 3 * n_cyl */
  const double* cyl_ray;     /* 3 * n_cyl */
  const double* cyl_radius;  /* n_cyl */
  const int32_t* cyl_label;
  int n_cube;
  const double* cube_pose7;  /* 7 * n_cube */
  const double* cube_scale;  /* 3 * n_cube */
  const int32_t* cube_label;
  int n_ell;
  const double* ell_pose7;
  const double* ell_scale;
  const int32_t* ell_label;
} slide_detections_t;

typedef struct slide_frame_result_t {
  double out_pose7[7];   /* outPose (sloamNode.cpp:1028) */
  int32_t* cyl_match;    /* caller buffers (may be NULL): submap index or -1, sloam.cpp:224-226 */
  int32_t* cube_match;
  int32_t* ell_match;
  int32_t* cyl_id;       /* global landmark ids used in the graph (L / C / U index) */
  int32_t* cube_id;
  int32_t* ell_id;
  int optimized;         /* addSLOAMObservation's return value */
  double ms_association; /* the two timers the reference keeps (sloamNode.cpp:845-849, 888-897), host wall clock */
  double ms_graph;
} slide_frame_result_t;

#define SLIDE_FRAME_HOST 0           /* host robot's own key frame: pose = prev * rel; add + solve + map refresh */
#define SLIDE_FRAME_HOST_DEFERRED 1  /* same, map refresh deferred to slide_backend_end_frame (foreign packets pending) */
#define SLIDE_FRAME_FOREIGN 2        /* another robot's packet (sloamNode.cpp:938-999): prev7 is the key pose ALREADY in the host frame; no solve */

slide_backend_t* slide_backend_create(const slide_params_t* p);
void slide_backend_destroy(slide_backend_t* b);
int slide_backend_process_frame(slide_backend_t* b, int mode, int robot, const double rel7[7], const double prev7[7],
                                const slide_detections_t* det, slide_frame_result_t* res);
int slide_backend_ingest_solve(slide_backend_t* b);                        /* sloamNode.cpp:1000 */
int slide_backend_end_frame(slide_backend_t* b, int robot, double out7[7]); /* sloamNode.cpp:1010-1014 */
slide_graph_t* slide_backend_graph(slide_backend_t* b);                    /* borrowed; owned by the backend */
/* [cyl_counter_, cube_counter_, point_landmark_counter_, #factors] + pose_counter_robot_ (graphWrapper.h:128-134) */
int slide_backend_counts(slide_backend_t* b, uint64_t out4[4], uint64_t* pose_counters, int n_robots);
/* map model read-back (getRawMap): cylinder -> 7 doubles; cube / ellipsoid -> xyz + scale (6 doubles) */
int slide_backend_map_model(slide_backend_t* b, int cls, int idx, double* out, int* hits, int* label);

/* ------------------------------------------------------------------------------------------------
 * S4 — inter-robot map-to-map association
 * ---------------------------------------------------------------------------------------------- */
/* SlideMatch parameters: PlaceRecognition::ParamInit (src/core/place_recognition.cpp:24-78). */
typedef struct slide_place_params_t {
  double dilation_factor;            /* 1.2 */
  double search_xy_step_size;        /* 0.5 */
  double match_yaw_half_range;       /* rad (180 deg) */
  double search_yaw_step_size;       /* rad (2 deg) */
  double match_threshold_position;   /* 0.5 */
  double match_threshold_dimension;  /* 1.0 */
  int disable_yaw_search;
  int ignore_dimension;
  int min_num_inliers;               /* 5 */
  int use_nonlinear_least_squares;   /* use_lsq, true */
  int min_num_map_objects_to_start;  /* 1 */
  int max_rings;                     /* -1 = all; replaces the wall-clock compute_budget_sec */
} slide_place_params_t;
void slide_place_default_params(slide_place_params_t* p);
/* PlaceRecognition::MatchMaps place_recognition.cpp:98-387 on already-centred maps.
 * ref7 / qry7: rows [label, x, y, z, d1, d2, d3].  best_xyyaw: winning (x, y, yaw); pair_*: matched pairs
 * (caller buffers of nq entries).  Returns the best inlier count (>= 0) or a negative SLIDE_ERR_*. */
int slide_match_maps(const double* ref7, int nr, const double* qry7, int nq, const slide_place_params_t* p,
                     double best_xyyaw[3], int32_t* pair_ref_idx, int32_t* pair_qry_idx, int64_t* n_candidates);
/* PlaceRecognition::findInterLoopClosure :498-538 (centring, sweep, inlier gate, Kabsch refinement).
 * Returns 1 found / 0 not found / negative error.  tf16: 4x4 row-major query->reference. */
int slide_find_inter_loop_closure(const double* ref7, int nr, const double* qry7, int nq, const slide_place_params_t* p,
                                  double tf16[16], int* inliers, double xyzyaw[4]);

/* PlaceRecognition::findIntraLoopClosure :389-496 (same-robot loop closure: the object detections around the query key pose against
 * the submap around an older candidate key pose).  meas7: detections in the query pose's LOCAL frame, submap7: map objects in the
 * map frame (rows [label, x, y, z, d1, d2, d3]).  The detections go into the map frame with the (drifted) query pose, then
 * findTransformation runs with inter_loop_closure == false (:801-816): no centring, the search window is the three intra half
 * ranges (match_{x,y,yaw}_half_range_intra, defaults 5 m / 5 m / 10 deg :53-63; yaw in radians here).  tf16 (row-major 4x4) =
 * candidate^-1 * query * [Rz(yaw) | (x, y, 0)] = tfFromQuery2Candidate.  Returns 1 found / 0 not found (fewer than 4 detections,
 * an empty input, too few inliers) / negative error. */
int slide_find_intra_loop_closure(const double* meas7, int nm, const double* submap7, int ns, const double query_pose7[7],
                                  const double candidate_pose7[7], const slide_place_params_t* p, double x_half_range_intra,
                                  double y_half_range_intra, double yaw_half_range_intra, double tf16[16], int* inliers,
                                  double xyzyaw[4]);

/* CLIPPER pairwise-consistency affinity (clipper_semantic_object/src/clipper.cpp:21-65 with the
 * EuclideanDistance invariant src/invariants/euclidean_distance.cpp:13-31).  D1: n1 points of `dim`
 * doubles (point-major), A: m x 2 association list.  M_out: m x m row-major, upper triangle filled. */
int slide_clipper_affinity(const double* D1, int n1, const double* D2, int n2, int dim, const int32_t* A, int m,
                           double sigma, double epsilon, double mindist, double affinityeps, double* M_out);

/* CLIPPER solver parameters: clipper.h:27-60 (solver) and invariants/euclidean_distance.h:24-29 (sigma, epsilon, mindist). */
typedef struct {
  double tol_u, tol_F;       /* 1e-8, 1e-9 */
  int maxiniters, maxoliters; /* 200, 1000 */
  double beta;               /* 0.25 */
  int maxlsiters;            /* 99 */
  double eps;                /* 1e-9 */
  double affinityeps;        /* 1e-4 */
  int rescale_u0;            /* 1 */
  double sigma, epsilon, mindist; /* 0.01, 0.06, 0 */
} slide_clipper_params_t;
void slide_clipper_default_params(slide_clipper_params_t* p);
/* CLIPPER::findDenseClique clipper.cpp:172-323 with DSD_HEU rounding (the only mode sloam uses).  M_upper: n x n row-major,
 * upper triangle filled (slide_clipper_affinity's output).  u0: n start weights; NULL draws them from a fixed-seed
 * generator (the reference uses a std::random_device-seeded mt19937, utils.cpp:22-29, i.e. is not reproducible).
 * nodes_out: caller buffer of n entries (selected associations, largest weight first); u_out (n) / score may be NULL.
 * The whole solve runs on the device: M as CSR (like the reference's sparse M / C, clipper.cpp:55-64), every product, reduction,
 * projection, line-search and stopping decision in one persistent workgroup — no host round trip inside the iteration. */
int slide_clipper_dense_clique(const double* M_upper, int n, const double* u0, const slide_clipper_params_t* p,
                               int32_t* nodes_out, int* n_nodes, double* u_out, double* score);
/* One LARGE problem (n >= 1024 associations; SURVEY A15 speaks of m ~ 1e4) runs on several co-resident workgroups — the rows of the
 * sparse product over the waves of up to 128 workgroups (cooperative launch), one grid barrier per gradient evaluation, everything
 * else repeated per workgroup so that the iterates equal the one-workgroup solve's bit for bit.  SLIDE_CLIPPER_WGS=<k> in the
 * environment forces the workgroup count (1 = one workgroup).  slide_clipper_last_solve_info: how the last
 * slide_clipper_dense_clique call of this process ran (workgroups, gradient evaluations); either pointer may be NULL. */
void slide_clipper_last_solve_info(int* n_workgroups, double* grad_evals);
/* Measurement aid of bench.py's SlideMatch / SlideGraph / CLIPPER legs (pair-tests/s of place_recognition.cpp:98-387, triangle
 * pairs/s of semantic_clipper.cpp:49-118, nnz * 12 B per product of clipper.cpp:172-323): what the LAST stand-alone call of this process
 * spent in its kernels — HIP events on the launch stream right around the launches; uploads, host prefix sums and read-backs excluded —
 * and the work those kernels were given. */
enum {
  SLIDE_MS_PLACE_SWEEP = 0,       /* ms: k_place_sweep + k_place_argmax of the last slide_match_maps / loop-closure search */
  SLIDE_MS_TRI_MATCH = 1,         /* ms: k_tri_prepare x 2 + k_tri_match count and emit passes */
  SLIDE_MS_CLQ_CSR = 2,           /* ms: k_clq_csr count + fill (CSR of the affinity matrix from its dense upper triangle) */
  SLIDE_MS_CLQ_SOLVE = 3,         /* ms: the projected-gradient solve (k_clq_solve or k_clq_solve_coop) */
  SLIDE_MS_AFFINITY = 4,          /* ms: k_clipper_affinity */
  SLIDE_MS_CLQ_NNZ = 5,           /* count: non-zeros of the last solve's CSR */
  SLIDE_MS_PLACE_PAIR_TESTS = 6,  /* count: candidates x query objects x reference objects of the last sweep (a first hit ends a query
                                     object's scan early: upper bound of the pair tests executed) */
  SLIDE_MS_TRI_PAIRS = 7,         /* count: model triangles x data triangles of the last triangle match */
  SLIDE_MS_PLACE_DIST_TESTS = 8,  /* count: distance tests the bucketed sweep was given (candidates x sum over the query objects of the
                                     reference objects of their label): what is left of the pair tests once the label test is a table */
  SLIDE_MS_COUNT = 9
};
int slide_last_device_ms(int what, double* out);
/* The same for several independent problems in ONE launch, a persistent workgroup per problem — the robot pairs of a multi-robot job
 * (semantic_clipper.cpp:227-235 once per pair; 28 pairs at eight robots).  Job j: M_upper[j] (n[j] x n[j]), u0[j] or NULL,
 * nodes_out[j] (>= n[j] ints), u_out[j] (n[j] doubles or NULL); n_nodes[j], score[j].  Results equal n_jobs single calls. */
int slide_clipper_dense_clique_batch(int n_jobs, const double* const* M_upper, const int* n, const double* const* u0, const slide_clipper_params_t* p,
                                     int32_t* const* nodes_out, int* n_nodes, double* const* u_out, double* score);
/* semantic_clipper::match_triangles / compute_triangle_diff semantic_clipper.cpp:49-118.  Triangles: 3 x (x, y) doubles each
 * (the Delaunay triangulation, observation.cpp:13-88, is the caller's).  pts_out: per matched pair three rows
 * [model x, model y, data x, data y] in ascending vertex-to-centroid distance; pairs in the reference's loop order
 * (model-major).  n_pairs returns the total; at most cap_pairs are written. */
int slide_match_triangles(const double* tri_model, int ntm, const double* tri_data, int ntd, double threshold,
                          double* pts_out, double* diffs_out, int cap_pairs, int* n_pairs);
/* semantic_clipper::estimate_tf :122-138 (2-D Kabsch a -> b; tf3 row-major 3x3). */
int slide_estimate_tf2d(const double* a_xy, const double* b_xy, int n, double tf3[9]);
/* semantic_clipper::run_semantic_clipper :140-274 from the triangle lists on: triangle matching, identity association
 * list, affinity, dense clique, min_num_pairs gate, estimate_tf, yaw + xy in a 4x4 row-major tf16 (query -> reference;
 * the caller inverts as place_recognition.cpp:621-624 does).  u0: start weights for the 3 * n_pairs putative
 * associations or NULL (fixed-seed generator).  counts: {putative associations, inliers}.  *found = 1 / 0. */
int slide_semantic_clipper(const double* tri_model, int ntm, const double* tri_data, int ntd, const slide_clipper_params_t* p,
                           int min_num_pairs, double matching_threshold, const double* u0, int n_u0, double tf16[16],
                           int counts[2], int32_t* inliers_out, int cap_inliers, int* found);

/* Input::PickNextMeasurementToAdd backend/sloam/src/core/input.cpp:26-108 (chronological measurement picker + key-frame
 * distance gate; pinned by src/test/input_test.cpp).  Queues are flat arrays with the front at index 0; stamps are (sec, nsec).
 * out4 = {meas_to_add (0 none, 1 odometry, 2 observation, 3 relative measurement), entries the reference pops from the front
 * of the odometry / observation / relative-measurement queue}.  Host bookkeeping, no kernel. */
int slide_pick_next_measurement(const int64_t* odom_sec, const int64_t* odom_nsec, const double* odom_pose7, int n_odom,
                                const int64_t* obs_sec, const int64_t* obs_nsec, int n_obs, const int64_t* rel_sec,
                                const int64_t* rel_nsec, int n_rel, int64_t latest_sec, int64_t latest_nsec,
                                const double latest_pose7[7], double current_time, double msg_delay_tolerance,
                                float min_odom_distance, int out4[4]);
/* CylinderMapManager::InLoopClosureRegion cylinderMapManager.cpp:115-160: is any key pose older than
 * at_least_num_of_poses_old within (max_dist_xy, max_dist_z) of the pose?  cloud: float32 xyz of the key poses
 * (robotPoseCloud_).  *inside = 0 / 1.  Host bookkeeping (a few thousand points per call). */
int slide_in_loop_closure_region(const float* cloud_xyz, int n, const double pose_xyz[3], double max_dist_xy, double max_dist_z,
                                 uint64_t at_least_num_of_poses_old, int* inside);

/* CylinderMapManager::getLoopCandidateIdx cylinderMapManager.cpp:160-184: the first key pose, in FLANN's nearest-first order, within
 * max_dist (float32 squared distances, strict '<') of key pose pose_idx that is not pose_idx itself and has
 * pose_idx - idx > at_least_num_of_poses_old in size_t arithmetic (so an index ABOVE pose_idx wraps around and qualifies, as in the
 * reference).  Equidistant neighbours are ordered by index (FLANN leaves that order unspecified).  Fewer than 50 key poses: not found.
 * Host bookkeeping. */
int slide_loop_candidate_idx(const float* cloud_xyz, int n, double max_dist, uint64_t pose_idx, uint64_t at_least_num_of_poses_old,
                             uint64_t* candidate_idx, int* found);

/* 2-D Delaunay triangulation (replaces the qhull call of DelaunayTriangulation::Observation, triangulation/observation.cpp:13-88,
 * options "Qt Qbb Qc Qz Q12 d").  Host code (sweep-hull + Lawson flips, long double predicates).  tri_out: vertex-index triples,
 * ascending inside a triangle, triangles in lexicographic order — the same triangle SET as qhull for points in general position;
 * the list order (qhull: facet order) only permutes SlideGraph's putative association list.  At most cap triangles are written. */
int slide_delaunay_2d(const double* xy, int n, int32_t* tri_out, int cap, int* n_tri);
/* semantic_clipper::run_semantic_clipper semantic_clipper.cpp:140-274 from the two object maps on (rows [label, x, y, z, d1, d2,
 * d3]; only x, y are used there): Delaunay of both, then slide_semantic_clipper.  sigma / epsilon: the EuclideanDistance
 * invariant's parameters as passed by the reference's caller. */
int slide_run_semantic_clipper(const double* ref7, int nr, const double* qry7, int nq, double sigma, double epsilon,
                               int min_num_pairs, double matching_threshold, const double* u0, int n_u0, double tf16[16],
                               int counts[2], int* found);

/* sloam::FindRelativeMeasurementMatch / GetIndexClosestPoseMstPair (src/core/sloam.cpp:321-440).
 * Stamps are (sec, nsec) pairs.  Host-side logic (tiny, sequential): no kernel. */
int slide_closest_stamp(const int64_t* sec, const int64_t* nsec, int n, int64_t qsec, int64_t qnsec, int* idx, double* diff);
/* sloam::FindRelativeMeasurementMatch sloam.cpp:321-412.  Packets of all robots concatenated (pk_sec / pk_nsec) with
 * offsets[n_robots + 1]; pose_counter per robot; pending measurements (stamp, observed robot, onlyUseOdom flag, caller tag) are
 * compacted IN PLACE exactly as feasible_relative_meas_for_factors is (matched and stale ones erased).  match_out: 4 ints per
 * match {tag, position in the pending list at match time, host pose index, other pose index}.  *n_matches >= 0; returns
 * SLIDE_ERR_INVALID where the reference throws std::runtime_error (robotIndex == host, onlyUseOdom measurement). */
int slide_find_relative_meas_match(int n_robots, const int64_t* pk_sec, const int64_t* pk_nsec, const int32_t* offsets,
                                   const uint64_t* pose_counter, int host, int* n_pending_io, int64_t* m_sec, int64_t* m_nsec,
                                   int32_t* m_robot, int32_t* m_only_odom, int32_t* m_tag, int32_t* match_out, int* n_matches);

#ifdef __cplusplus
}
#endif
#endif /* SLIDE_GPU_H_ */
