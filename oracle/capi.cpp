// ORACLE — TEST INFRASTRUCTURE ONLY (see lie.hpp header).  Flat C entry points so that
// tests/ (ctypes) and bench.py's cpu_baseline leg can drive the CPU restatement.
#include <cstring>
#include <vector>

#include "backend.hpp"
#include "clipper.hpp"
#include "slidegraph.hpp"
#include "input.hpp"
#include "place.hpp"
#include "relmeas.hpp"

using namespace orc;

static Pose pose_from12(const double* p) {
  Pose T;
  std::memcpy(T.R, p, 72);
  std::memcpy(T.t, p + 9, 24);
  return T;
}
static void pose_to12(const Pose& T, double* p) {
  std::memcpy(p, T.R, 72);
  std::memcpy(p + 9, T.t, 24);
}

extern "C" {

// ---- Lie group primitives (pose12 = R row-major (9) + t (3)) --------------------------------
void orc_pose_expmap(const double* xi, double* out12) { pose_to12(pose_expmap(xi), out12); }
void orc_pose_logmap(const double* p12, double* xi) { pose_logmap(pose_from12(p12), xi); }
void orc_pose_retract(const double* p12, const double* xi, int chart, double* out12) {
  pose_to12(pose_retract(pose_from12(p12), xi, chart), out12);
}
void orc_pose_local(const double* a12, const double* b12, int chart, double* xi) {
  pose_local(pose_from12(a12), pose_from12(b12), xi, chart);
}
void orc_pose_compose(const double* a12, const double* b12, double* out12) {
  pose_to12(pose_compose(pose_from12(a12), pose_from12(b12)), out12);
}
void orc_pose_inverse(const double* a12, double* out12) { pose_to12(pose_inverse(pose_from12(a12)), out12); }
void orc_pose_adjoint(const double* a12, double* out36) { pose_adjoint(pose_from12(a12), out36); }
void orc_pose7_to12(const double* p7, double* out12) { pose_to12(pose_from7(p7), out12); }
void orc_pose12_to7(const double* p12, double* out7) { pose_to7(pose_from12(p12), out7); }
void orc_so3_cayley(const double* w, double* R) { so3_cayley(w, R); }
void orc_so3_cayley_local(const double* R, double* w) { so3_cayley_local(R, w); }
void orc_so3_expmap(const double* w, double* R) { so3_expmap(w, R); }
void orc_so3_logmap(const double* R, double* w) { so3_logmap(R, w); }
void orc_unit3_basis(const double* n, double* B6) { unit3_basis(n, B6); }
void orc_unit3_local(const double* p, const double* q, double* out2) { unit3_local(p, q, out2); }

// ---- CubeMeasurement / CylinderMeasurement manifold ops (cubeFactor.h, cylinderFactor.h) -----
// cube15 = R(9) t(3) scale(3)
void orc_cube_Retract_static(const double* v9, double* out15) {  // CubeMeasurement::Retract cubeFactor.h:131-144
  pose_to12(pose_expmap(v9), out15);
  for (int i = 0; i < 3; ++i) out15[12 + i] = v9[6 + i];
}
void orc_cube_LocalCoordinates_static(const double* q15, double* v9) {  // cubeFactor.h:146-159
  pose_logmap(pose_from12(q15), v9);
  for (int i = 0; i < 3; ++i) v9[6 + i] = q15[12 + i];
}
void orc_cube_localCoordinates(const double* m15, const double* q15, double* v9) {  // cubeFactor.h:46-87
  Pose e = pose_compose(pose_inverse(pose_from12(q15)), pose_from12(m15));
  pose_logmap(e, v9);
  for (int i = 0; i < 3; ++i) v9[6 + i] = m15[12 + i] - q15[12 + i];
}
void orc_cube_retract(const double* m15, const double* v9, int chart, double* out15) {  // cubeFactor.h:95-114
  Var in{}, out{};
  in.type = V_CUBE;
  std::memcpy(in.val, m15, 15 * sizeof(double));
  var_retract(in, v9, chart, out);
  std::memcpy(out15, out.val, 15 * sizeof(double));
}

// ---- single-factor linearisation -------------------------------------------------------------
// type: FType; x0/x1: variable value layouts (graph.hpp); returns m; J row-major m x d.
int orc_linearize(int ftype, const double* x0, int x1type, const double* x1, const double* z, const double* sigma,
                  int chart, double numdiff_delta, double* r, double* J0, double* J1, int whiten) {
  std::vector<Var> vars(2);
  vars[0].type = V_POSE;
  std::memcpy(vars[0].val, x0, 12 * sizeof(double));
  vars[1].type = x1type;
  if (x1) std::memcpy(vars[1].val, x1, sizeof(double) * (x1type == V_POSE ? 12 : x1type == V_POINT ? 3 : x1type == V_CUBE ? 15 : 7));
  Factor f{};
  f.type = ftype;
  f.v0 = 0;
  f.v1 = 1;
  const int nz = (ftype == F_BR) ? 4 : (ftype == F_CUBE) ? 15 : (ftype == F_CYL) ? 7 : 12;
  std::memcpy(f.z, z, nz * sizeof(double));
  const int m = fac_dim(ftype);
  for (int i = 0; i < m; ++i) f.sigma[i] = whiten ? sigma[i] : 1.0;
  GraphParams P;
  P.pose_chart = chart;
  P.numdiff_delta = numdiff_delta;
  LinFactor L;
  linearize_factor(f, vars, P, L);
  std::memcpy(r, L.r, m * sizeof(double));
  std::memcpy(J0, L.J0, sizeof(double) * m * L.d0);
  if (L.d1) std::memcpy(J1, L.J1, sizeof(double) * m * L.d1);
  return m;
}

// ---- graph (SemanticFactorGraph seam) --------------------------------------------------------
struct OrcParams {
  int pose_chart;
  double relin_threshold;
  double prior_sigma[6], odom_sigma[6], cube_sigma[9], relmeas_sigma[6];
  double cyl_sigma, bearing_sigma;
  double cyl_thresh, cube_thresh, ell_thresh;
  int num_threads;
};
static void apply_params(const OrcParams* p, GraphParams& G) {
  if (!p) return;
  G.pose_chart = p->pose_chart;
  G.relin_threshold = p->relin_threshold;
  std::memcpy(G.prior_sigma, p->prior_sigma, 48);
  std::memcpy(G.odom_sigma, p->odom_sigma, 48);
  std::memcpy(G.cube_sigma, p->cube_sigma, 72);
  std::memcpy(G.relmeas_sigma, p->relmeas_sigma, 48);
  G.cyl_sigma = p->cyl_sigma;
  G.bearing_sigma = p->bearing_sigma;
  G.num_threads = p->num_threads > 0 ? p->num_threads : 1;
}

void* orc_graph_create(const OrcParams* p) {
  Graph* g = new Graph();
  apply_params(p, g->P);
  return g;
}
void orc_graph_destroy(void* h) { delete (Graph*)h; }
void orc_graph_set_prior(void* h, int robot, const double* pose7) { ((Graph*)h)->setPriors(pose_from7(pose7), robot); }
void orc_graph_add_keypose_between(void* h, int robot, uint64_t from, uint64_t to, const double* rel7,
                                   const double* est7) {
  ((Graph*)h)->addKeyPoseAndBetween(from, to, pose_from7(rel7), pose_from7(est7), robot);
}
void orc_graph_add_loop_closure(void* h, const double* rel7, uint64_t i1, int r1, uint64_t i2, int r2) {
  ((Graph*)h)->addLoopClosureFactor(pose_from7(rel7), i1, r1, i2, r2);
}
void orc_graph_add_relative_meas(void* h, const double* rel7, uint64_t i1, int r1, uint64_t i2, int r2) {
  ((Graph*)h)->addRelativeMeasFactor(pose_from7(rel7), i1, r1, i2, r2);
}
void orc_graph_add_point_landmark(void* h, uint64_t idx, const double* xyz) { ((Graph*)h)->addPointLandmarkKey(idx, xyz); }
void orc_graph_add_range_bearing(void* h, int robot, uint64_t poseIdx, uint64_t lmIdx, const double* bearing,
                                 double range) {
  ((Graph*)h)->addRangeBearingFactor(poseIdx, lmIdx, bearing, range, robot);
}
void orc_graph_add_cube(void* h, int robot, uint64_t poseIdx, uint64_t cubeIdx, const double* pose7,
                        const double* cube7, const double* scale, int exists) {
  ((Graph*)h)->addCubeFactor(poseIdx, cubeIdx, pose_from7(pose7), pose_from7(cube7), scale, exists != 0, robot);
}
void orc_graph_add_cylinder(void* h, int robot, uint64_t poseIdx, uint64_t cylIdx, const double* pose7,
                            const double* root, const double* ray, double radius, int exists) {
  ((Graph*)h)->addCylinderFactor(poseIdx, cylIdx, pose_from7(pose7), root, ray, radius, exists != 0, robot);
}
int orc_graph_solve(void* h) { return ((Graph*)h)->solve(); }
int orc_graph_get_pose(void* h, int robot, uint64_t idx, double* out7) {
  Pose T;
  const bool ok = ((Graph*)h)->getPose(idx, robot, T);
  pose_to7(T, out7);
  return ok ? 0 : 1;
}
int orc_graph_get_pose12(void* h, int robot, uint64_t idx, double* out12) {
  Pose T;
  const bool ok = ((Graph*)h)->getPose(idx, robot, T);
  pose_to12(T, out12);
  return ok ? 0 : 1;
}
// cls: 0 cylinder (7: root ray radius), 1 cube (15: R t scale), 2 point (3)
int orc_graph_get_landmark(void* h, int cls, uint64_t idx, double* out) {
  const char c = cls == 0 ? 'l' : cls == 1 ? 'c' : 'u';
  const Var* v = ((Graph*)h)->getLandmark(c, idx);
  const int n = cls == 0 ? 7 : cls == 1 ? 15 : 3;
  if (!v) { for (int i = 0; i < n; ++i) out[i] = 0.0; return 1; }
  std::memcpy(out, v->val, n * sizeof(double));
  return 0;
}
void orc_graph_stats(void* h, double* out8) {
  const SolveStats& s = ((Graph*)h)->stats;
  out8[0] = s.n_pose; out8[1] = s.n_lm; out8[2] = s.n_factors; out8[3] = s.n_relin;
  out8[4] = s.t_linearize; out8[5] = s.t_schur; out8[6] = s.t_chol; out8[7] = s.t_total;
}
int orc_graph_set_shared(void* h, const int* cls, const int64_t* idx, const int* owner, int n) {
  return ((Graph*)h)->set_shared(cls, idx, owner, n);
}
int orc_graph_dist_phase(void* h, int phase, double* buf) { return ((Graph*)h)->dist_phase(phase, buf); }
// joint solve of the sharded pass: PCG iterations (upper bound) and relative tolerance on sqrt(r^T M^-1 r); stats: iterations that
// did work in the last solve, first and last gamma, state (0 running, 1 converged, 2 breakdown)
void orc_graph_set_pcg(void* h, int iters, double tol) { ((Graph*)h)->pcg_iters = iters < 0 ? 0 : iters; ((Graph*)h)->pcg_tol = tol; }
void orc_graph_pcg_stats(void* h, double* out4) {
  const Graph* g = (Graph*)h;
  out4[0] = g->D.pcg_its; out4[1] = g->D.gamma0; out4[2] = g->D.gamma_last; out4[3] = g->D.pcg_done;
}
int orc_graph_set_separator(void* h, const int* off, int n) { return ((Graph*)h)->set_separator(off, n); }
int orc_graph_set_ghost_ids(void* h, const int* ids, int n, int n_total) { return ((Graph*)h)->set_ghost_ids(ids, n, n_total); }
void orc_graph_keep_factor(void* h, int on) { ((Graph*)h)->keep_factor = on != 0; }
int orc_graph_pose_covariance(void* h, int robot, uint64_t idx, double* cov36) {
  return ((Graph*)h)->pose_covariance(Graph::pose_key(robot, idx), cov36);
}
int orc_graph_set_ghosts(void* h, const int* own_robot, const int64_t* own_idx, int n) { return ((Graph*)h)->set_ghosts(own_robot, own_idx, n); }
void orc_graph_add_relative_meas_ghost(void* h, const double* rel7, uint64_t idx, int robot, int slot, int local_first) {
  ((Graph*)h)->addRelativeMeasGhost(pose_from7(rel7), idx, robot, slot, local_first != 0);
}
// landmark table of a backend (cls 0 cyl root / 1 cube / 2 point): xyz of the current estimate + label
int orc_backend_landmark_table(void* h, int cls, double* xyz, int* label, int cap) {
  Backend* b = (Backend*)h;
  const uint64_t counters[3] = {b->cyl_counter, b->cube_counter, b->point_counter};
  const int n = (int)std::min<uint64_t>(counters[cls], (uint64_t)cap);
  for (int i = 0; i < n; ++i) {
    const Var* v = b->graph.getLandmark(cls == 0 ? 'l' : (cls == 1 ? 'c' : 'u'), (uint64_t)i);
    if (!v) return -1;
    const double* pos = cls == 1 ? v->val + 9 : v->val;
    xyz[3 * i] = pos[0]; xyz[3 * i + 1] = pos[1]; xyz[3 * i + 2] = pos[2];
    label[i] = cls == 0 ? b->cylMap.models[i].label : (cls == 1 ? b->cubeMap.models[i].label : b->ellMap.models[i].label);
  }
  return n;
}
// force a full relinearisation on the next solve (batch Gauss-Newton mode = threshold 0)
void orc_graph_set_relin_threshold(void* h, double thr) { ((Graph*)h)->P.relin_threshold = thr; }
// [GTSAM] iSAM2's wildfire threshold on the back-substitution (Graph::wildfire_bound); 0 = off (the default); out3: blocks kept in
// total, in the last solve, the last solve's first dirty block column
void orc_graph_set_wildfire(void* h, double thr) { ((Graph*)h)->P.wildfire_threshold = thr > 0.0 ? thr : 0.0; }
// the rule on plain arrays (unit test of Graph::wildfire_bound): dp (n doubles, the exact solution, overwritten), prev (the last solve's,
// T * 64 doubles), prof_last[c] = last block row in the profile of block column c, c_d = the first dirty block column; returns the blocks kept
int orc_wildfire_rule(double* dp, const double* prev, int n, int T, const int* prof_last, int c_d, double thr) {
  Graph g;
  g.P.wildfire_threshold = thr;
  g.wf_prev.assign(prev, prev + (size_t)T * 64);
  g.wf_Tprev = T;
  g.wf_dirty_min_pose = (c_d * 64 + 5) / 6;      // the lowest pose whose first coordinate lies in block column c_d
  CholProfile pf;
  pf.rend.resize(T);
  for (int c = 0; c < T; ++c) pf.rend[c] = std::min(n, (prof_last[c] + 1) * 64);
  std::vector<double> v(dp, dp + n);
  g.wildfire_bound(v, n, pf);
  std::memcpy(dp, v.data(), sizeof(double) * n);
  return g.wf_kept_last;
}
void orc_graph_wildfire_stats(void* h, long long* out3) {
  const Graph* g = (const Graph*)h;
  out3[0] = g->wf_kept_total; out3[1] = g->wf_kept_last; out3[2] = g->wf_last_cd;
}

// ---- association primitives ------------------------------------------------------------------
// Self-check of the profile-restricted Cholesky (graph.hpp chol_profile): solves A x = b for a row-major SPD matrix (lower triangle
// used) once with the dense loops and once inside the profile; x_dense / x_prof: n doubles each.  Returns 0, or 1 + failing pivot.
int orc_chol_solve_both(const double* A, int n, const double* b, double* x_dense, double* x_prof, int* rows_in_profile) {
  std::vector<double> L1(A, A + (size_t)n * n), L2(A, A + (size_t)n * n);
  int rc = chol_lower(L1.data(), n, n, 1, nullptr);
  if (rc != 0) return rc;
  std::memcpy(x_dense, b, sizeof(double) * n);
  chol_solve_lower(L1.data(), n, n, x_dense, nullptr);
  const CholProfile prof = chol_profile(L2.data(), n, n);
  rc = chol_lower(L2.data(), n, n, 1, &prof);
  if (rc != 0) return rc;
  std::memcpy(x_prof, b, sizeof(double) * n);
  chol_solve_lower(L2.data(), n, n, x_prof, &prof);
  if (rows_in_profile) {
    long long t = 0;
    for (size_t bl = 0; bl < prof.rend.size(); ++bl) t += prof.rend[bl] - (long long)bl * 64;
    *rows_in_profile = (int)std::min<long long>(t, 2147483647LL);
  }
  return 0;
}
int orc_knn_f32(const float* cloud_xyz, int n, const double* query, int K, int* out_idx) {
  std::vector<float> c(cloud_xyz, cloud_xyz + 3 * (size_t)n);
  std::vector<int> o;
  knn_f32(c, query, K, o);
  for (size_t i = 0; i < o.size(); ++i) out_idx[i] = o[i];
  return (int)o.size();
}
static void fill_cyl(int n, const double* root, const double* ray, const double* radius, const int* label,
                     std::vector<CylObj>& v) {
  v.resize(n);
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) { v[i].root[k] = root[3 * i + k]; v[i].ray[k] = ray[3 * i + k]; }
    v[i].radius = radius[i];
    v[i].label = label[i];
  }
}
static void fill_box(int n, const double* pose7, const double* scale, const int* label, std::vector<BoxObj>& v) {
  v.resize(n);
  for (int i = 0; i < n; ++i) {
    v[i].pose = pose_from7(pose7 + 7 * i);
    for (int k = 0; k < 3; ++k) v[i].scale[k] = scale[3 * i + k];
    v[i].label = label[i];
  }
}
// cls 0: cylinders (a = root, b = ray, c = radius); world-frame objects on both sides
void orc_match_cylinders(int n_cur, const double* root, const double* ray, const double* radius, const int* label,
                         int n_map, const double* mroot, const double* mray, const double* mradius,
                         const int* mlabel, double thresh, int* out) {
  std::vector<CylObj> cur, map;
  fill_cyl(n_cur, root, ray, radius, label, cur);
  fill_cyl(n_map, mroot, mray, mradius, mlabel, map);
  std::vector<int> idx(n_cur, -1);
  match_cylinders(cur, map, thresh, idx);
  for (int i = 0; i < n_cur; ++i) out[i] = idx[i];
}
// cls 1 cube / 2 ellipsoid: positions only matter (xyz), labels for ellipsoids
void orc_match_boxes(int cls, int n_cur, const double* xyz, const int* label, int n_map, const double* mxyz,
                     const int* mlabel, double thresh, int* out) {
  std::vector<BoxObj> cur(n_cur), map(n_map);
  for (int i = 0; i < n_cur; ++i) { pose_identity(cur[i].pose); std::memcpy(cur[i].pose.t, xyz + 3 * i, 24); cur[i].label = label[i]; }
  for (int i = 0; i < n_map; ++i) { pose_identity(map[i].pose); std::memcpy(map[i].pose.t, mxyz + 3 * i, 24); map[i].label = mlabel[i]; }
  std::vector<int> idx(n_cur, -1);
  if (cls == 1) match_cubes(cur, map, thresh, idx);
  else match_ellipsoids(cur, map, thresh, idx);
  for (int i = 0; i < n_cur; ++i) out[i] = idx[i];
}

// ---- backend (runSLOAMNode seam) -------------------------------------------------------------
void* orc_backend_create(const OrcParams* p, int num_robots) {
  Backend* b = new Backend(num_robots);
  apply_params(p, b->graph.P);
  if (p) { b->mp.cyl_thresh = p->cyl_thresh; b->mp.cube_thresh = p->cube_thresh; b->mp.ell_thresh = p->ell_thresh; }
  return b;
}
void orc_backend_destroy(void* h) { delete (Backend*)h; }

static void emit(const FrameResult& r, double* out7, int* cm, int* bm, int* em, int* cid, int* bid, int* eid,
                 double* timers) {
  if (out7) pose_to7(r.out_pose, out7);
  for (size_t i = 0; i < r.cyl_match.size(); ++i) { if (cm) cm[i] = r.cyl_match[i]; if (cid) cid[i] = r.cyl_map_idx[i]; }
  for (size_t i = 0; i < r.cube_match.size(); ++i) { if (bm) bm[i] = r.cube_match[i]; if (bid) bid[i] = r.cube_map_idx[i]; }
  for (size_t i = 0; i < r.ell_match.size(); ++i) { if (em) em[i] = r.ell_match[i]; if (eid) eid[i] = r.ell_map_idx[i]; }
  if (timers) { timers[0] = r.t_assoc; timers[1] = r.t_graph; }
}

// mode 0: host frame (prev7 * rel7, solve, map refresh); 1: host frame with deferred map refresh;
// 2: foreign packet (prev7 is the pose ALREADY in the host frame; no solve)
int orc_backend_process_frame(void* h, int mode, int robot, const double* rel7, const double* prev7, int n_cyl,
                              const double* cyl_root, const double* cyl_ray, const double* cyl_radius,
                              const int* cyl_label, int n_cube, const double* cube_pose7, const double* cube_scale,
                              const int* cube_label, int n_ell, const double* ell_pose7, const double* ell_scale,
                              const int* ell_label, double* out_pose7, int* cyl_match, int* cube_match,
                              int* ell_match, int* cyl_id, int* cube_id, int* ell_id, double* timers2) {
  Backend* b = (Backend*)h;
  Detections d;
  fill_cyl(n_cyl, cyl_root, cyl_ray, cyl_radius, cyl_label, d.cyl);
  fill_box(n_cube, cube_pose7, cube_scale, cube_label, d.cube);
  fill_box(n_ell, ell_pose7, ell_scale, ell_label, d.ell);
  FrameResult r;
  if (mode == 2) r = b->ingest_packet(robot, pose_from7(rel7), pose_from7(prev7), d);
  else r = b->process_frame(robot, pose_from7(rel7), pose_from7(prev7), d, mode == 1);
  emit(r, out_pose7, cyl_match, cube_match, ell_match, cyl_id, cube_id, ell_id, timers2);
  return r.solve_status;
}
int orc_backend_ingest_solve(void* h) { return ((Backend*)h)->ingest_solve(); }
int orc_backend_end_frame(void* h, int robot, double* out7) {
  Pose T;
  const bool ok = ((Backend*)h)->end_frame(robot, T);
  pose_to7(T, out7);
  return ok ? 0 : 1;
}
void* orc_backend_graph(void* h) { return &((Backend*)h)->graph; }
void orc_backend_counts(void* h, uint64_t* out4, uint64_t* pose_counters, int n_robots) {
  Backend* b = (Backend*)h;
  out4[0] = b->cyl_counter; out4[1] = b->cube_counter; out4[2] = b->point_counter; out4[3] = b->graph.factors.size();
  for (int i = 0; i < n_robots && i < (int)b->pose_counter.size(); ++i) pose_counters[i] = b->pose_counter[i];
}
// map model read-back: cls 0 -> 7 doubles (root ray radius); 1/2 -> 6 doubles (xyz scale)
int orc_backend_map_model(void* h, int cls, int idx, double* out, int* hits, int* label) {
  Backend* b = (Backend*)h;
  if (cls == 0) {
    if (idx >= (int)b->cylMap.models.size()) return 1;
    const CylObj& c = b->cylMap.models[idx];
    for (int k = 0; k < 3; ++k) { out[k] = c.root[k]; out[3 + k] = c.ray[k]; }
    out[6] = c.radius; *hits = b->cylMap.hits[idx]; *label = c.label;
  } else {
    auto& M = cls == 1 ? b->cubeMap : b->ellMap;
    if (idx >= (int)M.models.size()) return 1;
    const BoxObj& c = M.models[idx];
    for (int k = 0; k < 3; ++k) { out[k] = c.pose.t[k]; out[3 + k] = c.scale[k]; }
    *hits = M.hits[idx]; *label = c.label;
  }
  return 0;
}

// ---- CLIPPER ---------------------------------------------------------------------------------
struct OrcClipperParams {
  double tol_u, tol_F;
  int maxiniters, maxoliters;
  double beta;
  int maxlsiters;
  double eps, affinityeps;
  int rescale_u0;
  double sigma, epsilon, mindist;
};
static ClipperParams cp_from(const OrcClipperParams* p) {
  ClipperParams P;
  if (!p) return P;
  P.tol_u = p->tol_u; P.tol_F = p->tol_F; P.maxiniters = p->maxiniters; P.maxoliters = p->maxoliters;
  P.beta = p->beta; P.maxlsiters = p->maxlsiters; P.eps = p->eps; P.affinityeps = p->affinityeps;
  P.rescale_u0 = p->rescale_u0 != 0; P.sigma = p->sigma; P.epsilon = p->epsilon; P.mindist = p->mindist;
  return P;
}
void orc_clipper_default_params(OrcClipperParams* p) {
  ClipperParams P;
  p->tol_u = P.tol_u; p->tol_F = P.tol_F; p->maxiniters = P.maxiniters; p->maxoliters = P.maxoliters;
  p->beta = P.beta; p->maxlsiters = P.maxlsiters; p->eps = P.eps; p->affinityeps = P.affinityeps;
  p->rescale_u0 = P.rescale_u0; p->sigma = P.sigma; p->epsilon = P.epsilon; p->mindist = P.mindist;
}
// A_io: in: m x 2 (or m = 0 -> all-to-all, caller passes room for n1*n2 rows); returns m; M: m x m upper
int orc_clipper_affinity(const double* D1, int n1, const double* D2, int n2, int dim, int* A_io, int m_in,
                         const OrcClipperParams* p, double* M_out) {
  std::vector<int> A(A_io, A_io + 2 * (size_t)m_in);
  std::vector<double> M;
  clipper_affinity(D1, n1, D2, n2, dim, A, cp_from(p), M);
  const int m = (int)(A.size() / 2);
  for (size_t i = 0; i < A.size(); ++i) A_io[i] = A[i];
  if (M_out) std::memcpy(M_out, M.data(), sizeof(double) * M.size());
  return m;
}
int orc_clipper_solve(const double* Mup, int n, const double* u0, const OrcClipperParams* p, int* nodes_out,
                      double* u_out, double* score_out) {
  std::vector<double> M(Mup, Mup + (size_t)n * n), u(u0, u0 + n);
  ClipperSolution s = clipper_dense_clique(M, n, u, cp_from(p));
  for (size_t i = 0; i < s.nodes.size(); ++i) nodes_out[i] = s.nodes[i];
  if (u_out) std::memcpy(u_out, s.u.data(), sizeof(double) * n);
  if (score_out) *score_out = s.score;
  return (int)s.nodes.size();
}

// ---- SlideGraph triangle matching (A14) -----------------------------------------------------------
// pts_out: rows [mx, my, dx, dy], three per matched pair; returns the number of matched pairs (writes at most cap_pairs)
int orc_match_triangles(const double* tm, int ntm, const double* td, int ntd, double thr, double* pts_out, double* diffs_out,
                        int cap_pairs) {
  std::vector<double> pts, diffs;
  const size_t np = match_triangles(tm, ntm, td, ntd, thr, pts, diffs);
  const size_t w = std::min<size_t>(np, (size_t)cap_pairs);
  if (pts_out) std::memcpy(pts_out, pts.data(), sizeof(double) * 12 * w);
  if (diffs_out) std::memcpy(diffs_out, diffs.data(), sizeof(double) * w);
  return (int)np;
}
void orc_estimate_tf2d(const double* a, const double* b, int n, double* tf3) { estimate_tf2d(a, b, n, tf3); }
// returns 1 when a transform was found; counts[0] = putative associations, counts[1] = inliers
int orc_semantic_clipper(const double* tm, int ntm, const double* td, int ntd, const OrcClipperParams* p, int min_num_pairs,
                         double matching_threshold, const double* u0, double* tf16, int* counts, int* inliers_out) {
  SemanticClipperOut o = semantic_clipper(tm, ntm, td, ntd, cp_from(p), min_num_pairs, matching_threshold, u0);
  std::memcpy(tf16, o.tf16, sizeof(o.tf16));
  counts[0] = o.n_putative;
  counts[1] = o.n_inliers;
  if (inliers_out) for (size_t i = 0; i < o.inliers.size(); ++i) inliers_out[i] = o.inliers[i];
  return o.ok ? 1 : 0;
}

// ---- key-frame gating (N3) -----------------------------------------------------------------------
void orc_pick_next_measurement(const int64_t* odom_sec, const int64_t* odom_nsec, const double* odom_pose7, int n_odom,
                               const int64_t* obs_sec, const int64_t* obs_nsec, int n_obs, const int64_t* rel_sec,
                               const int64_t* rel_nsec, int n_rel, int64_t latest_sec, int64_t latest_nsec,
                               const double* latest7, double current_time, double msg_delay_tolerance, float min_odom_distance,
                               int* out4) {
  std::vector<double> p12(12 * (size_t)std::max(n_odom, 1));
  for (int i = 0; i < n_odom; ++i) pose_to12(pose_from7(odom_pose7 + 7 * (size_t)i), p12.data() + 12 * (size_t)i);
  double l12[12];
  pose_to12(pose_from7(latest7), l12);
  const PickResult r = pick_next_measurement(odom_sec, odom_nsec, p12.data(), n_odom, obs_sec, obs_nsec, n_obs, rel_sec, rel_nsec, n_rel,
                                             latest_sec, latest_nsec, l12, current_time, msg_delay_tolerance, min_odom_distance);
  out4[0] = r.meas_to_add; out4[1] = r.pop_odom; out4[2] = r.pop_obs; out4[3] = r.pop_rel;
}
int orc_loop_candidate_idx(const float* cloud, int n, double max_dist, uint64_t pose_idx, uint64_t at_least, uint64_t* cand) {
  return loop_candidate_idx(cloud, n, max_dist, pose_idx, at_least, cand) ? 1 : 0;
}
int orc_in_loop_closure_region(const float* cloud, int n, const double* pose_t, double max_xy, double max_z, uint64_t at_least) {
  return in_loop_closure_region(cloud, n, pose_t, max_xy, max_z, at_least) ? 1 : 0;
}

// ---- relative-measurement matching (A16) -----------------------------------------------------
void orc_closest_stamp(const int64_t* sec, const int64_t* nsec, int n, int64_t qsec, int64_t qnsec, int* idx,
                       double* diff) {
  std::vector<Stamp> pk(n);
  for (int i = 0; i < n; ++i) pk[i] = {sec[i], nsec[i]};
  closest_stamp(pk, Stamp{qsec, qnsec}, *idx, *diff);
}
// packets: concatenated per robot with offsets[n_robots+1]; pending arrays are compacted in place.
// returns #matches, or -1 when the reference would throw std::runtime_error.
int orc_find_relmeas(int n_robots, const int64_t* pk_sec, const int64_t* pk_nsec, const int* offsets,
                     const uint64_t* pose_counter, int host, int* n_pending_io, int64_t* m_sec, int64_t* m_nsec,
                     int* m_robot, int* m_only_odom, int* m_tag, int* match_out /* 4 per match */) {
  std::vector<std::vector<Stamp>> packets(n_robots);
  for (int r = 0; r < n_robots; ++r)
    for (int i = offsets[r]; i < offsets[r + 1]; ++i) packets[r].push_back({pk_sec[i], pk_nsec[i]});
  std::vector<size_t> pc(pose_counter, pose_counter + n_robots);
  std::vector<RelMeas> pending(*n_pending_io);
  for (int i = 0; i < *n_pending_io; ++i) pending[i] = {Stamp{m_sec[i], m_nsec[i]}, m_robot[i], m_only_odom[i] != 0, m_tag[i]};
  std::vector<RelMeasMatch> matches;
  try {
    find_relmeas_matches(pending, pc, packets, host, matches);
  } catch (const std::runtime_error&) {
    return -1;
  }
  *n_pending_io = (int)pending.size();
  for (size_t i = 0; i < pending.size(); ++i) {
    m_sec[i] = pending[i].stamp.sec; m_nsec[i] = pending[i].stamp.nsec; m_robot[i] = pending[i].robotIndex;
    m_only_odom[i] = pending[i].onlyUseOdom; m_tag[i] = pending[i].tag;
  }
  for (size_t i = 0; i < matches.size(); ++i) {
    match_out[4 * i] = matches[i].tag; match_out[4 * i + 1] = matches[i].index;
    match_out[4 * i + 2] = matches[i].hostIdx; match_out[4 * i + 3] = matches[i].otherIdx;
  }
  return (int)matches.size();
}

// ---- SlideMatch (A13) ------------------------------------------------------------------------
int orc_match_maps(const double* ref7, int nr, const double* qry7, int nq, const OrcPlaceParams* p, double* best_xyyaw,
                   int* pair_ref_idx, int* pair_qry_idx) {
  PlaceParams P = place_params_from(p);
  MatchMapsResult R;
  match_maps(ref7, nr, qry7, nq, P, R);
  best_xyyaw[0] = R.x; best_xyyaw[1] = R.y; best_xyyaw[2] = R.yaw;
  for (size_t i = 0; i < R.ref_idx.size(); ++i) { pair_ref_idx[i] = R.ref_idx[i]; pair_qry_idx[i] = R.qry_idx[i]; }
  return R.best_inliers;
}
int orc_find_transformation(const double* ref7, int nr, const double* qry7, int nq, const OrcPlaceParams* p,
                            double* tf16, int* inliers, double* xyzyaw) {
  PlaceParams P = place_params_from(p);
  return find_inter_loop_closure(ref7, nr, qry7, nq, P, tf16, inliers, xyzyaw) ? 1 : 0;
}
void orc_place_default_params(OrcPlaceParams* p) { place_default_params(p); }
int orc_find_intra_loop_closure(const double* meas7, int nm, const double* submap7, int ns, const double* query7, const double* cand7,
                                const OrcPlaceParams* p, double x_half, double y_half, double yaw_half, double* tf16, int* inliers,
                                double* xyzyaw) {
  PlaceParams P = place_params_from(p);
  return find_intra_loop_closure(meas7, nm, submap7, ns, pose_from7(query7), pose_from7(cand7), P, x_half, y_half, yaw_half, tf16,
                                 inliers, xyzyaw) ? 1 : 0;
}

}  // extern "C"
