// Host side of the device-resident factor graph (see host_graph.hpp).
#include <chrono>

#include "host_graph.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace sl {

// Chunks the column blocks of a lambda-block product are cut into at most (split K, launch_border_syrk): the lambda block has nl tile
// rows, (nl + 1) nl tiles — a handful of workgroups for a sum over the whole separator unless it is split; up to 16 chunks while the
// tiles are few, about 256 workgroups in all beyond that (SURVEY 8d's density of relative-pose factors on C4: nl = 6, 42 tiles, 6 chunks —
// unsplit, each of the three products of a pass kept 42 CUs busy for 35 - 60 us).  The same rule on every rank and in a whole pass: the
// order of the sum is part of the bit-stable arithmetic.
static inline int lam_ks_cap(int nl) {
  const int tiles = (nl + 1) * nl;
  if (tiles <= 0) return 1;
  return tiles <= 32 ? 16 : std::max(1, std::min(16, 256 / tiles));
}

thread_local std::string g_last_error;
// Every host thread that drives the library puts its stream-capture mode to THREAD-LOCAL once.  Evidence (round 5,
// gpurun_out/r5_threads_capture.log): with eight rank threads capturing their passes one after the other (g_capture_mtx), a capture came
// back "invalidated" although the capturing thread's own calls all succeeded — while it captured, the OTHER threads were in
// begin_pass / prepare_pass (hipMalloc, hipFree, synchronous copies).  A thread in the default (global) mode that makes such a call checks
// for ongoing captures of every thread and invalidates them; in thread-local mode it only answers for its own.  (hipStreamBeginCapture's
// mode argument covers the capturing thread alone.)
void thread_capture_mode_local() {
  thread_local bool done = false;
  if (done) return;
  hipStreamCaptureMode m = hipStreamCaptureModeThreadLocal;
  (void)hipThreadExchangeStreamCaptureMode(&m);
  done = true;
}

bool hip_ok(hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  g_last_error = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
  return false;
}

// status words of a pass: [0] a landmark block, [1] bit 0 the reduced pose system is not positive definite; [1] bit 1: a workgroup of
// the chained backward substitution gave up waiting for its predecessors — a scheduling stall, not a numerical failure
int decode_status(const int* st) {
  if (st[0]) { g_last_error = "landmark block not positive definite"; return SLIDE_ERR_NOT_SPD; }
  if (st[1] & 1) { g_last_error = "reduced pose system not positive definite"; return SLIDE_ERR_NOT_SPD; }
  if (st[1] & 4) { g_last_error = "joint solve: the conjugate-gradient iteration broke down (coupled pose system not positive definite)"; return SLIDE_ERR_NOT_SPD; }
  if (st[1] & 2) { g_last_error = "backward substitution: a workgroup waited too long for the blocks it depends on (scheduling stall, the update was rejected)"; return SLIDE_ERR_RUNTIME; }
  return SLIDE_OK;
}

// ---- profiler ----------------------------------------------------------------------------------
int Profiler::id_of(const char* name) {
  for (size_t i = 0; i < names.size(); ++i)
    if (names[i] == name) return (int)i;
  names.push_back(name);
  ms.push_back(0.0);
  count.push_back(0);
  return (int)names.size() - 1;
}
void Profiler::begin(int id, hipStream_t s) {
  if (!on) return;
  Rec r;
  r.id = id;
  for (hipEvent_t* e : {&r.a, &r.b}) {
    if (!pool.empty()) { *e = pool.back(); pool.pop_back(); }
    else (void)hipEventCreate(e);
  }
  (void)hipEventRecord(r.a, s);
  recs.push_back(r);
}
void Profiler::end(hipStream_t s) {
  if (!on || recs.empty()) return;
  (void)hipEventRecord(recs.back().b, s);
}
void Profiler::collect() {
  for (auto& r : recs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms[r.id] += t; count[r.id] += 1; }
    pool.push_back(r.a);
    pool.push_back(r.b);
  }
  recs.clear();
}
void Profiler::reset() {
  for (auto& m : ms) m = 0.0;
  for (auto& c : count) c = 0;
}
Profiler::~Profiler() {
  for (auto& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : pool) (void)hipEventDestroy(e);
}

// ---- keys ----------------------------------------------------------------------------------------
uint64_t HostGraph::pose_key(int robot, uint64_t idx) {
  // gtsam::Symbol chars of SemanticFactorGraph::getSymbol (graph.cpp:325-371)
  static const char cs[SLIDE_MAX_ROBOTS] = {'x', 'y', 'z', 'm', 'n', 'o', 'p', 'q', 'r', 's', 't', 'v', 'w'};
  return ((uint64_t)(unsigned char)cs[robot] << 56) | idx;
}
uint64_t HostGraph::lm_key(int cls, uint64_t idx) {
  const char c = cls == SLIDE_CLS_CYLINDER ? 'l' : (cls == SLIDE_CLS_CUBE ? 'c' : 'u');   // graph.h:41-58
  return ((uint64_t)(unsigned char)c << 56) | idx;
}

HostGraph::HostGraph(const slide_params_t& p) : P(p) {}
HostGraph::~HostGraph() {
  if (batch) batch->detach(this);         // the batch must not keep a pointer to a dead graph
  // nothing of this graph may still be in flight when its streams, events and (member destructors, after this body) buffers go
  if (stream) (void)hipStreamSynchronize(stream);
  if (stream2) (void)hipStreamSynchronize(stream2);
  if (gexec) (void)hipGraphExecDestroy(gexec);
  for (auto& pg : phase_graph)
    if (pg.exec) (void)hipGraphExecDestroy(pg.exec);
  for (auto e : ev_dp) (void)hipEventDestroy(e);
  for (auto e : ev_upd) (void)hipEventDestroy(e);
  if (stream2) (void)hipStreamDestroy(stream2);
  if (stream) (void)hipStreamDestroy(stream);
}
int HostGraph::init() {
  thread_capture_mode_local();
  if (P.device >= 0) SL_HIP(hipSetDevice(P.device));
  // SLIDE_NONBLOCKING_STREAMS=1: the graph's own streams do not synchronise with the legacy (null) stream.  Needed when several host
  // threads drive graphs of ONE process side by side (the ranks-as-threads rehearsal of a multi-rank job): a thread-local stream capture
  // on a blocking stream makes every legacy-stream call of any other thread (a synchronous hipMemcpy, a torch kernel) fail with
  // "operation would make the legacy stream depend on a capturing blocking stream".
  static const bool nonblocking = getenv("SLIDE_NONBLOCKING_STREAMS") && getenv("SLIDE_NONBLOCKING_STREAMS")[0] == '1';
  SL_HIP(hipStreamCreateWithFlags(&stream, nonblocking ? hipStreamNonBlocking : hipStreamDefault));
  SL_HIP(hipStreamCreateWithFlags(&stream2, nonblocking ? hipStreamNonBlocking : hipStreamDefault));
  if (nonblocking) {
    static bool said = false;
    unsigned fl = 0;
    SL_HIP(hipStreamGetFlags(stream, &fl));
    if (!said) { said = true; fprintf(stderr, "slide_slam_amd: graph streams are non-blocking (SLIDE_NONBLOCKING_STREAMS=1; flags %u)\n", fl); }
  }
  init_solver_kernels();
  if (d_status.ensure(8, 0, stream, true) != SLIDE_OK) return SLIDE_ERR_HIP;
  return SLIDE_OK;
}

static bool robot_ok(int r) { return r >= 0 && r < SLIDE_MAX_ROBOTS; }
static void put12(const SE3& T, double* z) { to12(T, z); }

int HostGraph::set_prior(int robot, const double* pose7) {
  if (!robot_ok(robot)) return SLIDE_ERR_INVALID;
  const SE3 T = from7(pose7);
  PendFac f{};
  f.type = 0;
  f.k0 = pose_key(robot, 0);
  put12(T, f.z);
  for (int i = 0; i < 6; ++i) f.sigma[i] = P.noise_model_prior_first_pose_vec[i];
  pend_facs.push_back(f);
  PendVar v{};
  v.key = f.k0;
  v.type = VT_POSE;
  put12(T, v.val);
  pend_vars.push_back(v);
  return SLIDE_OK;
}
int HostGraph::add_between_sigma(uint64_t k0, uint64_t k1, const SE3& rel, const double* sigma6) {
  PendFac f{};
  f.type = 1;
  f.k0 = k0;
  f.k1 = k1;
  put12(rel, f.z);
  for (int i = 0; i < 6; ++i) f.sigma[i] = sigma6[i];
  pend_facs.push_back(f);
  return SLIDE_OK;
}
int HostGraph::add_keypose_between(int robot, uint64_t from, uint64_t to, const double* rel7, const double* est7) {
  if (!robot_ok(robot)) return SLIDE_ERR_INVALID;
  const SE3 rel = from7(rel7);
  // odom sigma scaled by max(|t_rel|, noise_floor) (graph.cpp:54-60)
  const double dist = std::max(norm(rel.t), P.noise_floor);
  double s[6];
  for (int i = 0; i < 6; ++i) s[i] = P.noise_model_odom_vec[i] * dist;
  add_between_sigma(pose_key(robot, from), pose_key(robot, to), rel, s);
  PendVar v{};
  v.key = pose_key(robot, to);
  v.type = VT_POSE;
  put12(from7(est7), v.val);
  pend_vars.push_back(v);
  return SLIDE_OK;
}
int HostGraph::add_loop_closure(const double* rel7, uint64_t i1, int r1, uint64_t i2, int r2) {
  if (!robot_ok(r1) || !robot_ok(r2)) return SLIDE_ERR_INVALID;
  double s[6];
  for (int i = 0; i < 6; ++i) s[i] = P.noise_model_odom_vec[i] * 0.01;   // noise_model_closure graphWrapper.cpp:55
  return add_between_sigma(pose_key(r1, i1), pose_key(r2, i2), from7(rel7), s);
}
int HostGraph::add_relative_meas(const double* rel7, uint64_t i1, int r1, uint64_t i2, int r2) {
  if (!robot_ok(r1) || !robot_ok(r2)) return SLIDE_ERR_INVALID;
  const SE3 rel = from7(rel7);
  const double dist = std::max(norm(rel.t), P.noise_floor);   // graph.cpp:251-252
  double s[6];
  for (int i = 0; i < 6; ++i) s[i] = P.noise_model_rel_meas_vec[i] * dist;
  return add_between_sigma(pose_key(r1, i1), pose_key(r2, i2), rel, s);
}
// addRelativeMeasFactor (graph.cpp:247-258) in sharded mode: the other pose is owned by another rank and enters as the
// constant value of ghost slot `slot` (refreshed every pass by dist_phase 20 / 21)
int HostGraph::add_relative_meas_ghost(const double* rel7, uint64_t idx, int robot, int slot, bool local_first) {
  if (!robot_ok(robot) || slot < 0) return SLIDE_ERR_INVALID;
  const SE3 rel = from7(rel7);
  const double dist = std::max(norm(rel.t), P.noise_floor);
  PendFac f{};
  f.type = PF_GHOST;
  f.k0 = pose_key(robot, idx);
  f.k1 = (uint64_t)slot;
  put12(rel, f.z);
  f.z[12] = local_first ? 1.0 : 0.0;
  for (int i = 0; i < 6; ++i) f.sigma[i] = P.noise_model_rel_meas_vec[i] * dist;
  pend_facs.push_back(f);
  return SLIDE_OK;
}
// Ghost slots: a global, rank-independent enumeration of the poses touched by inter-robot relative-pose factors;
// slot i is this rank's pose (own_robot[i], own_idx[i]) or none (own_robot[i] < 0).
int HostGraph::set_ghosts(const int32_t* own_robot, const int64_t* own_idx, int n_slots) {
  topo_dirty = true;      // the slot tables are part of the device view upload_new assembles
  int rc = merge_pending();
  if (rc != SLIDE_OK) return rc;
  rc = upload_new();
  if (rc != SLIDE_OK) return rc;
  h_gslot_pose.assign(n_slots, -1);
  for (int i = 0; i < n_slots; ++i) {
    if (own_robot[i] < 0) continue;
    if (!robot_ok(own_robot[i])) return SLIDE_ERR_INVALID;
    auto it = key2pose.find(pose_key(own_robot[i], (uint64_t)own_idx[i]));
    if (it == key2pose.end()) { g_last_error = "set_ghosts: pose is not in the graph"; return SLIDE_ERR_INVALID; }
    h_gslot_pose[i] = it->second;
  }
  std::vector<double> ident(12 * (size_t)std::max(n_slots, 1), 0.0);
  for (int i = 0; i < n_slots; ++i) ident[12 * (size_t)i] = ident[12 * (size_t)i + 4] = ident[12 * (size_t)i + 8] = 1.0;
  if (d_gslot_pose.ensure(std::max(n_slots, 1), 0, stream) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_ghost_val.ensure(12 * (size_t)std::max(n_slots, 1), 0, stream) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_gslot_pose.upload(h_gslot_pose.data(), 0, n_slots, stream) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_ghost_val.upload(ident.data(), 0, 12 * (size_t)n_slots, stream) != SLIDE_OK) return SLIDE_ERR_HIP;
  SL_HIP(hipStreamSynchronize(stream));
  G.n_gslots = n_slots; G.ghost_val = d_ghost_val.d; G.gslot_pose = d_gslot_pose.d;
  return SLIDE_OK;
}
int HostGraph::add_point_landmark(uint64_t idx, const double* xyz) {
  PendVar v{};
  v.key = lm_key(SLIDE_CLS_ELLIPSOID, idx);
  v.type = VT_POINT;
  v.val[0] = xyz[0]; v.val[1] = xyz[1]; v.val[2] = xyz[2];
  pend_vars.push_back(v);
  return SLIDE_OK;
}
int HostGraph::add_range_bearing(int robot, uint64_t pose_idx, uint64_t lm_idx, const double* bearing, double range) {
  if (!robot_ok(robot)) return SLIDE_ERR_INVALID;
  PendFac f{};
  f.type = FT_BR;
  f.k0 = pose_key(robot, pose_idx);
  f.k1 = lm_key(SLIDE_CLS_ELLIPSOID, lm_idx);
  const double n = std::sqrt(bearing[0] * bearing[0] + bearing[1] * bearing[1] + bearing[2] * bearing[2]);
  for (int i = 0; i < 3; ++i) f.z[i] = bearing[i] / n;   // Pose3().bearing(p) = Unit3(p)  (graph.cpp:163)
  f.z[3] = range;
  pend_facs.push_back(f);
  return SLIDE_OK;
}
int HostGraph::add_cube(int robot, uint64_t pose_idx, uint64_t cube_idx, const SE3& pose, const SE3& cube_world,
                        const double* scale, bool exists) {
  if (!robot_ok(robot)) return SLIDE_ERR_INVALID;
  const SE3 loc = compose(inverse(pose), cube_world);             // cube_global_meas.project(pose.inverse()) graph.cpp:211
  const double dist = std::max(norm(loc.t), 0.1);                 // graph.cpp:214
  PendFac f{};
  f.type = FT_CUBE;
  f.k0 = pose_key(robot, pose_idx);
  f.k1 = lm_key(SLIDE_CLS_CUBE, cube_idx);
  put12(loc, f.z);
  for (int i = 0; i < 3; ++i) f.z[12 + i] = scale[i];
  for (int i = 0; i < 9; ++i) f.sigma[i] = P.noise_model_cube_vec[i] * dist;
  pend_facs.push_back(f);
  if (!exists) {
    PendVar v{};
    v.key = f.k1;
    v.type = VT_CUBE;
    put12(cube_world, v.val);
    for (int i = 0; i < 3; ++i) v.val[12 + i] = scale[i];
    pend_vars.push_back(v);
  }
  return SLIDE_OK;
}
int HostGraph::add_cylinder(int robot, uint64_t pose_idx, uint64_t cyl_idx, const SE3& pose, const double* root,
                            const double* ray, double radius, bool exists) {
  if (!robot_ok(robot)) return SLIDE_ERR_INVALID;
  const SE3 inv = inverse(pose);                                  // cylinder.project(pose.inverse()) graph.cpp:190
  const V3 lr = transform_from(inv, V3{root[0], root[1], root[2]});
  const V3 la = mul(inv.R, V3{ray[0], ray[1], ray[2]});
  PendFac f{};
  f.type = FT_CYL;
  f.k0 = pose_key(robot, pose_idx);
  f.k1 = lm_key(SLIDE_CLS_CYLINDER, cyl_idx);
  f.z[0] = lr.x; f.z[1] = lr.y; f.z[2] = lr.z; f.z[3] = la.x; f.z[4] = la.y; f.z[5] = la.z; f.z[6] = radius;
  pend_facs.push_back(f);
  if (!exists) {
    PendVar v{};
    v.key = f.k1;
    v.type = VT_CYL;
    for (int i = 0; i < 3; ++i) { v.val[i] = root[i]; v.val[3 + i] = ray[i]; }
    v.val[6] = radius;
    pend_vars.push_back(v);
  }
  return SLIDE_OK;
}

// ---- merge fgraph / fvalues into the resident arrays (what isam->update(fgraph, fvalues) ingests) ------
// isam->update(fgraph, fvalues) throws on a factor whose key is in neither the graph nor fvalues and on a value whose key exists
// already; here such an entry is refused (counted in n_rejected, named in the error text), everything else of the batch is merged,
// and the call returns SLIDE_ERR_INVALID so that the caller's solve() does not pass silently.
static std::string key_name(uint64_t k) {
  return std::string(1, (char)(k >> 56)) + std::to_string((unsigned long long)(k & 0x00ffffffffffffffull));
}
int HostGraph::merge_pending() {
  thread_capture_mode_local();
  if (!pend_vars.empty() || !pend_facs.empty()) topo_dirty = true;
  int refused = 0;
  std::string first;
  auto refuse = [&](const char* what, uint64_t key) {
    if (!refused++) first = std::string(what) + " " + key_name(key);
  };
  for (const PendVar& v : pend_vars) {
    if (v.type == VT_POSE) {
      if (key2pose.count(v.key)) { refuse("value inserted twice:", v.key); continue; }   // GTSAM: ValuesKeyAlreadyExists
      key2pose[v.key] = (int)(h_pose_val.size() / 12);
      h_pose_val.insert(h_pose_val.end(), v.val, v.val + 12);
      pose_fids.emplace_back();
      pose_bt.emplace_back();
      h_reach.push_back((int)h_reach.size());
    } else {
      if (key2lm.count(v.key)) { refuse("value inserted twice:", v.key); continue; }
      key2lm[v.key] = (int)h_lm_type.size();
      h_lm_first.push_back(1 << 30);
      h_lm_type.push_back(v.type);
      h_lm_val.insert(h_lm_val.end(), v.val, v.val + 15);
      lm_fids.emplace_back();
      h_lm_last.push_back(-1);
    }
  }
  pend_vars.clear();
  for (const PendFac& f : pend_facs) {
    auto a = key2pose.find(f.k0);
    if (a == key2pose.end()) { refuse("factor on a pose that is not in the graph:", f.k0); continue; }
    dirty_min_pose = std::min(dirty_min_pose, a->second);
    if (f.type == 0) {
      h_pr_pose.push_back(a->second);
      h_pr_z.insert(h_pr_z.end(), f.z, f.z + 12);
      h_pr_sigma.insert(h_pr_sigma.end(), f.sigma, f.sigma + 6);
    } else if (f.type == PF_GHOST) {
      h_gh_pose.push_back(a->second);
      h_gh_slot.push_back((int)f.k1);
      h_gh_first.push_back(f.z[12] != 0.0 ? 1 : 0);
      h_gh_z.insert(h_gh_z.end(), f.z, f.z + 12);
      h_gh_sigma.insert(h_gh_sigma.end(), f.sigma, f.sigma + 6);
    } else if (f.type == 1) {
      auto b = key2pose.find(f.k1);
      if (b == key2pose.end()) { refuse("factor on a pose that is not in the graph:", f.k1); continue; }
      const int bi = (int)h_bt_i.size();
      dirty_min_pose = std::min(dirty_min_pose, b->second);
      h_bt_i.push_back(a->second);
      h_bt_j.push_back(b->second);
      h_bt_z.insert(h_bt_z.end(), f.z, f.z + 12);
      h_bt_sigma.insert(h_bt_sigma.end(), f.sigma, f.sigma + 6);
      pose_bt[a->second].push_back(bi << 1);
      pose_bt[b->second].push_back((bi << 1) | 1);
      csr_bt.touch(std::min(a->second, b->second));
      {
        const int lo = std::min(a->second, b->second), hi = std::max(a->second, b->second);
        h_reach[lo] = std::max(h_reach[lo], hi);
      }
    } else {
      auto b = key2lm.find(f.k1);
      if (b == key2lm.end()) { refuse("factor on a landmark that is not in the graph:", f.k1); continue; }
      const int fid = (int)h_lf_type.size();
      // a new factor changes the landmark's H_ll, which enters the Schur terms of EVERY pose observing it
      dirty_min_pose = std::min(dirty_min_pose, h_lm_first[b->second]);
      h_lm_first[b->second] = std::min(h_lm_first[b->second], a->second);
      h_lf_type.push_back(f.type);
      h_lf_pose.push_back(a->second);
      h_lf_lm.push_back(b->second);
      if (f.type == FT_BR) {
        h_lf_slot.push_back((int)(h_br_z.size() / 4));
        h_br_z.insert(h_br_z.end(), f.z, f.z + 4);
      } else if (f.type == FT_CUBE) {
        h_lf_slot.push_back((int)(h_cu_z.size() / 15));
        h_cu_z.insert(h_cu_z.end(), f.z, f.z + 15);
        h_cu_sigma.insert(h_cu_sigma.end(), f.sigma, f.sigma + 9);
      } else {
        h_lf_slot.push_back((int)(h_cy_z.size() / 7));
        h_cy_z.insert(h_cy_z.end(), f.z, f.z + 7);
      }
      h_lf_joff.push_back(jbuf_used);
      h_lf_eoff.push_back(ebuf_used);
      jbuf_used += lf_jsize(f.type);
      ebuf_used += lf_esize(f.type);
      lm_fids[b->second].push_back(fid);
      csr_lm.touch(b->second);
      csr_pose.touch(a->second);
      lm_first_from = std::min(lm_first_from, b->second);
      // the profile's input, kept current: every observer of the landmark now couples to pose max(last observer, this pose); this pose to it too
      {
        const int pnew = a->second, l = b->second;
        if (pnew > h_lm_last[l]) {
          h_lm_last[l] = pnew;
          for (int x : lm_fids[l]) h_reach[h_lf_pose[x]] = std::max(h_reach[h_lf_pose[x]], pnew);
        }
        h_reach[pnew] = std::max(h_reach[pnew], h_lm_last[l]);
      }
      // keep the pose's list sorted by (landmark id, factor id): the Schur kernel merges two such lists
      std::vector<int>& pl = pose_fids[a->second];
      auto pos = std::upper_bound(pl.begin(), pl.end(), fid, [&](int x, int y) {
        return h_lf_lm[x] != h_lf_lm[y] ? h_lf_lm[x] < h_lf_lm[y] : x < y;
      });
      pl.insert(pos, fid);
    }
  }
  pend_facs.clear();
  if (refused) {
    n_rejected += refused;
    g_last_error = "graph update refused " + std::to_string(refused) + " entr" + (refused == 1 ? "y" : "ies") + " (" + first + ")";
    return SLIDE_ERR_INVALID;
  }
  return SLIDE_OK;
}

constexpr int CHOL_BATCH_HOST_MAX = 8;
// the Cholesky view of a graph's reduced system (joint: with the f32 factor copy of the PCG preconditioner)
static CholSystem chol_system_of(const GraphDev& G, bool joint, float* L32, const int* h_prof, double* ctab, const int* bfirst, const int* h_bfirst) {
  CholSystem c{};
  c.S = G.S; c.ld = G.ld; c.T = G.T; c.Ld = G.Ld; c.Winv = G.Winv; c.yv = G.yv; c.dp = G.dp; c.status = G.status;
  c.L32 = joint ? L32 : nullptr;
  c.h_prof = h_prof; c.prof = G.prof; c.first = G.first; c.ctab = ctab;
  c.nbr = G.arrow ? G.nbr : 0; c.bord = G.bord; c.ldb = G.ldb; c.bfirst = bfirst;
  c.bord_src = G.arrow ? G.bord0 : nullptr;      // (the robots' border product writes bord = bord0 - W W^T)
  c.h_bfirst = (G.arrow && G.nbr > 0) ? h_bfirst : nullptr;
  return c;
}
static int chol_ll_mask();
static bool chol_ll_enabled();
static int chol_pair_mask();
CholBatch::CholBatch(int n_) : n(n_ < 1 ? 1 : (n_ > 8 ? 8 : n_)), sys(n), ev_in(n, nullptr), bufs(n, nullptr), graphs(n, nullptr) {}
void CholBatch::set_graph(int slot, HostGraph* g) {
  std::lock_guard<std::mutex> lk(mtx);
  if (slot < 0 || slot >= n) return;
  graphs[slot] = g;
  pass_dirty = true;
}
void CholBatch::detach(HostGraph* g) {
  std::lock_guard<std::mutex> lk(mtx);
  for (auto& p : graphs)
    if (p == g) { p = nullptr; pass_dirty = true; }
}
void HostGraph::join_batch(CholBatch* b, int slot) {
  pred_valid = false; status_clean = false; cache_pose = -1;      // (outside the streaming update: nothing it left behind can be relied on)
  CholBatch* old = nullptr;
  {
    std::lock_guard<std::mutex> lk(mtx);
    old = batch;
    batch = b;
    batch_slot = slot;
    factor_valid = false;   // (batched passes factor into the same S: a factor left by one is not the streaming path's)
    topo_dirty = true;      // (the joint-solve buffers depend on the batch's setting: upload_new looks again)
  }
  if (old && old != b) old->detach(this);
  if (b) b->set_graph(slot, this);
}
CholBatch::~CholBatch() {
  if (master) (void)hipStreamSynchronize(master);
  for (hipStream_t a : aux) if (a) (void)hipStreamSynchronize(a);
  for (auto& e : part_exec) if (e) (void)hipGraphExecDestroy(e);
  for (HostGraph* g : graphs)
    if (g) { std::lock_guard<std::mutex> gl(g->mtx); if (g->batch == this) g->batch = nullptr; }
  for (hipEvent_t e : ev_in) if (e) (void)hipEventDestroy(e);
  if (ev_out) (void)hipEventDestroy(ev_out);
  if (ev_fork) (void)hipEventDestroy(ev_fork);
  if (pass_exec) (void)hipGraphExecDestroy(pass_exec);
  free_separator();
  free_ll_band_plans();
  if (d_ctr2) (void)hipFree(d_ctr2);
  if (d_pair_tickets) (void)hipFree(d_pair_tickets);
  if (d_syrk_jobs) (void)hipFree(d_syrk_jobs);
  if (d_l2_jobs) (void)hipFree(d_l2_jobs);
  if (d_Gs) (void)hipFree(d_Gs);
  if (d_status_all) (void)hipFree(d_status_all);
  if (ev_aux0) (void)hipEventDestroy(ev_aux0);

  for (hipEvent_t e : ev_aux1) if (e) (void)hipEventDestroy(e);
  for (hipStream_t a : aux) if (a) (void)hipStreamDestroy(a);
  if (master) (void)hipStreamDestroy(master);
  if (d_ctr) (void)hipFree(d_ctr);
}
int CholBatch::factor_solve(int slot, const GraphDev& G, hipStream_t s) {
  if (slot < 0 || slot >= n) return SLIDE_ERR_INVALID;
  {
    std::lock_guard<std::mutex> lk(mtx);
    sys[slot] = chol_system_of(G, pcg_iters > 0 && G.n_slots > 0, graphs[slot]->d_L32.d, graphs[slot]->h_prof.data(), graphs[slot]->d_ctab.d,
                               graphs[slot]->d_bfirst.d, graphs[slot]->h_bfirst.data());
  }
  return rendezvous(slot, s, false, 0);
}
int CholBatch::all_reduce(int slot, double* d_buf, int count, hipStream_t s) {
  if (slot < 0 || slot >= n) return SLIDE_ERR_INVALID;
  {
    std::lock_guard<std::mutex> lk(mtx);
    bufs[slot] = d_buf;
  }
  return rendezvous(slot, s, true, count);
}
// Every joined graph's thread arrives with its work enqueued on its own stream `s`; the last one enqueues the joint work on the
// batch's stream behind all of them; every stream continues behind that.
int CholBatch::rendezvous(int slot, hipStream_t s, bool reduce, int count) {
  std::unique_lock<std::mutex> lk(mtx);
  if (!master) {
    SL_HIP(hipStreamCreateWithFlags(&master, hipStreamNonBlocking));
    SL_HIP(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming));
  }
  if (!ev_in[slot]) SL_HIP(hipEventCreateWithFlags(&ev_in[slot], hipEventDisableTiming));
  SL_HIP(hipEventRecord(ev_in[slot], s));
  const unsigned long long my_gen = generation;
  if (++arrived == n) {
    int rc = SLIDE_OK;
    if (!reduce) {
      int Tmax = 0;
      for (const CholSystem& c : sys) Tmax = c.T > Tmax ? c.T : Tmax;
      if (CHOL_BATCH_HOST_MAX * (Tmax + 2) > ctr_cap) {
        if (d_ctr) { (void)hipStreamSynchronize(master); (void)hipFree(d_ctr); d_ctr = nullptr; }
        ctr_cap = CHOL_BATCH_HOST_MAX * 2 * (Tmax + 2);
        if (hipMalloc(reinterpret_cast<void**>(&d_ctr), ctr_cap * sizeof(int)) != hipSuccess) rc = SLIDE_ERR_HIP;
        else if (hipMemsetAsync(d_ctr, 0, ctr_cap * sizeof(int), master) != hipSuccess) rc = SLIDE_ERR_HIP;
      }
    }
    for (int i = 0; rc == SLIDE_OK && i < n; ++i)
      if (hipStreamWaitEvent(master, ev_in[i], 0) != hipSuccess) rc = SLIDE_ERR_HIP;
    if (rc == SLIDE_OK) {
      if (reduce) launch_sum_bcast(bufs.data(), n, count, master);
      else launch_chol_batch(sys.data(), n, d_ctr, master);
      if (hipEventRecord(ev_out, master) != hipSuccess) rc = SLIDE_ERR_HIP;
    }
    gen_status = rc;
    arrived = 0;
    ++generation;
    cv.notify_all();
  } else {
    if (!cv.wait_for(lk, std::chrono::seconds(60), [&] { return generation != my_gen; })) {
      --arrived;
      g_last_error = "batch rendezvous: the other graphs of the batch did not arrive";
      return SLIDE_ERR_RUNTIME;
    }
  }
  if (gen_status != SLIDE_OK) return gen_status;
  SL_HIP(hipStreamWaitEvent(s, ev_out, 0));
  return SLIDE_OK;
}

// ---- the whole pass of all joined graphs as one captured graph ------------------------------------------------------------------
// device-side tables of a pass (systems of the batched factorisation, the graphs' device views, the work counters)
int CholBatch::prepare_pass() {
  for (int i = 0; i < n; ++i)
    if (!ev_in[i]) SL_HIP(hipEventCreateWithFlags(&ev_in[i], hipEventDisableTiming));
  if (!ev_fork) SL_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  int Tmax = 0;
  hG.resize(n);
  for (int i = 0; i < n; ++i) {
    const GraphDev& G = graphs[i]->G;
    sys[i] = chol_system_of(G, pcg_iters > 0 && G.n_slots > 0, graphs[i]->d_L32.d, graphs[i]->h_prof.data(), graphs[i]->d_ctab.d, graphs[i]->d_bfirst.d,
                            graphs[i]->h_bfirst.data());
    Tmax = G.T > Tmax ? G.T : Tmax;
    hG[i] = G;
    hG[i].save_S0 = (pcg_iters > 0 && G.n_slots > 0) ? 1 : 0;      // the batched Schur assembly writes S0 itself
  }
  if (CHOL_BATCH_HOST_MAX * (Tmax + 2) > ctr_cap) {
    if (d_ctr) { SL_HIP(hipStreamSynchronize(master)); SL_HIP(hipFree(d_ctr)); d_ctr = nullptr; }
    ctr_cap = CHOL_BATCH_HOST_MAX * 2 * (Tmax + 2);
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_ctr), ctr_cap * sizeof(int)));
    SL_HIP(hipMemsetAsync(d_ctr, 0, ctr_cap * sizeof(int), master));
    SL_HIP(hipStreamSynchronize(master));
  }
  if (!d_pair_tickets && chol_pair_mask() != 0) {
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_pair_tickets), 12 * CHOL_STEP_BATCH_MAX * sizeof(int)));
    SL_HIP(hipMemsetAsync(d_pair_tickets, 0, 12 * CHOL_STEP_BATCH_MAX * sizeof(int), master));
    SL_HIP(hipStreamSynchronize(master));
  }
  if (!d_Gs) SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_Gs), CHOL_BATCH_HOST_MAX * sizeof(GraphDev)));
  if (!d_status_all) SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_status_all), CHOL_BATCH_HOST_MAX * 8 * sizeof(int)));
  SL_HIP(hipMemcpyAsync(d_Gs, hG.data(), n * sizeof(GraphDev), hipMemcpyHostToDevice, master));      // (not the legacy stream: other host threads may be capturing)
  SL_HIP(hipStreamSynchronize(master));
  // exact joint passes: the systems the steps run on — every graph's segments (views of its S; the whole band when it is not cut) — and
  // the second-level systems of the graphs that are cut (the separator poses' block inside the border block, the rest of the border
  // as its border)
  seg_sys.clear(); l2_sys.clear(); l2_graph.clear(); seg_hord.clear();
  free_ll_band_plans();
  if (arrow && hG[0].n_slots > 0) {
    for (int i = 0; i < n; ++i) {
      HostGraph* g = graphs[i];
      const GraphDev& G = hG[i];
      if (G.nsep <= 0 || g->segs.empty()) { seg_sys.push_back(sys[i]); seg_hord.push_back(nullptr); continue; }
      for (size_t k = 0; k < g->segs.size(); ++k) {
        const HostGraph::Seg sg = g->segs[k];
        CholSystem c = sys[i];
        c.S = G.S + (size_t)sg.t0 * NB * G.ld + (size_t)sg.t0 * NB;
        c.T = sg.t1 - sg.t0;
        c.Ld = G.Ld + (size_t)sg.t0 * NB * NB; c.Winv = G.Winv + (size_t)sg.t0 * 1024;
        c.yv = G.yv + (size_t)sg.t0 * NB; c.dp = G.dp + (size_t)sg.t0 * NB;
        c.h_prof = g->seg_prof[k].data(); c.prof = g->d_seg_prof.d + g->seg_prof_off[k]; c.first = nullptr;
        c.b0 = G.T - sg.t0; c.kofs = sg.t0;
        c.L32 = nullptr; c.ctab = nullptr;
        const int* hord = nullptr;
        if (k < g->seg_ord.size() && !getenv("SLIDE_SEG_PLAIN")) {      // the segment's own set and order of active border rows
          c.h_bfirst = g->seg_sfirst[k].data();
          c.ord = g->d_seg_ord.d + k * (size_t)G.nbr;
          hord = g->seg_ord[k].data();
        }
        seg_sys.push_back(c);
        seg_hord.push_back(hord);
      }
      if (!g->seg_tab.empty() && !getenv("SLIDE_SEG_PLAIN")) sys[i].segtab = g->d_seg_tab.d;      // (the border product sums per segment)
      CholSystem l2{};
      l2.S = G.bord; l2.ld = G.ldb; l2.T = G.nsep; l2.Ld = g->d_Ld2.d; l2.Winv = g->d_Winv2.d; l2.yv = g->d_yv2.d; l2.dp = g->d_dp2.d;
      l2.status = G.status; l2.nbr = G.nbr - G.nsep;
      l2.bord = G.bord + (size_t)G.nsep * NB * G.ldb + (size_t)G.nsep * NB; l2.ldb = G.ldb;
      l2_sys.push_back(l2);
      l2_graph.push_back(i);
    }
    {
      // the border product's job table: its order only schedules (the kernel takes the sums' ranges from the device's bfirst)
      std::vector<std::pair<int, int>> jl;
      for (int i = 0; i < n; ++i) {
        const int nb = sys[i].nbr, T = sys[i].T;
        const int* hb = sys[i].h_bfirst;
        if (nb > 1023 || n > 2047) { jl.clear(); break; }
        const std::vector<int>& stb = graphs[i]->seg_tab;
        const bool seg = sys[i].segtab != nullptr && !stb.empty();
        for (int jb = 0; jb < nb; ++jb)
          for (int ib = jb; ib <= nb; ++ib) {
            int K = T - (hb ? std::max(hb[ib], hb[jb]) : 0);
            if (seg) {      // (the lengths only order the table)
              K = 0;
              const int NS = stb[0];
              for (int q = 0; q < NS; ++q) {
                const int* sf = stb.data() + 1 + NS + (size_t)q * (nb + 1);
                const int c0 = std::max(sf[ib], sf[jb]);
                if (c0 < stb[1 + q]) K += stb[1 + q] - c0;
              }
            }
            jl.emplace_back(K, i << 20 | ib << 10 | jb);
          }
      }
      std::stable_sort(jl.begin(), jl.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first; });
      if (getenv("SLIDE_SYRK_XCD") && !jl.empty()) {
        // (experiment, off: workgroup p of a launch runs on XCD p % 8; position p takes the next (longest) job of "its" robot while that
        // robot has any left, else of the robot with the most left, so that the jobs of a robot — which read the same ~20 row panels —
        // meet in ONE L2.  tools/syrk_bench.hip on eight equal robots: 0.632 -> 0.604 ms; on C4's unequal ones 0.435 -> 0.468 ms)
        std::vector<std::vector<std::pair<int, int>>> q(n);
        for (const auto& e : jl) q[e.second >> 20].push_back(e);
        std::vector<size_t> head(n, 0);
        std::vector<std::pair<int, int>> out;
        out.reserve(jl.size());
        for (size_t p = 0; p < jl.size(); ++p) {
          int r = (int)(p % 8) % n;
          if (head[r] >= q[r].size()) {
            size_t best = 0;
            for (int t = 0; t < n; ++t)
              if (q[t].size() - head[t] > best) { best = q[t].size() - head[t]; r = t; }
          }
          out.push_back(q[r][head[r]++]);
        }
        jl.swap(out);
      }
      static const bool syrk_plain_order = getenv("SLIDE_SYRK_ORDER") && !strcmp(getenv("SLIDE_SYRK_ORDER"), "plain");
      if (!syrk_plain_order && !getenv("SLIDE_SYRK_XCD") && n >= 4 && !jl.empty()) {
        // Round 4, the default for four or more systems: the same idea with BALANCED queues — every robot's jobs cut into four strips of
        // tile columns of about equal work, the 4 n strips packed onto the eight XCD queues longest first (LPT), each queue longest job
        // first, position p of the table from queue p % 8 — an exhausted queue's positions go to the queue with the most work left.
        // C4 on the real pass: 0.211 -> 0.196 ms per launch (by robot without balancing: 0.200); the tiles' arithmetic is untouched,
        // only where and when a tile is computed.  SLIDE_SYRK_ORDER=plain: longest first over all robots, as in round 3.
        const double PRO = 0.7;      // a job's prologue + epilogue in block columns of work
        std::vector<std::vector<std::pair<int, int>>> byrob(n);
        for (const auto& e : jl) byrob[e.second >> 20].push_back(e);
        struct Unit { double w; std::vector<std::pair<int, int>> jobs; };
        std::vector<Unit> units;
        for (int i = 0; i < n; ++i) {
          auto& v = byrob[i];
          std::stable_sort(v.begin(), v.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return (a.second & 1023) < (b.second & 1023); });
          double tot = 0;
          for (const auto& e : v) tot += e.first + PRO;
          const int U = 4;
          size_t k = 0;
          for (int u = 0; u < U && k < v.size(); ++u) {
            Unit un; un.w = 0;
            const double want = tot * (u + 1) / U;
            double run = 0;
            for (size_t t = 0; t < k; ++t) run += v[t].first + PRO;
            while (k < v.size() && (u == U - 1 || run + 0.5 * (v[k].first + PRO) <= want)) { un.jobs.push_back(v[k]); un.w += v[k].first + PRO; run += v[k].first + PRO; ++k; }
            if (!un.jobs.empty()) units.push_back(std::move(un));
          }
        }
        std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.w > b.w; });
        std::vector<std::vector<std::pair<int, int>>> q(8);
        std::vector<double> load(8, 0.0);
        for (auto& un : units) {
          int best = 0;
          for (int x = 1; x < 8; ++x) if (load[x] < load[best]) best = x;
          load[best] += un.w;
          for (auto& e : un.jobs) q[best].push_back(e);
        }
        for (auto& v : q) std::stable_sort(v.begin(), v.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first; });
        if (getenv("SLIDE_SYRK_DEBUG")) {
          fprintf(stderr, "syrk queues:");
          for (int x = 0; x < 8; ++x) fprintf(stderr, " %.0f (%zu)", load[x], q[x].size());
          fprintf(stderr, "\n");
        }
        std::vector<size_t> head(8, 0);
        std::vector<double> left = load;
        std::vector<std::pair<int, int>> out;
        out.reserve(jl.size());
        for (size_t p = 0; p < jl.size(); ++p) {
          int x = (int)(p % 8);
          if (head[x] >= q[x].size()) {
            double best = -1;
            for (int t = 0; t < 8; ++t)
              if (head[t] < q[t].size() && left[t] > best) { best = left[t]; x = t; }
          }
          left[x] -= q[x][head[x]].first + PRO;
          out.push_back(q[x][head[x]++]);
        }
        jl.swap(out);
      }
      n_syrk_jobs = (int)jl.size();
      if (n_syrk_jobs > syrk_jobs_cap) {
        if (d_syrk_jobs) { SL_HIP(hipStreamSynchronize(master)); SL_HIP(hipFree(d_syrk_jobs)); d_syrk_jobs = nullptr; }
        syrk_jobs_cap = 2 * n_syrk_jobs;
        SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_syrk_jobs), (size_t)syrk_jobs_cap * sizeof(int)));
      }
      if (n_syrk_jobs > 0) {
        std::vector<int> codes(jl.size());
        for (size_t k = 0; k < jl.size(); ++k) codes[k] = jl[k].second;
        SL_HIP(hipStreamSynchronize(master));
        SL_HIP(hipMemcpyAsync(d_syrk_jobs, codes.data(), codes.size() * sizeof(int), hipMemcpyHostToDevice, master)); SL_HIP(hipStreamSynchronize(master));
      }
      const char* e = getenv("SLIDE_SYRK_LDS");
      syrk_lds_pad = e ? atoi(e) : 0;
      if (getenv("SLIDE_SYRK_PLAIN")) n_syrk_jobs = 0;
    }
    {
      // the second level's border products: one workgroup per real tile (the plain grid launches as many idle ones)
      std::vector<int> codes;
      for (size_t i = 0; i < l2_sys.size() && i < 2047; ++i) {
        const int nb = l2_sys[i].nbr;
        if (nb > 1023) { codes.clear(); break; }
        for (int jb = 0; jb < nb; ++jb)
          for (int ib = jb; ib <= nb; ++ib) codes.push_back((int)i << 20 | ib << 10 | jb);
      }
      n_l2_jobs = (int)codes.size();
      if (n_l2_jobs > l2_jobs_cap) {
        if (d_l2_jobs) { SL_HIP(hipStreamSynchronize(master)); SL_HIP(hipFree(d_l2_jobs)); d_l2_jobs = nullptr; }
        l2_jobs_cap = 2 * n_l2_jobs;
        SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_l2_jobs), (size_t)l2_jobs_cap * sizeof(int)));
      }
      if (n_l2_jobs > 0) {
        SL_HIP(hipStreamSynchronize(master));
        SL_HIP(hipMemcpyAsync(d_l2_jobs, codes.data(), codes.size() * sizeof(int), hipMemcpyHostToDevice, master)); SL_HIP(hipStreamSynchronize(master));
      }
    }
    if (!d_ctr2) {
      SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_ctr2), 64 * sizeof(int)));
      SL_HIP(hipMemsetAsync(d_ctr2, 0, 64 * sizeof(int), master)); SL_HIP(hipStreamSynchronize(master));
    }
    if ((int)seg_sys.size() > 8 * CHOL_BATCH_HOST_MAX || (int)l2_sys.size() > CHOL_BATCH_HOST_MAX) { g_last_error = "exact joint step: too many segment systems"; return SLIDE_ERR_CAPACITY; }
    if (chol_ll_enabled()) {
      SL_HIP(hipStreamSynchronize(master));
      if (chol_ll_mask() & 1) ll_seg = chol_ll_plan_create(seg_sys.data(), (int)seg_sys.size(), seg_hord.data(), master);
      if (!l2_sys.empty() && (chol_ll_mask() & 2)) ll_l2 = chol_ll_plan_create(l2_sys.data(), (int)l2_sys.size(), nullptr, master);
      if (((chol_ll_mask() & 1) && !ll_seg) || (!l2_sys.empty() && (chol_ll_mask() & 2) && !ll_l2)) { g_last_error = "exact joint step: the left-looking factorisation's tables could not be allocated"; return SLIDE_ERR_HIP; }
    }
    return prepare_separator();
  }
  return SLIDE_OK;
}
// ---- exact joint step: the separator system of all shared landmarks -------------------------------------------------------------------
// SLIDE_CHOL_LL: bit mask of the levels of an exact joint pass that run the left-looking persistent factorisation (k_chol_ll) instead
// of one step launch per block column — 1: the bands' segments, 2: the bands' second level, 4: the separator's leaves, 8: its top block
static int chol_ll_mask() {
  static const int m = getenv("SLIDE_CHOL_LL") ? atoi(getenv("SLIDE_CHOL_LL")) : 0;      // (opt-in until measured)
  return m;
}
static bool chol_ll_enabled() { return chol_ll_mask() != 0; }
// SLIDE_CHOL_PAIR: bit mask (same levels) of what is factored TWO block columns per launch (k_chol_pair_batched, chol_kernels.hip):
// half the chain-bound launches of a pass in a row.  Only the exact joint passes ask for it (the streaming path's single systems
// keep the step kernels: a pair launch's redundant work pays on chain-bound launches only).
static int chol_pair_mask() {
  static const int m = getenv("SLIDE_CHOL_PAIR") ? atoi(getenv("SLIDE_CHOL_PAIR")) : 0;
  return m;
}
void CholBatch::free_ll_band_plans() {
  for (CholLLPlan** p : {&ll_seg, &ll_l2}) if (*p) { if (master) (void)hipStreamSynchronize(master); chol_ll_plan_destroy(*p); *p = nullptr; }
}
void CholBatch::free_ll_sep_plans() {
  for (CholLLPlan** p : {&ll_leaves, &ll_leaf_own[0], &ll_leaf_own[1], &ll_top}) if (*p) { if (master) (void)hipStreamSynchronize(master); chol_ll_plan_destroy(*p); *p = nullptr; }
}
void CholBatch::sep_leaf_systems(CholSystem* lv) const {
  const int ld_s = (sep_Ts + sep_nl + 1) * NB;
  const int sTa = sep_leafT[0], sTL = sTa + sep_leafT[1], sTt = sep_Ts - sTL;
  for (int b = 0; b < 2; ++b) {
    const int t0 = b ? sTa : 0;
    CholSystem c{};
    c.S = sepS + (size_t)t0 * NB * ld_s + (size_t)t0 * NB; c.ld = ld_s; c.T = sep_leafT[b];
    c.Ld = sep_Ld + (size_t)t0 * NB * NB; c.Winv = sep_Winv + (size_t)t0 * 1024; c.yv = sep_yv + (size_t)t0 * NB; c.dp = sep_dp + (size_t)t0 * NB;
    c.status = sep_status; c.h_prof = h_leaf_prof[b].data(); c.prof = d_leaf_prof + (b ? sTa : 0);
    c.nbr = sTt + sep_nl; c.b0 = sTL - t0; c.kofs = t0;
    lv[b] = c;
  }
}
CholSystem CholBatch::sep_top_system() const {
  const int ld_s = (sep_Ts + sep_nl + 1) * NB;
  const int TL = sep_dissected() ? sep_leafT[0] + sep_leafT[1] : 0;
  CholSystem c{};
  c.S = sepS + (size_t)TL * NB * ld_s + (size_t)TL * NB; c.ld = ld_s; c.T = sep_Ts - TL;
  c.Ld = sep_Ld + (size_t)TL * NB * NB; c.Winv = sep_Winv + (size_t)TL * 1024; c.yv = sep_yv + (size_t)TL * NB; c.dp = sep_dp + (size_t)TL * NB;
  c.status = sep_status; c.nbr = sep_nl;
  if (!sep_dissected() && sep_prof_on) c.h_prof = h_sep_prof.data();
  return c;
}
void CholBatch::free_separator() {
  free_ll_sep_plans();
  if (sepS) (void)hipFree(sepS);
  sepS = nullptr; sep_len = 0;
  for (double** p : {&sep_Ld, &sep_Winv, &sep_yv, &sep_dp, &sep_bord, &lamS, &lam_Ld, &lam_Winv, &lam_yv, &lam_dp, &lam_scratch}) if (*p) { (void)hipFree(*p); *p = nullptr; }
  for (int** p : {&sep_status, &sep_ctr, &d_sep_off, &lam_status, &lam_ctr, &d_sep_prof, &d_leaf_prof, &sep_ctr2, &d_sep_jobs, &d_sep_tmask}) if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (sep_scratch) { (void)hipFree(sep_scratch); sep_scratch = nullptr; }
  for (double** p : {&sep_top2, &sep_bord2}) if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (d_sep_jobs2) { (void)hipFree(d_sep_jobs2); d_sep_jobs2 = nullptr; }
  if (d_lam_jobs2) { (void)hipFree(d_lam_jobs2); d_lam_jobs2 = nullptr; }
  sep_cap = 0; lam_cap = -1;
}
void CholBatch::set_segments(int n) {
  std::lock_guard<std::mutex> pl(pass_mtx);
  std::vector<HostGraph*> gs;
  {
    std::lock_guard<std::mutex> lk(mtx);
    n_seg = n < 1 ? 1 : (n > 8 ? 8 : n);
    pass_dirty = true;
    gs.assign(graphs.begin(), graphs.end());
  }
  for (HostGraph* g : gs)
    if (g) { std::lock_guard<std::mutex> gl(g->mtx); g->topo_dirty = true; }
}
int CholBatch::set_separator_profile(const int32_t* prof, int n) {
  std::lock_guard<std::mutex> pl(pass_mtx);
  std::lock_guard<std::mutex> lk(mtx);
  for (int c = 0; c < n; ++c)
    if (prof[c] < c || prof[c] >= n || (c && prof[c] < prof[c - 1])) { g_last_error = "separator profile: prof[c] must be monotone with c <= prof[c] < n"; return SLIDE_ERR_INVALID; }
  h_sep_prof.assign(prof, prof + n);
  pass_dirty = true;
  return SLIDE_OK;
}
int CholBatch::set_separator_blocks(int Ta, int Tb, int used_a, int used_b) {
  std::lock_guard<std::mutex> pl(pass_mtx);
  std::lock_guard<std::mutex> lk(mtx);
  if (Ta < 0 || Tb < 0 || (Ta > 0) != (Tb > 0) || used_a < 0 || used_b < 0 || used_a > Ta * NB || used_b > Tb * NB || (Ta > 0 && (used_a <= (Ta - 1) * NB || used_b <= (Tb - 1) * NB))) {
    g_last_error = "separator blocks: two leaf blocks of Ta, Tb > 0 tile columns whose last tiles hold at least one used coordinate (or 0, 0: not dissected)";
    return SLIDE_ERR_INVALID;
  }
  sep_leafT[0] = Ta; sep_leafT[1] = Tb; sep_used[0] = used_a; sep_used[1] = used_b;
  pass_dirty = true;
  return SLIDE_OK;
}
int CholBatch::set_separator_owner(int leaf, bool leader) {
  std::lock_guard<std::mutex> pl(pass_mtx);
  std::lock_guard<std::mutex> lk(mtx);
  if (leaf < -1 || leaf > 1) { g_last_error = "separator owner: leaf 0, 1 or -1 (none)"; return SLIDE_ERR_INVALID; }
  sep_owner = leaf; sep_leader = leader;
  pass_dirty = true;
  return SLIDE_OK;
}
// packed exchange buffer of a dissected layout: tile columns [0, Ta) | [Ta, Ta + Tb) | [Ta + Tb, Ts + nl), each a contiguous run
void CholBatch::sep_segment(int ms, int lam, int Ta, int Tb, int which, long long* off, long long* len) {
  const long long Tt = (ms + NB - 1) / NB + (lam + NB - 1) / NB;
  auto upto = [&](long long tj) {      // doubles in front of tile column tj (sep_packed_addr)
    return (long long)NB * NB * (tj * (Tt + 1) - tj * (tj - 1) / 2 - (long long)Tb * std::min<long long>(tj, Ta));
  };
  const long long b[4] = {0, Ta, (long long)Ta + Tb, Tt};
  *off = upto(b[which]);
  *len = upto(b[which + 1]) - upto(b[which]);
}
int CholBatch::set_arrow(bool on, double* sep_buf, long long len) {
  std::lock_guard<std::mutex> pl(pass_mtx);
  std::vector<HostGraph*> gs;
  {
    std::lock_guard<std::mutex> lk(mtx);
    if (master) (void)hipStreamSynchronize(master);
    sep_x = sep_buf; sep_x_len = sep_buf ? len : 0;
    arrow = on;
    pass_dirty = true;
    gs.assign(graphs.begin(), graphs.end());
  }
  for (HostGraph* g : gs)
    if (g) { std::lock_guard<std::mutex> gl(g->mtx); g->topo_dirty = true; }
  return SLIDE_OK;
}
// the separator's buffers for the joined graphs' current slots (called with the graphs up to date)
int CholBatch::prepare_separator() {
  for (int i = 0; i < n; ++i) {
    HostGraph* g = graphs[i];
    if (!g->arrow_on() || g->h_sep_off != graphs[0]->h_sep_off) {
      g_last_error = "exact joint step: every graph of the batch needs the shared slots and the same separator offsets (slide_graph_set_shared, slide_graph_set_separator)";
      return SLIDE_ERR_INVALID;
    }
  }
  const std::vector<int>& off = graphs[0]->h_sep_off;
  sep_m = off.back();
  sep_Ts = (sep_m + NB - 1) / NB;
  sep_lam = graphs[0]->lam_total;
  for (int i = 1; i < n; ++i)
    if (graphs[i]->lam_total != sep_lam) { g_last_error = "exact joint step: the graphs disagree on the number of relative-pose measurements"; return SLIDE_ERR_INVALID; }
  sep_nl = (sep_lam + NB - 1) / NB;
  const long long need = (long long)(sep_Ts + sep_nl + 1) * NB * sep_Ts * NB;
  if (sep_nl != lam_cap) {
    SL_HIP(hipStreamSynchronize(master));
    for (double** p : {&sep_bord, &lamS, &lam_Ld, &lam_Winv, &lam_yv, &lam_dp, &lam_scratch}) if (*p) { SL_HIP(hipFree(*p)); *p = nullptr; }
    for (int** p : {&lam_status, &lam_ctr}) if (*p) { SL_HIP(hipFree(*p)); *p = nullptr; }
    lam_cap = sep_nl;
    if (sep_nl > 0) {
      const size_t nb = (size_t)(sep_nl + 1) * NB * sep_nl * NB;
      SL_HIP(hipMalloc(reinterpret_cast<void**>(&sep_bord), nb * sizeof(double)));
      if (lam_ks_cap(sep_nl) > 1)
        SL_HIP(hipMalloc(reinterpret_cast<void**>(&lam_scratch), 2 * (size_t)(sep_nl + 1) * sep_nl * (lam_ks_cap(sep_nl) - 1) * NB * NB * sizeof(double)));      // (x 2: both leaves' partial tiles in one launch)
      {
        std::vector<int> codes;      // the lambda block's tiles + right-hand-side row, for both leaves as two systems (launch_border_syrk_jobs)
        for (int sy = 0; sy < 2; ++sy)
          for (int jb = 0; jb < sep_nl; ++jb)
            for (int ib = jb; ib <= sep_nl; ++ib) codes.push_back(sy << 20 | ib << 10 | jb);
        if (d_lam_jobs2) { SL_HIP(hipFree(d_lam_jobs2)); d_lam_jobs2 = nullptr; }
        n_lam_jobs2 = (int)codes.size();
        SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_lam_jobs2), codes.size() * sizeof(int)));
        SL_HIP(hipMemcpyAsync(d_lam_jobs2, codes.data(), codes.size() * sizeof(int), hipMemcpyHostToDevice, master)); SL_HIP(hipStreamSynchronize(master));
      }
      SL_HIP(hipMalloc(reinterpret_cast<void**>(&lamS), nb * sizeof(double)));
      SL_HIP(hipMemsetAsync(sep_bord, 0, nb * sizeof(double), master)); SL_HIP(hipStreamSynchronize(master));
      SL_HIP(hipMemsetAsync(lamS, 0, nb * sizeof(double), master)); SL_HIP(hipStreamSynchronize(master));
      SL_HIP(hipMalloc(reinterpret_cast<void**>(&lam_Ld), (size_t)sep_nl * NB * NB * sizeof(double)));
      SL_HIP(hipMalloc(reinterpret_cast<void**>(&lam_Winv), (size_t)sep_nl * 1024 * sizeof(double)));
      SL_HIP(hipMalloc(reinterpret_cast<void**>(&lam_yv), (size_t)sep_nl * NB * sizeof(double)));
      SL_HIP(hipMalloc(reinterpret_cast<void**>(&lam_dp), (size_t)sep_nl * NB * sizeof(double)));
      SL_HIP(hipMalloc(reinterpret_cast<void**>(&lam_status), 8 * sizeof(int)));
      SL_HIP(hipMalloc(reinterpret_cast<void**>(&lam_ctr), ((size_t)sep_nl + 2) * sizeof(int)));
      SL_HIP(hipMemsetAsync(lam_ctr, 0, ((size_t)sep_nl + 2) * sizeof(int), master)); SL_HIP(hipStreamSynchronize(master));
      SL_HIP(hipMemsetAsync(lam_status, 0, 8 * sizeof(int), master)); SL_HIP(hipStreamSynchronize(master));
    }
  }
  if (sep_x && sep_x_len < sep_buffer_len(sep_m, sep_lam)) { g_last_error = "exact joint step: the separator exchange buffer is too small (slide_chol_batch_sep_buffer_len)"; return SLIDE_ERR_INVALID; }
  if (!sepS || sep_len < need) {
    if (sepS) { SL_HIP(hipStreamSynchronize(master)); SL_HIP(hipFree(sepS)); sepS = nullptr; }
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&sepS), std::max<long long>(need, 1) * sizeof(double)));
    sep_len = need;
  }
  if (sep_Ts > sep_cap) {
    SL_HIP(hipStreamSynchronize(master));
    for (double** p : {&sep_Ld, &sep_Winv, &sep_yv, &sep_dp}) if (*p) { SL_HIP(hipFree(*p)); *p = nullptr; }
    for (int** p : {&sep_status, &sep_ctr}) if (*p) { SL_HIP(hipFree(*p)); *p = nullptr; }
    sep_cap = sep_Ts;
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&sep_Ld), (size_t)sep_cap * NB * NB * sizeof(double)));
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&sep_Winv), (size_t)sep_cap * 1024 * sizeof(double)));
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&sep_yv), (size_t)sep_cap * NB * sizeof(double)));
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&sep_dp), (size_t)sep_cap * NB * sizeof(double)));
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&sep_status), 8 * sizeof(int)));
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&sep_ctr), ((size_t)sep_cap + 2) * sizeof(int)));
    SL_HIP(hipMemsetAsync(sep_ctr, 0, ((size_t)sep_cap + 2) * sizeof(int), master)); SL_HIP(hipStreamSynchronize(master));
    SL_HIP(hipMemsetAsync(sep_status, 0, 8 * sizeof(int), master)); SL_HIP(hipStreamSynchronize(master));
  }
  // zero once: the strict upper triangle and the idle rows of the right-hand-side tile are never written by the gather
  SL_HIP(hipMemsetAsync(sepS, 0, (size_t)need * sizeof(double), master)); SL_HIP(hipStreamSynchronize(master));
  if (sep_x) SL_HIP(hipMemsetAsync(sep_x, 0, (size_t)sep_buffer_len(sep_m, sep_lam) * sizeof(double), master)); SL_HIP(hipStreamSynchronize(master));
  // tile profile of the landmark part (the caller's, from the robots' observer sets — two shared landmarks couple only if some robot
  // observes both); absent or of another size: dense
  if (d_sep_prof) { SL_HIP(hipFree(d_sep_prof)); d_sep_prof = nullptr; }
  sep_prof_on = (int)h_sep_prof.size() == sep_Ts && sep_Ts > 0;
  if (sep_prof_on) {
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_sep_prof), (size_t)sep_Ts * sizeof(int)));
    SL_HIP(hipMemcpyAsync(d_sep_prof, h_sep_prof.data(), (size_t)sep_Ts * sizeof(int), hipMemcpyHostToDevice, master)); SL_HIP(hipStreamSynchronize(master));
  }
  if (sep_dissected()) {
    const int Ta = sep_leafT[0], Tb = sep_leafT[1], TL = Ta + Tb, Tt = sep_Ts - TL, nb = Tt + sep_nl;
    if (Tt <= 0 || !sep_prof_on || h_sep_prof[Ta - 1] != Ta - 1 || h_sep_prof[TL - 1] != TL - 1 || sep_m <= TL * NB) {
      g_last_error = "separator blocks: they need a profile (slide_chol_batch_set_separator_profile) that ends each leaf block at its own last tile row, and a top block behind them";
      return SLIDE_ERR_INVALID;
    }
    SL_HIP(hipStreamSynchronize(master));
    for (int** p : {&d_leaf_prof, &sep_ctr2, &d_sep_jobs}) if (*p) { SL_HIP(hipFree(*p)); *p = nullptr; }
    if (sep_scratch) { SL_HIP(hipFree(sep_scratch)); sep_scratch = nullptr; }
    std::vector<int> both;
    for (int b = 0; b < 2; ++b) {
      const int t0 = b ? Ta : 0, T = sep_leafT[b];
      h_leaf_prof[b].resize(T);
      for (int c = 0; c < T; ++c) h_leaf_prof[b][c] = h_sep_prof[t0 + c] - t0;
      both.insert(both.end(), h_leaf_prof[b].begin(), h_leaf_prof[b].end());
    }
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_leaf_prof), both.size() * sizeof(int)));
    SL_HIP(hipMemcpyAsync(d_leaf_prof, both.data(), both.size() * sizeof(int), hipMemcpyHostToDevice, master)); SL_HIP(hipStreamSynchronize(master));
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&sep_ctr2), ((size_t)std::max(Ta, Tb) + 2) * sizeof(int)));
    SL_HIP(hipMemsetAsync(sep_ctr2, 0, ((size_t)std::max(Ta, Tb) + 2) * sizeof(int), master)); SL_HIP(hipStreamSynchronize(master));
    // the top block's product over the leaves' TL column blocks: tile columns of the top block only (the lambda x lambda block is
    // formed after the top block's own steps, over all Ts column blocks), K split so that the launch fills the chip
    std::vector<int> codes;
    for (int jb = 0; jb < Tt; ++jb)
      for (int ib = jb; ib <= nb; ++ib) codes.push_back(ib << 10 | jb);
    n_sep_jobs = (int)codes.size();
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_sep_jobs), codes.size() * sizeof(int)));
    SL_HIP(hipMemcpyAsync(d_sep_jobs, codes.data(), codes.size() * sizeof(int), hipMemcpyHostToDevice, master)); SL_HIP(hipStreamSynchronize(master));
    sep_ks = std::max(1, std::min({8, TL / 4, (1024 + n_sep_jobs - 1) / std::max(n_sep_jobs, 1)}));
    if (getenv("SLIDE_SEP_KS")) sep_ks = std::max(1, std::min(16, atoi(getenv("SLIDE_SEP_KS"))));      // (diagnostic)
    {
      std::vector<int> codes2;
      for (int sy = 0; sy < 2; ++sy)
        for (int c : codes) codes2.push_back(sy << 20 | c);
      if (d_sep_jobs2) { SL_HIP(hipFree(d_sep_jobs2)); d_sep_jobs2 = nullptr; }
      n_sep_jobs2 = (int)codes2.size();
      SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_sep_jobs2), codes2.size() * sizeof(int)));
      SL_HIP(hipMemcpyAsync(d_sep_jobs2, codes2.data(), codes2.size() * sizeof(int), hipMemcpyHostToDevice, master)); SL_HIP(hipStreamSynchronize(master));
    }
    if (sep_ks > 1) {
      const size_t len = 2 * (size_t)(nb + 1) * nb * (sep_ks - 1) * NB * NB * sizeof(double);      // (two systems' partial tiles: launch_border_syrk_jobs with n = 2)
      SL_HIP(hipMalloc(reinterpret_cast<void**>(&sep_scratch), len));
      SL_HIP(hipMemsetAsync(sep_scratch, 0, len, master)); SL_HIP(hipStreamSynchronize(master));      // (the idle half of a right-hand-side tile's partials is never written: it must read as zero)
    }
    // the backward substitution runs over the whole system: its profile must cover the top block's rows under the leaves' columns
    // the other half's partial top block (canonical sums of a whole pass) and which joined graphs hold leaf b
    for (double** p : {&sep_top2, &sep_bord2}) if (*p) { SL_HIP(hipFree(*p)); *p = nullptr; }
    sep_ld2 = (Tt + sep_nl + 1) * NB;
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&sep_top2), (size_t)sep_ld2 * Tt * NB * sizeof(double)));
    SL_HIP(hipMemsetAsync(sep_top2, 0, (size_t)sep_ld2 * Tt * NB * sizeof(double), master));
    if (sep_nl > 0) {
      const size_t nb2 = (size_t)(sep_nl + 1) * NB * sep_nl * NB;
      SL_HIP(hipMalloc(reinterpret_cast<void**>(&sep_bord2), nb2 * sizeof(double)));
      SL_HIP(hipMemsetAsync(sep_bord2, 0, nb2 * sizeof(double), master));
    }
    SL_HIP(hipStreamSynchronize(master));
    sep_mask_b = 0;
    unsigned mask_a = 0;
    for (int i = 0; i < n && i < 8; ++i) {
      const std::vector<int>& mp = graphs[i]->h_sep_map;
      for (int g = 0; g < TL * NB && g < sep_m && g < (int)mp.size(); ++g)
        if (mp[g] >= 0) { if (g < Ta * NB) mask_a |= 1u << i; else sep_mask_b |= 1u << i; }
    }
    if (mask_a & sep_mask_b) { g_last_error = "separator blocks: a robot holds coordinates of both leaves"; return SLIDE_ERR_INVALID; }
    std::vector<int> cover(h_sep_prof);
    for (int c = 0; c < TL; ++c) cover[c] = sep_Ts - 1;
    SL_HIP(hipMemcpyAsync(d_sep_prof, cover.data(), (size_t)sep_Ts * sizeof(int), hipMemcpyHostToDevice, master)); SL_HIP(hipStreamSynchronize(master));
  }
  {
    // which graphs hold coordinates of which tile (landmark tiles, then the lambdas' tiles at virtual index Ts * NB + b)
    std::vector<int> tm(sep_Ts + sep_nl, 0);
    for (int i = 0; i < n && i < 32; ++i) {
      const std::vector<int>& mp = graphs[i]->h_sep_map;
      for (int g = 0; g < sep_m + sep_lam && g < (int)mp.size(); ++g)
        if (mp[g] >= 0) tm[g < sep_m ? g / NB : sep_Ts + (g - sep_m) / NB] |= 1 << i;
    }
    if (d_sep_tmask) { SL_HIP(hipStreamSynchronize(master)); SL_HIP(hipFree(d_sep_tmask)); d_sep_tmask = nullptr; }
    if (n <= 32 && !tm.empty()) {
      SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_sep_tmask), tm.size() * sizeof(int)));
      SL_HIP(hipMemcpyAsync(d_sep_tmask, tm.data(), tm.size() * sizeof(int), hipMemcpyHostToDevice, master)); SL_HIP(hipStreamSynchronize(master));
    }
  }
  if (d_sep_off) { SL_HIP(hipFree(d_sep_off)); d_sep_off = nullptr; }
  SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_sep_off), off.size() * sizeof(int)));
  SL_HIP(hipMemcpyAsync(d_sep_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice, master)); SL_HIP(hipStreamSynchronize(master));
  free_ll_sep_plans();
  if (chol_ll_enabled() && sep_Ts > 0) {
    const CholSystem top = sep_top_system();
    bool ok = true;
    if (chol_ll_mask() & 8) { ll_top = chol_ll_plan_create(&top, 1, nullptr, master); ok = ll_top != nullptr; }
    if (sep_dissected() && (chol_ll_mask() & 4)) {
      CholSystem lv[2];
      sep_leaf_systems(lv);
      ll_leaves = chol_ll_plan_create(lv, 2, nullptr, master);
      ll_leaf_own[0] = chol_ll_plan_create(&lv[0], 1, nullptr, master);
      ll_leaf_own[1] = chol_ll_plan_create(&lv[1], 1, nullptr, master);
      ok = ok && ll_leaves && ll_leaf_own[0] && ll_leaf_own[1];
    }
    if (!ok) { g_last_error = "exact joint step: the left-looking factorisation's tables could not be allocated"; return SLIDE_ERR_HIP; }
  }
  return SLIDE_OK;
}
// One exact joint Gauss-Newton pass of all joined graphs.  part -1: the whole pass; 0: up to this GPU's partial sum of the separator
// system (the caller all-reduces the separator buffer across the GPUs on the pass's stream); 2: the rest.
// Ghost poses of the inter-robot relative-pose factors (slide_graph_set_ghosts): refreshed at the start of every pass — every robot
// packs the current estimates of the ghost slots it owns into its exchange buffer (12 doubles per slot, zeros elsewhere), the buffers
// are summed (part 20 of a cut pass ends here: the caller all-reduces d_bufs[0][0 .. 12 n_gslots)), every robot adopts the sums.
int CholBatch::enqueue_ghost_refresh(double* const* d_bufs, int part) {
  const int ng = hG[0].n_gslots;
  for (int i = 1; i < n; ++i)
    if (hG[i].n_gslots != ng) { g_last_error = "batched pass: the graphs disagree on the ghost slots"; return SLIDE_ERR_INVALID; }
  if (ng <= 0) return SLIDE_OK;
  const bool whole = part < 0;
  if (whole) { launch_ghost_refresh_local(d_Gs, n, ng, master); return SLIDE_OK; }
  if (whole || part == 20) {
    launch_ghost_exchange_batched(d_Gs, n, ng, 0, d_bufs, master);
    launch_sum_bcast(d_bufs, n, 12 * ng, master);
  }
  if (whole || part == 0) {
    if (!whole) launch_bcast(d_bufs, n, 12 * ng, master);
    launch_ghost_exchange_batched(d_Gs, n, ng, 1, d_bufs, master);
  }
  return SLIDE_OK;
}
int CholBatch::enqueue_arrow(double* const* d_bufs, int part, hipEvent_t e0, hipEvent_t e1) {
  const bool whole = part < 0;
  if (part == 20) return enqueue_ghost_refresh(d_bufs, part);
  auto mark = [&](int i) { if (prof_ev[0]) (void)hipEventRecord(prof_ev[i], master); };
  const int* maps[CHOL_BATCH_HOST_MAX];
  double* xloc[CHOL_BATCH_HOST_MAX];
  for (int i = 0; i < n; ++i) { maps[i] = graphs[i]->d_sep_map.d; xloc[i] = graphs[i]->d_xloc.d; }
  const int ld_s = (sep_Ts + sep_nl + 1) * NB;
  const SepLayout Y = sep_layout();
  // (dissected separator) the two leaf blocks as systems of their own: views of sepS, the top block's and the lambdas' rows as border
  CholSystem lv[2];
  const int sTa = sep_leafT[0], sTL = sTa + sep_leafT[1], sTt = sep_Ts - sTL;
  if (sep_dissected()) sep_leaf_systems(lv);
  // a rank that owns a leaf (set_separator_owner; cut passes only): part 1 between the exchanges
  const bool owned = !whole && sep_dissected() && sep_owner >= 0;
  const int own = owned ? sep_owner : 0, own_t0 = own ? sTa : 0, own_T = sep_leafT[own];
  CholSystem top{};      // the top block's Schur complement over a range of the leaves' column blocks (all of them, or the own leaf's)
  top.S = sepS + (owned ? (size_t)own_t0 * NB * ld_s : 0); top.ld = ld_s; top.T = owned ? own_T : sTL; top.b0 = sTL; top.nbr = sTt + sep_nl;
  top.bord = sepS + (size_t)sTL * NB * ld_s + (size_t)sTL * NB; top.ldb = ld_s;
  if (part == 1) {
    if (!owned) return SLIDE_OK;
    if (sep_leader) launch_sep_unpack(Y, master, sTL, sep_Ts + sep_nl);      // this rank's own partial sum of the top block (and the lambdas')
    launch_sep_unpack(Y, master, own_t0, own_t0 + own_T);                      // the leaf, summed over the ranks of this half
    if (ll_leaf_own[own]) launch_chol_ll(ll_leaf_own[own], &lv[own], 1, master);
    else launch_chol_batch(&lv[own], 1, sep_ctr2, master, nullptr, false, 100, (chol_pair_mask() & 4) ? pair_tickets(10) : nullptr);
    if (sep_leader) {
      launch_border_syrk_jobs(&top, 1, d_sep_jobs, n_sep_jobs, 0, master, sep_scratch, sep_ks, sTt);
      if (sep_nl > 0) {
        CholSystem sl{};      // the lambdas' own block: this leaf's part of - W W^T
        sl.S = sepS + (size_t)own_t0 * NB * ld_s; sl.ld = ld_s; sl.T = own_T; sl.b0 = sep_Ts; sl.nbr = sep_nl; sl.bord = sep_bord; sl.ldb = (sep_nl + 1) * NB;
        const int ks = std::max(1, std::min(lam_ks_cap(sep_nl), own_T / 2));
        launch_border_syrk(&sl, 1, master, lam_scratch, ks);
      }
      launch_sep_unpack(Y, master, sTL, sep_Ts + sep_nl, true);               // the top block (with this leaf's Schur complement) back into the exchange buffer
    }
    return SLIDE_OK;
  }
  if (whole || part == 0) {
    launch_status_clear(d_Gs, n, master, sep_status, sep_nl > 0 ? lam_status : nullptr);
    const int rg = enqueue_ghost_refresh(d_bufs, part);
    if (rg != SLIDE_OK) return rg;
    launch_phase0_batched(d_Gs, hG.data(), n, d_bufs, master, false);      // relinearise, linearise, the robots' own per-landmark sums (nothing to pack: no exchange of them)
    launch_phase3_arrow_batched(d_Gs, hG.data(), n, master);         // private landmarks eliminated, reduced pose systems, borders
    {
      int nq[CHOL_BATCH_HOST_MAX];
      for (int i = 0; i < n; ++i) nq[i] = graphs[i]->n_sep_poses;
      launch_sep_extract_batched(d_Gs, hG.data(), n, nq, master);    // nested dissection: the separator poses out of the bands
    }
    if (e0) (void)hipEventRecord(e0, master);
    mark(0);
    if (ll_seg) {                                                    // the segments' factorisations: W^T and y in the border rows
      launch_chol_ll(ll_seg, seg_sys.data(), (int)seg_sys.size(), master);
      last_groups = 1;
      if (e1) (void)hipEventRecord(e1, master);
    } else {
      const int rc = factor_all(e1);
      if (rc != SLIDE_OK) return rc;
    }
    mark(1);
    if (n_syrk_jobs > 0) launch_border_syrk_jobs(sys.data(), n, d_syrk_jobs, n_syrk_jobs, syrk_lds_pad, master);      // border blocks: C_a - W^T W, b_s - W^T y
    else launch_border_syrk(sys.data(), n, master);
    mark(6);
    if (!l2_sys.empty()) {
      // second level: the separator poses' own system (dense, nsep block columns) with the rest of the border as its border
      if (ll_l2) launch_chol_ll(ll_l2, l2_sys.data(), (int)l2_sys.size(), master);
      else launch_chol_batch(l2_sys.data(), (int)l2_sys.size(), d_ctr2, master, nullptr, false, 100, (chol_pair_mask() & 2) ? pair_tickets(8) : nullptr);
      if (n_l2_jobs > 0) launch_border_syrk_jobs(l2_sys.data(), (int)l2_sys.size(), d_l2_jobs, n_l2_jobs, 0, master);
      else launch_border_syrk(l2_sys.data(), (int)l2_sys.size(), master);
    }
    mark(2);
    // a cut pass leaves this GPU's partial sum in the caller's exchange buffer (packed), a whole pass writes the system itself
    static const bool env_canon = !(getenv("SLIDE_SEP_CANONICAL") && getenv("SLIDE_SEP_CANONICAL")[0] == '0');
    const bool canon = whole && sep_dissected() && env_canon && n <= 8;      // (see sep_top2)
    if (canon) launch_sep_gather(hG.data(), n, maps, Y, false, master, d_sep_tmask, sTL * NB, sep_mask_b, sep_top2, sep_ld2, sep_bord2);
    else launch_sep_gather(hG.data(), n, maps, Y, !whole, master, d_sep_tmask);
    mark(3);
  }
  if (whole || part == 2) {
    if (!whole && !owned) launch_sep_unpack(Y, master);
    if (owned) launch_sep_unpack(Y, master, sTL, sep_Ts + sep_nl);            // the top block summed over all ranks (the leaf was unpacked and factored in part 1)
    // landmark part of the separator: the dense step kernels, the lambda coordinates' coupling rows riding as ITS border
    if (sep_dissected()) {
      // the leaves side by side (no robot couples them); the top block's Schur complement; the top block's own steps
      const int TL = sTL, Tt = sTt;
      static const bool env_canon2 = !(getenv("SLIDE_SEP_CANONICAL") && getenv("SLIDE_SEP_CANONICAL")[0] == '0');
      const bool canon2 = whole && env_canon2 && n <= 8;
      if (!owned) {
        if (ll_leaves) launch_chol_ll(ll_leaves, lv, 2, master);
        else launch_chol_batch(lv, 2, sep_ctr2, master, nullptr, false, 100, (chol_pair_mask() & 4) ? pair_tickets(9) : nullptr);
        if (canon2) {
          // each half's partial top block minus ITS leaf's Schur complement, with the split of the column blocks a rank owning that leaf
          // uses (part 1), then the sum of the two — the arithmetic of a two-rank job, bit for bit
          CholSystem tp2[2] = {top, top};
          for (int hh = 0; hh < 2; ++hh) {
            tp2[hh].S = sepS + (size_t)(hh ? sTa : 0) * NB * ld_s; tp2[hh].T = sep_leafT[hh]; tp2[hh].b0 = sTL;
            if (hh) { tp2[hh].bord = sep_top2; tp2[hh].ldb = sep_ld2; }
          }
          CholSystem sl2[2];
          int ks_lam[2] = {1, 1};
          for (int hh = 0; hh < 2 && sep_nl > 0; ++hh) {
            CholSystem sl{};
            sl.S = sepS + (size_t)(hh ? sTa : 0) * NB * ld_s; sl.ld = ld_s; sl.T = sep_leafT[hh]; sl.b0 = sep_Ts; sl.nbr = sep_nl;
            sl.bord = hh ? sep_bord2 : sep_bord; sl.ldb = (sep_nl + 1) * NB;
            ks_lam[hh] = std::max(1, std::min(lam_ks_cap(sep_nl), sep_leafT[hh] / 2));
            sl2[hh] = sl;
          }
          if (sep_scratch && (sep_nl == 0 || (lam_scratch && d_lam_jobs2))) {
            // both leaves in ONE launch each for the top block's columns and for the lambda block, every leaf's column blocks cut the way a
            // rank owning that leaf cuts them, and one reduction each that also adds the second half's result onto the first
            launch_border_syrk_jobs(tp2, 2, d_sep_jobs2, n_sep_jobs2, 0, master, sep_scratch, sep_ks, Tt, nullptr, true);
            if (sep_nl > 0) launch_border_syrk_jobs(sl2, 2, d_lam_jobs2, n_lam_jobs2, 0, master, lam_scratch, 1, -1, ks_lam, true);
          } else {
            for (int hh = 0; hh < 2; ++hh) {
              launch_border_syrk_jobs(&tp2[hh], 1, d_sep_jobs, n_sep_jobs, 0, master, sep_scratch, sep_ks, Tt);
              if (sep_nl > 0) launch_border_syrk(&sl2[hh], 1, master, lam_scratch, ks_lam[hh]);
            }
            launch_sep_top_add(Y, sTL, sep_top2, sep_ld2, sep_bord2, master);
          }
        } else {
          launch_border_syrk_jobs(&top, 1, d_sep_jobs, n_sep_jobs, 0, master, sep_scratch, sep_ks, Tt);
        }
      }
      if (ll_top) {
        const CholSystem ts = sep_top_system();
        launch_chol_ll(ll_top, &ts, 1, master);
      } else if ((chol_pair_mask() & 8) && d_pair_tickets) {
        const CholSystem ts = sep_top_system();
        launch_chol_batch(&ts, 1, sep_ctr, master, nullptr, false, 100, pair_tickets(11));
      } else {
        double* St = sepS + (size_t)TL * NB * ld_s + (size_t)TL * NB;
        for (int k = 0; k < Tt; ++k)
          launch_chol_step(St, ld_s, k, Tt, sep_Ld + (size_t)(TL + k) * NB * NB, sep_Winv + (size_t)(TL + k) * 1024, sep_status, sep_ctr, nullptr, nullptr, master, sep_nl);
        launch_chol_extract_y(St, ld_s, Tt, sep_yv + (size_t)TL * NB, sep_dp + (size_t)TL * NB, sep_status, master, sep_nl);
      }
    } else if (ll_top) {
      const CholSystem ts = sep_top_system();
      launch_chol_ll(ll_top, &ts, 1, master);
    } else if ((chol_pair_mask() & 8) && d_pair_tickets) {
      const CholSystem ts = sep_top_system();
      launch_chol_batch(&ts, 1, sep_ctr, master, nullptr, false, 100, pair_tickets(11));
    } else {
      for (int k = 0; k < sep_Ts; ++k)
        launch_chol_step(sepS, ld_s, k, sep_Ts, sep_Ld + (size_t)k * NB * NB, sep_Winv + (size_t)k * 1024, sep_status, sep_ctr, nullptr,
                         sep_prof_on ? h_sep_prof.data() : nullptr, master, sep_nl);
      launch_chol_extract_y(sepS, ld_s, sep_Ts, sep_yv, sep_dp, sep_status, master, sep_nl);
    }
    if (sep_nl > 0) {
      // the inter-robot relative-pose factors: K22 - L21 L21^T is negative definite; factor its negative, lambda = -M^-1 (r2 - L21 z1),
      // then z1 -= L21^T lambda before the landmark part's backward substitution
      CholSystem ss{};
      ss.S = sepS; ss.ld = ld_s; ss.T = sep_Ts; ss.yv = sep_yv; ss.nbr = sep_nl; ss.bord = sep_bord; ss.ldb = (sep_nl + 1) * NB; ss.bfirst = nullptr;
      CholSystem sr = ss;      // the column blocks whose part of - W W^T is still missing: all, or (a rank that owns a leaf / a canonical whole pass) the top block's
      static const bool env_canon3 = !(getenv("SLIDE_SEP_CANONICAL") && getenv("SLIDE_SEP_CANONICAL")[0] == '0');
      if (owned || (whole && sep_dissected() && env_canon3 && n <= 8)) { sr.S = sepS + (size_t)sTL * NB * ld_s; sr.T = sTt; sr.b0 = sep_Ts; }
      const int ks = std::max(1, std::min(lam_ks_cap(sep_nl), sr.T / 2));      // few border tiles: split the column blocks
      launch_border_syrk(&sr, 1, master, lam_scratch, ks);
      launch_lam_prepare(sep_bord, sep_nl, sep_lam, lamS, master);
      const int ld_l = (sep_nl + 1) * NB;
      for (int k = 0; k < sep_nl; ++k)
        launch_chol_step(lamS, ld_l, k, sep_nl, lam_Ld + (size_t)k * NB * NB, lam_Winv + (size_t)k * 1024, lam_status, lam_ctr, nullptr, nullptr, master);
      launch_chol_extract_y(lamS, ld_l, sep_nl, lam_yv, lam_dp, lam_status, master);
      launch_chol_bwd_all(lamS, ld_l, sep_nl, lam_Ld, lam_Winv, lam_yv, lam_dp, lam_status, nullptr, master);
      const double* xl = lam_dp;
      if (!owned) launch_border_apply(&ss, 1, &xl, master);
      else {      // the own leaf's and the top block's columns only (the other leaf's hold nothing of this pass)
        CholSystem ap2[2] = {ss, ss};
        ap2[0].S = sepS + (size_t)own_t0 * NB * ld_s; ap2[0].T = own_T; ap2[0].b0 = sep_Ts; ap2[0].yv = sep_yv + (size_t)own_t0 * NB;
        ap2[1].S = sepS + (size_t)sTL * NB * ld_s; ap2[1].T = sTt; ap2[1].b0 = sep_Ts; ap2[1].yv = sep_yv + (size_t)sTL * NB;
        const double* xl2[2] = {xl, xl};
        launch_border_apply(ap2, 2, xl2, master);
      }
    }
    if (sep_dissected()) {
      // back through the levels: the top block, y_leaf -= W_top^T x_top, the two leaves side by side (37 hops instead of 60)
      double* St = sepS + (size_t)sTL * NB * ld_s + (size_t)sTL * NB;
      launch_chol_bwd_all(St, ld_s, sTt, sep_Ld + (size_t)sTL * NB * NB, sep_Winv + (size_t)sTL * 1024, sep_yv + (size_t)sTL * NB, sep_dp + (size_t)sTL * NB,
                          sep_status, nullptr, master);
      launch_ints_clear(sep_status + 4, 1, master);                  // (the chain's ticket counter: the leaves' chains draw from it next)
      CholSystem ap[2] = {lv[0], lv[1]};
      ap[0].nbr = ap[1].nbr = sTt;                                    // (the lambdas' part went onto y above, over all column blocks)
      const double* xt[2] = {sep_dp + (size_t)sTL * NB, sep_dp + (size_t)sTL * NB};
      if (!owned) {
        launch_border_apply(ap, 2, xt, master);
        launch_chol_bwd_batch(lv, 2, master);
      } else {
        launch_border_apply(&ap[own], 1, xt, master);
        launch_chol_bwd_batch(&lv[own], 1, master);
      }
    } else {
      launch_chol_bwd_all(sepS, ld_s, sep_Ts, sep_Ld, sep_Winv, sep_yv, sep_dp, sep_status, sep_prof_on ? d_sep_prof : nullptr, master);
    }
    mark(4);
    launch_sep_xloc(n, maps, sep_m, sep_lam, sep_dp, lam_dp, xloc, master);
    if (!l2_sys.empty()) {
      // second level back: y2 -= W2 x (shared landmarks, lambdas), L2^T x_sep = y2, x_sep into the head of the border vector
      const double* x2[CHOL_BATCH_HOST_MAX];
      for (size_t i = 0; i < l2_sys.size(); ++i) x2[i] = xloc[l2_graph[i]] + (size_t)hG[l2_graph[i]].nsep * NB;
      launch_border_apply(l2_sys.data(), (int)l2_sys.size(), x2, master);
      launch_chol_bwd_batch(l2_sys.data(), (int)l2_sys.size(), master);
      launch_ints_clear(l2_sys[0].status + 4, 1, master);            // (the chain's ticket counter: the segments' chains of that graph draw from it next)
      {
        const double* src[CHOL_BATCH_HOST_MAX]; double* dst[CHOL_BATCH_HOST_MAX]; int cnt[CHOL_BATCH_HOST_MAX];
        for (size_t i = 0; i < l2_sys.size(); ++i) { src[i] = l2_sys[i].dp; dst[i] = xloc[l2_graph[i]]; cnt[i] = hG[l2_graph[i]].nsep * NB; }
        launch_copy_pairs(src, dst, cnt, (int)l2_sys.size(), master);
      }
    }
    launch_border_apply(sys.data(), n, xloc, master);                // y -= W x_s
    launch_chol_bwd_batch(seg_sys.data(), (int)seg_sys.size(), master);      // L^T dp = y, all segments side by side
    launch_sep_pose_scatter_batched(d_Gs, hG.data(), n, xloc, master);
    launch_arrow_finish_batched(d_Gs, hG.data(), n, sep_dp, d_sep_off, master);
    launch_status_gather(d_Gs, n, d_status_all, master, sep_status, sep_nl > 0 ? lam_status : nullptr);      // (the separator's not-SPD / chain flags are reported with graph 0's)
    mark(5);
  }
  return SLIDE_OK;
}
// One exact joint pass issued directly (no graph) with HIP events on the pass's stream between its stages: out6 = ms of {assembly
// (linearisation .. borders), the bands' factorisations, the border products, the separator gather, the separator's factorisation +
// solve, the back-substitutions}, *n_sep_steps = block columns of the separator system
int CholBatch::profile_arrow(double* const* d_bufs, double* out6, int* n_sep_steps) {
  std::lock_guard<std::mutex> pl(pass_mtx);
  bool same = false;
  int rc = begin_pass(d_bufs, &same);
  if (rc != SLIDE_OK) return rc;
  if (!(arrow && hG[0].n_slots > 0)) { g_last_error = "profile_arrow: the batch does not run exact joint passes"; return SLIDE_ERR_INVALID; }
  hipEvent_t start = nullptr;
  SL_HIP(hipEventCreate(&start));
  for (auto& e : prof_ev) SL_HIP(hipEventCreate(&e));
  (void)hipEventRecord(start, master);
  rc = enqueue_arrow(d_bufs, -1, nullptr, nullptr);
  const hipError_t es = hipStreamSynchronize(master);
  if (rc == SLIDE_OK && es == hipSuccess) {
    // stages: assembly | segments' steps | robots' border product | second level of the bands (steps + its product) | gather |
    // separator | back-substitution; the second level counts as band factorisation
    hipEvent_t seq[8] = {start, prof_ev[0], prof_ev[1], prof_ev[6], prof_ev[2], prof_ev[3], prof_ev[4], prof_ev[5]};
    float d[7] = {};
    for (int i = 0; i < 7; ++i)
      if (hipEventElapsedTime(&d[i], seq[i], seq[i + 1]) != hipSuccess) rc = SLIDE_ERR_HIP;
    out6[0] = d[0]; out6[1] = d[1] + d[3]; out6[2] = d[2]; out6[3] = d[4]; out6[4] = d[5]; out6[5] = d[6];
    if (n_sep_steps) *n_sep_steps = sep_Ts;
  } else if (rc == SLIDE_OK) rc = SLIDE_ERR_HIP;
  (void)hipEventDestroy(start);
  for (auto& e : prof_ev) { (void)hipEventDestroy(e); e = nullptr; }
  return rc == SLIDE_OK ? end_pass() : rc;
}
// the launches of one pass of all joined graphs (captured by capture_pass, or issued directly by profile_pass with events e0 / e1
// around the batched step kernels)
// part: -1 = the whole pass (every robot of the job is in this batch); 0 / 1 / 2 = the pass cut at its two exchanges, for a job that
// spans GPUs: after part 0 every local buffer holds the LOCAL sum of the 54-doubles-per-slot blocks and the caller all-reduces
// bufs[0] across the ranks on the batch's stream (RCCL); part 1 hands bufs[0] back to the other local buffers, runs up to the local
// sum of the 9-doubles-per-slot t_l; the caller all-reduces bufs[0] again; part 2 hands it back and finishes the pass.
int CholBatch::enqueue_pass(double* const* d_bufs, hipEvent_t e0, hipEvent_t e1, int part) {
  static const bool batch_p3 = !(getenv("SLIDE_BATCH_PHASE3") && getenv("SLIDE_BATCH_PHASE3")[0] == '0');     // diagnostic
  int rc = SLIDE_OK;
  // every robot's stream continues behind the batch's stream, runs `phase`, and is joined back
  auto each = [&](int phase) {
    if (hipEventRecord(ev_fork, master) != hipSuccess) { rc = SLIDE_ERR_HIP; return; }
    for (int i = 0; i < n && rc == SLIDE_OK; ++i) {
      HostGraph* g = graphs[i];
      if (hipStreamWaitEvent(g->stream, ev_fork, 0) != hipSuccess) { rc = SLIDE_ERR_HIP; break; }
      rc = g->enqueue_phase(phase, d_bufs[i]);
      if (rc == SLIDE_OK && (hipEventRecord(ev_in[i], g->stream) != hipSuccess || hipStreamWaitEvent(master, ev_in[i], 0) != hipSuccess))
        rc = SLIDE_ERR_HIP;
    }
  };
  const int n_slots = graphs[0]->G.n_slots;
  if (arrow && n_slots > 0) return enqueue_arrow(d_bufs, part, e0, e1);
  const bool whole = part < 0;
  const bool joint = pcg_iters > 0 && n_slots > 0;       // PCG over the robots' coupled systems instead of the plain block solves
  if (part == 20) return enqueue_ghost_refresh(d_bufs, part);
  if (whole || part == 0) {
    launch_status_clear(d_Gs, n, master);      // (a kernel node: a captured hipMemsetAsync did not clear on replay, DESIGN §4 finding 6)
    if ((rc = enqueue_ghost_refresh(d_bufs, part)) != SLIDE_OK) return rc;
    if (batch_p3) launch_phase0_batched(d_Gs, hG.data(), n, d_bufs, master);      // (all robots in one launch sequence: no fork / join)
    else each(0);
    if (rc == SLIDE_OK) launch_sum_bcast(d_bufs, n, 54 * n_slots, master);
  }
  if (whole || part == 1) {
    if (rc == SLIDE_OK && !whole) launch_bcast(d_bufs, n, 54 * n_slots, master);
    if (rc == SLIDE_OK) {
      if (batch_p3) launch_phase3_batched(d_Gs, hG.data(), n, d_bufs, master);      // five launches for all robots (blockIdx.z = robot)
      else each(3);
    }
    if (rc == SLIDE_OK && joint && !batch_p3) rc = save_systems();      // (the batched assembly wrote S0 along with S)
    if (rc == SLIDE_OK && e0) (void)hipEventRecord(e0, master);
    if (rc == SLIDE_OK) rc = factor_all(e1);
    if (rc == SLIDE_OK && joint) rc = enqueue_pcg_head(d_bufs);
  }
  if (joint) {
    // the PCG iterations: whole pass = all of them inline; cut pass = part 10 (after the t_l exchange), part 11 / 12 (after the
    // exchange of the two dot products; 12 = the last iteration)
    for (int it = 0; it < pcg_iters && rc == SLIDE_OK; ++it) {
      const bool last = it == pcg_iters - 1;
      if (whole || (part == 10 && it == 0)) {
        if (!whole) launch_bcast(d_bufs, n, 9 * n_slots, master);
        rc = enqueue_pcg_mid(d_bufs, whole);
      }
      if (rc == SLIDE_OK && (whole || (part == 11 && it == 0 && !last) || (part == 12 && last))) {
        if (!whole) launch_bcast(d_bufs, n, 2, master);
        rc = enqueue_pcg_tail(d_bufs, last, whole);
      }
    }
  }
  if (whole || part == 1 || part == 12) {
    const bool here = whole || (part == 1 && !joint) || (part == 12 && joint);
    if (rc == SLIDE_OK && here) {
      if (batch_p3) launch_phase4_batched(d_Gs, hG.data(), n, d_bufs, master);
      else each(4);
    }
    if (rc == SLIDE_OK && here) launch_sum_bcast(d_bufs, n, 9 * n_slots, master);
  }
  if (whole || part == 2) {
    if (rc == SLIDE_OK && !whole) launch_bcast(d_bufs, n, 9 * n_slots, master);
    if (rc == SLIDE_OK) {
      if (batch_p3) launch_phase2_batched(d_Gs, hG.data(), n, d_bufs, master);
      else each(2);
    }
    if (rc == SLIDE_OK) launch_status_gather(d_Gs, n, d_status_all, master);
  }
  return rc;
}
// One capture at a time in the process: batches driven by different host threads (the ranks-as-threads rehearsal) capture their passes
// thread-locally, but concurrent captures were seen to fail on this stack ("stream capture failed"); a capture takes milliseconds, once.
static std::mutex g_capture_mtx;
int CholBatch::capture_pass(double* const* d_bufs, int part, hipGraphExec_t* exec) {
  std::lock_guard<std::mutex> cap(g_capture_mtx);
  if (*exec) { (void)hipGraphExecDestroy(*exec); *exec = nullptr; }
  hipGraph_t graph = nullptr;
  SL_HIP(hipStreamBeginCapture(master, hipStreamCaptureModeThreadLocal));
  g_last_error.clear();
  const int rc = enqueue_pass(d_bufs, nullptr, nullptr, part);
  std::string inner = g_last_error;      // (what the first failing call inside the captured region said)
  {
    const hipError_t le = hipPeekAtLastError();      // (launches are not checked one by one)
    if (le != hipSuccess) inner += std::string(inner.empty() ? "" : "; ") + "last launch error: " + hipGetErrorString(le);
  }
  const hipError_t e = hipStreamEndCapture(master, &graph);
  if (rc != SLIDE_OK || e != hipSuccess || graph == nullptr) {
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    g_last_error = std::string("batched pass: stream capture failed (") + hipGetErrorString(e) + ", enqueue rc " + std::to_string(rc) + (inner.empty() ? "" : "; " + inner) + ")";
    return rc != SLIDE_OK ? rc : SLIDE_ERR_HIP;
  }
  const hipError_t ei = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess) { *exec = nullptr; (void)hipGetLastError(); g_last_error = "batched pass: graph instantiation failed"; return SLIDE_ERR_HIP; }
  return SLIDE_OK;
}
// One pass issued directly (no graph), with HIP events on the batch's stream around the batched step kernels: their total device
// time and launch count (the bench's roofline of k_chol_step_batched).
int CholBatch::profile_pass(double* const* d_bufs, double* ms_steps, int* n_launches) {
  std::lock_guard<std::mutex> pl(pass_mtx);
  {
    std::lock_guard<std::mutex> lk(mtx);
    for (int i = 0; i < n; ++i)
      if (!graphs[i] || !d_bufs[i]) { g_last_error = "batched pass: a slot of the batch is empty"; return SLIDE_ERR_INVALID; }
  }
  if (!master) {
    SL_HIP(hipStreamCreateWithFlags(&master, hipStreamNonBlocking));
    SL_HIP(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming));
  }
  for (int i = 0; i < n; ++i) {
    HostGraph* g = graphs[i];
    std::lock_guard<std::mutex> gl(g->mtx);
    g->pred_valid = false; g->status_clean = false; g->cache_pose = -1;      // (a batched pass rewrites the status words, deltas and estimates)
    int rc = g->merge_pending();
    if (rc == SLIDE_OK) rc = g->upload_new();
    if (rc != SLIDE_OK) return rc;
    g->G.relin_thr = 0.0;
    SL_HIP(hipStreamSynchronize(g->stream));
  }
  int rc = prepare_pass();
  if (rc != SLIDE_OK) return rc;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  SL_HIP(hipEventCreate(&e0));
  SL_HIP(hipEventCreate(&e1));
  rc = enqueue_pass(d_bufs, e0, e1, -1);
  const hipError_t es = hipStreamSynchronize(master);
  for (int i = 0; i < n; ++i) (void)hipStreamSynchronize(graphs[i]->stream);
  float ms = 0.f;
  if (rc == SLIDE_OK && es == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
    if (ms_steps) *ms_steps = ms;
    int Tmax = 0;
    const bool exact = arrow && !hG.empty() && hG[0].n_slots > 0;      // (exact passes step over the bands' segments)
    for (const CholSystem& c : (exact ? seg_sys : sys)) Tmax = c.T > Tmax ? c.T : Tmax;
    if (n_launches) *n_launches = Tmax * last_groups;      // step launches of the pass: the longest system's per launch sequence
  } else if (rc == SLIDE_OK) rc = SLIDE_ERR_HIP;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

// Brings every joined graph up to date, clears the status words and decides whether the captured launch sequences still fit
// (same device views, same exchange buffers, same set of graphs).  Called at the start of a pass (whole, or part 0).
int CholBatch::begin_pass(double* const* d_bufs, bool* same) {
  thread_capture_mode_local();
  bool dirty;
  {
    std::lock_guard<std::mutex> lk(mtx);
    for (int i = 0; i < n; ++i)
      if (!graphs[i] || !d_bufs[i]) { g_last_error = "batched pass: a slot of the batch is empty"; return SLIDE_ERR_INVALID; }
    dirty = pass_dirty;
    pass_dirty = false;
  }
  if (!master) {
    SL_HIP(hipStreamCreateWithFlags(&master, hipStreamNonBlocking));
    SL_HIP(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming));
  }
  *same = !dirty && (int)pass_G.size() == n;
  for (int i = 0; i < n; ++i) {
    HostGraph* g = graphs[i];
    std::lock_guard<std::mutex> gl(g->mtx);
    g->pred_valid = false; g->status_clean = false; g->cache_pose = -1;      // (a batched pass rewrites the status words, deltas and estimates)
    int rc = g->merge_pending();
    if (rc == SLIDE_OK) rc = g->upload_new();
    if (rc != SLIDE_OK) return rc;
    g->G.relin_thr = 0.0;
    if (g->G.n_slots != graphs[0]->G.n_slots) { g_last_error = "batched pass: the graphs disagree on the shared slots"; return SLIDE_ERR_INVALID; }
    SL_HIP(hipStreamSynchronize(g->stream));                       // (uploads of a changed graph; idle otherwise)
    *same = *same && std::memcmp(&pass_G[i], &g->G, sizeof(GraphDev)) == 0 && pass_bufs[i] == d_bufs[i];
  }
  if (!*same) {
    // every captured sequence is stale: drop them, refresh the device-side tables once
    if (pass_exec) { (void)hipGraphExecDestroy(pass_exec); pass_exec = nullptr; }
    for (auto& e : part_exec) if (e) { (void)hipGraphExecDestroy(e); e = nullptr; }
    const int rc = prepare_pass();
    if (rc != SLIDE_OK) return rc;
    pass_G.resize(n);
    pass_bufs.assign(d_bufs, d_bufs + n);
    for (int i = 0; i < n; ++i) pass_G[i] = graphs[i]->G;
  }
  return SLIDE_OK;
}
int CholBatch::end_pass() {
  int st[CHOL_BATCH_HOST_MAX][8];
  SL_HIP(hipMemcpyAsync(st, d_status_all, (size_t)n * 8 * sizeof(int), hipMemcpyDeviceToHost, master));      // (gathered by the pass's last node)
  SL_HIP(hipStreamSynchronize(master));
  SL_HIP(hipGetLastError());
  for (int i = 0; i < n; ++i) {
    const int rc = decode_status(st[i]);
    if (rc != SLIDE_OK) return rc;
  }
  return SLIDE_OK;
}

// The batched factor + solve of all joined systems in several launch sequences on as many streams (two on wide profiles, four on
// narrow ones; SLIDE_CHOL_GROUPS overrides): the systems are independent, so nothing synchronises the sequences between the fork and
// the join, and the launch tails / serial chains of one sequence overlap with the work of the others.  `after` (measurement): recorded
// behind everything (the join included).
int CholBatch::factor_all(hipEvent_t after) {
  static const int env_groups = getenv("SLIDE_CHOL_GROUPS") ? atoi(getenv("SLIDE_CHOL_GROUPS")) : 0;
  // wide profiles: two sequences (the floods share the CUs; four measured slower); narrow profiles (every launch is a handful of
  // chain-bound workgroups): four sequences of two systems — more launches in flight hide each other's gaps and prologues (eight
  // robots: 3.67 ms per pass with two sequences, 3.50 with four, 4.29 with eight: the cross-stream joins then cost more than they hide)
  const bool exact = arrow && hG[0].n_slots > 0;      // exact joint step: the systems are the graphs' segments; the backward substitutions wait for the separator's solution
  const std::vector<CholSystem>& S_ = exact ? seg_sys : sys;
  const int ns = (int)S_.size();
  bool narrow = true;
  for (int i = 0; i < n; ++i) narrow = narrow && hG[i].schur_split == 1;
  int groups = env_groups > 0 ? env_groups : (narrow ? 4 : 2);
  // cut bands: three times the systems, a third of the launches each — few sequences, but two: one alone leaves every dispatch gap
  // open (24 systems: 0.60 ms in one sequence, 0.53 in two of twelve, 0.54 in three of eight, 0.65 in four of six, 0.85 in six of four)
  if (env_groups <= 0 && exact && ns > n) groups = std::max(2, (ns + 11) / 12);
  if (groups > (env_groups > 0 ? ns : ns / 2)) groups = env_groups > 0 ? ns : ns / 2;      // (at least two systems per sequence by default)
  while (groups > 0 && (ns + groups - 1) / groups > CHOL_STEP_BATCH_MAX) ++groups;          // (the step kernel's argument block holds that many systems)
  if (groups > 8) groups = 8;
  last_groups = groups < 1 ? 1 : groups;
  const bool solve = !exact;
  if (groups < 2) {
    launch_chol_batch(S_.data(), ns, d_ctr, master, nullptr, solve, 100, (exact && (chol_pair_mask() & 1)) ? pair_tickets(0) : nullptr);
    if (after) SL_HIP(hipEventRecord(after, master));
    return SLIDE_OK;
  }
  if (!ev_aux0) SL_HIP(hipEventCreateWithFlags(&ev_aux0, hipEventDisableTiming));
  SL_HIP(hipEventRecord(ev_aux0, master));
  const int per = (ns + groups - 1) / groups;
  for (int g = 0; g < groups; ++g) {
    const int lo = g * per, hi = std::min(ns, lo + per);
    if (lo >= hi) break;
    hipStream_t st = master;
    if (g > 0) {
      if (!aux[g]) {
        SL_HIP(hipStreamCreateWithFlags(&aux[g], hipStreamNonBlocking));
        SL_HIP(hipEventCreateWithFlags(&ev_aux1[g], hipEventDisableTiming));
      }
      st = aux[g];
      SL_HIP(hipStreamWaitEvent(st, ev_aux0, 0));
    }
    launch_chol_batch(S_.data() + lo, hi - lo, d_ctr + (size_t)g * (ctr_cap / CHOL_BATCH_HOST_MAX), st, nullptr, solve, std::max(25, 100 / groups),
                      (exact && (chol_pair_mask() & 1)) ? pair_tickets(g) : nullptr);
    if (g > 0) {
      SL_HIP(hipEventRecord(ev_aux1[g], st));
      SL_HIP(hipStreamWaitEvent(master, ev_aux1[g], 0));
    }
  }
  if (after) SL_HIP(hipEventRecord(after, master));
  return SLIDE_OK;
}

void CholBatch::set_pcg(int iters, double tol) {
  std::lock_guard<std::mutex> pl(pass_mtx);
  std::vector<HostGraph*> gs;
  {
    std::lock_guard<std::mutex> lk(mtx);
    pcg_iters = iters < 0 ? 0 : iters;
    pcg_tol = tol > 0.0 ? tol : 0.0;
    pass_dirty = true;
    gs.assign(graphs.begin(), graphs.end());
  }
  for (HostGraph* g : gs)              // the joined graphs (re)build their device view: the joint-solve buffers are allocated on demand
    if (g) { std::lock_guard<std::mutex> gl(g->mtx); g->topo_dirty = true; }
}
// S -> S0 for every joined graph (the factorisation works in place; the joint solve multiplies with the original blocks)
int CholBatch::save_systems() {
  for (int i = 0; i < n; ++i) {
    const GraphDev& G = hG[i];
    const size_t bytes = (size_t)G.ld * G.T * NB * sizeof(double);
    if (bytes) SL_HIP(hipMemcpyAsync(G.S0, G.S, bytes, hipMemcpyDeviceToDevice, master));
  }
  return SLIDE_OK;
}
int CholBatch::enqueue_pcg_head(double* const* d_bufs) {
  launch_pcg_init(d_Gs, hG.data(), n, master);
  launch_pcg_tl(d_Gs, hG.data(), n, d_bufs, PCG_VEC_U, master);
  launch_sum_bcast(d_bufs, n, 9 * hG[0].n_slots, master);
  return SLIDE_OK;
}
// whole = a whole-pass graph (every robot of the job on this GPU): k_pcg_scalars sums the n robots' partial dot products itself
// (k_sum_bcast's order) instead of a sum node before it.  (The same for t_l inside k_pcg_cross was slower than the sum node: 35
// against 25 + 5 us — scattered reads of eight buffers.  Running the products w = S0 u on a side stream beside t_l and its exchange
// was slower too: the cross-stream join costs ~12 us, more than the 22 us product hides.)
int CholBatch::enqueue_pcg_mid(double* const* d_bufs, bool whole) {
  launch_pcg_matvec_dots(d_Gs, hG.data(), n, d_bufs, 1, master);
  if (!whole) launch_sum_bcast(d_bufs, n, 2, master);
  return SLIDE_OK;
}
int CholBatch::enqueue_pcg_tail(double* const* d_bufs, bool last, bool whole) {
  launch_pcg_update(d_Gs, hG.data(), n, d_bufs, whole ? n : 1, master);
  if (last) {
    launch_pcg_finish(d_Gs, hG.data(), n, master);
    return SLIDE_OK;
  }
  const double* in[CHOL_BATCH_HOST_MAX];
  double* out[CHOL_BATCH_HOST_MAX];
  double* nxt[CHOL_BATCH_HOST_MAX];
  for (int i = 0; i < n; ++i) {
    in[i] = hG[i].pcg + (size_t)PCG_VEC_R * hG[i].T * NB; out[i] = hG[i].pcg + (size_t)PCG_VEC_Y * hG[i].T * NB;
    nxt[i] = hG[i].pcg + (size_t)PCG_VEC_U * hG[i].T * NB;
  }
  launch_chain_batch(sys.data(), n, in, out, true, true, true, nxt, master);       // (y prepared by k_pcg_update; prepares u for the backward chain)
  for (int i = 0; i < n; ++i) { in[i] = out[i]; out[i] = nxt[i]; }
  launch_chain_batch(sys.data(), n, in, out, false, true, true, nullptr, master);
  launch_pcg_tl(d_Gs, hG.data(), n, d_bufs, PCG_VEC_U, master);
  launch_sum_bcast(d_bufs, n, 9 * hG[0].n_slots, master);
  return SLIDE_OK;
}

hipStream_t CholBatch::pass_stream() {
  std::lock_guard<std::mutex> pl(pass_mtx);
  if (!master) {
    if (hipStreamCreateWithFlags(&master, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&ev_out, hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  return master;
}

int CholBatch::pass_all(double* const* d_bufs) {
  std::lock_guard<std::mutex> pl(pass_mtx);
  bool same = false;
  int rc = begin_pass(d_bufs, &same);
  if (rc != SLIDE_OK) return rc;
  if (!pass_exec && (rc = capture_pass(d_bufs, -1, &pass_exec)) != SLIDE_OK) return rc;
  SL_HIP(hipGraphLaunch(pass_exec, master));
  last_part = -1;
  return end_pass();
}

// The pass in three stream-ordered parts for a job that spans GPUs (see enqueue_pass): parts 0 and 1 return without a host
// synchronisation — the caller's collective goes onto stream() behind them — part 2 ends with the one synchronisation of the pass.
int CholBatch::pass_part(double* const* d_bufs, int part) {
  thread_capture_mode_local();
  const int slot = part >= 0 && part <= 2 ? part : (part >= 10 && part <= 12 ? part - 7 : (part == 20 ? 6 : -1));
  if (slot < 0) return SLIDE_ERR_INVALID;
  std::lock_guard<std::mutex> pl(pass_mtx);
  int rc;
  if (part == 20) {           // (the ghost refresh opens a pass: it brings the graphs up to date like part 0 does; part 0 then finds them unchanged)
    bool same = false;
    if ((rc = begin_pass(d_bufs, &same)) != SLIDE_OK) return rc;
  } else if (part == 0) {
    bool same = false;
    if ((rc = begin_pass(d_bufs, &same)) != SLIDE_OK) return rc;
  } else if ((int)pass_G.size() != n || !master) {
    g_last_error = "batched pass: part 0 has not run";
    return SLIDE_ERR_INVALID;
  }
  // The parts of a cut pass come in ONE order (a part that arrives out of order used to be accepted, or skipped, silently):
  //   [20] 0 2                                   exact joint step      (20: only with ghost poses, then mandatory)
  //   [20] 0 1 { 10 11 } x (iterations - 1) 10 12 2    PCG            [20] 0 1 2    block-Jacobi
  const bool exact = arrow && hG[0].n_slots > 0, joint = !exact && pcg_iters > 0 && hG[0].n_slots > 0, ghosts = hG[0].n_gslots > 0;
  bool ok;
  switch (part) {
    case 20: ok = (last_part == -1 || last_part == 2) && ghosts; break;
    case 0: ok = ghosts ? last_part == 20 : (last_part == -1 || last_part == 2); break;
    case 1: ok = (!exact || (sep_dissected() && sep_owner >= 0)) && last_part == 0; break;
    case 10: ok = joint && (last_part == 1 || last_part == 11); break;
    case 11: case 12: ok = joint && last_part == 10; break;
    default: ok = exact ? last_part == ((sep_dissected() && sep_owner >= 0) ? 1 : 0) : (joint ? last_part == 12 : last_part == 1); break;      // part 2
  }
  if (part == 20 && !ghosts) return SLIDE_OK;                                             // (no ghost poses: nothing to refresh)
  if (!exact && !joint && part >= 10 && part <= 12) return SLIDE_OK;                      // (no joint solve: nothing between parts 1 and 2)
  if (!ok) {
    g_last_error = "batched pass: part " + std::to_string(part) + " does not follow part " + std::to_string(last_part) +
                   (exact ? " (exact joint step: [20] 0 2)" : joint ? " (PCG: [20] 0 1 {10 11} 10 12 2)" : " ([20] 0 1 2)") +
                   (ghosts ? "; the graphs hold ghost poses: a pass opens with part 20" : "");
    last_part = -1;
    return SLIDE_ERR_INVALID;
  }
  if (exact && !sep_x) { g_last_error = "exact joint step: a cut pass needs the caller's separator exchange buffer (slide_chol_batch_set_exact_joint)"; return SLIDE_ERR_INVALID; }
  // SLIDE_PASS_DIRECT=1: the part's launches issued directly instead of a replayed hipGraph (several host threads driving batches of one
  // process side by side — the ranks-as-threads rehearsal: stream captures of concurrent threads fail on this stack)
  static const bool direct = getenv("SLIDE_PASS_DIRECT") && getenv("SLIDE_PASS_DIRECT")[0] == '1';
  if (direct) {
    if ((rc = enqueue_pass(d_bufs, nullptr, nullptr, part)) != SLIDE_OK) return rc;
  } else {
    if (!part_exec[slot] && (rc = capture_pass(d_bufs, part, &part_exec[slot])) != SLIDE_OK) return rc;
    SL_HIP(hipGraphLaunch(part_exec[slot], master));
  }
  last_part = part;
  return part == 2 ? end_pass() : SLIDE_OK;
}

int HostGraph::factor_and_solve(hipStream_t s) {
  if (batch) return batch->factor_solve(batch_slot, G, s);
  for (int k = 0; k < G.T; ++k)
    launch_chol_step(G.S, G.ld, k, G.T, G.Ld + (size_t)k * NB * NB, G.Winv + (size_t)k * 1024, G.status, G.chol_ctr,
                     (pcg_iters > 0 && G.n_slots > 0) ? d_L32.d : nullptr, h_prof.data(), s);
  launch_chol_extract_y(G.S, G.ld, G.T, G.yv, G.dp, G.status, s);
  launch_chol_solve_bwd(CholSystem{G.S, G.ld, G.T, G.Ld, G.Winv, G.yv, G.dp, G.status, (pcg_iters > 0 && G.n_slots > 0) ? d_L32.d : nullptr,
                                   h_prof.data(), G.prof, G.first, d_ctab.d}, s);
  return SLIDE_OK;
}

thread_local UploadBatch* UploadBatch::current = nullptr;
UploadBatch::~UploadBatch() {
  if (current == this) current = nullptr;
  if (ev) (void)hipEventDestroy(ev);
  if (h_pin) (void)hipHostFree(h_pin);
  if (d_stage) (void)hipFree(d_stage);
}
int UploadBatch::begin() {
  if (!ev) SL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  if (in_flight) { SL_HIP(hipEventSynchronize(ev)); in_flight = false; }
  segs.clear();
  used = 0;
  current = this;
  return SLIDE_OK;
}
int UploadBatch::reserve(size_t need) {
  if (need <= h_cap) return SLIDE_OK;
  size_t nc = h_cap ? h_cap : (1u << 16);
  while (nc < need) nc *= 2;
  unsigned char* np = nullptr;
  SL_HIP(hipHostMalloc(reinterpret_cast<void**>(&np), nc, hipHostMallocDefault));
  if (used) std::memcpy(np, h_pin, used);
  if (h_pin) SL_HIP(hipHostFree(h_pin));
  h_pin = np;
  h_cap = nc;
  return SLIDE_OK;
}
int UploadBatch::add(void* dst, const void* src, size_t bytes) {
  const size_t off = (used + 15) & ~(size_t)15;
  if (reserve(off + bytes) != SLIDE_OK) return SLIDE_ERR_HIP;
  std::memcpy(h_pin + off, src, bytes);
  segs.push_back({(unsigned long long)(uintptr_t)dst, (unsigned)off, (unsigned)bytes});
  used = off + bytes;
  return SLIDE_OK;
}
int UploadBatch::flush(hipStream_t s) {
  current = nullptr;
  if (segs.empty()) return SLIDE_OK;
  const size_t desc_off = (used + 15) & ~(size_t)15;
  const size_t total = desc_off + segs.size() * sizeof(Seg);
  if (reserve(total) != SLIDE_OK) return SLIDE_ERR_HIP;          // the descriptors ride at the end of the same buffer
  std::memcpy(h_pin + desc_off, segs.data(), segs.size() * sizeof(Seg));
  if (total > d_cap) {
    size_t nc = d_cap ? d_cap : (1u << 16);
    while (nc < total) nc *= 2;
    if (d_stage) { SL_HIP(hipStreamSynchronize(s)); SL_HIP(hipFree(d_stage)); d_stage = nullptr; }
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_stage), nc));
    d_cap = nc;
  }
  SL_HIP(hipMemcpyAsync(d_stage, h_pin, total, hipMemcpyHostToDevice, s));
  SL_HIP(hipEventRecord(ev, s));
  in_flight = true;
  launch_scatter(d_stage, (unsigned)desc_off, (int)segs.size(), s);
  segs.clear();
  used = 0;
  return SLIDE_OK;
}

DownloadBatch::~DownloadBatch() {
  if (h_pin) (void)hipHostFree(h_pin);
  if (d_stage) (void)hipFree(d_stage);
}
void DownloadBatch::add(void* host, const void* dev, size_t bytes) {
  if (bytes == 0) return;
  const size_t off = (used + 15) & ~(size_t)15;
  segs.push_back({(unsigned long long)(uintptr_t)dev, (unsigned)off, (unsigned)bytes});
  host_dst.push_back(host);
  used = off + bytes;
}
int DownloadBatch::run(hipStream_t s) {
  if (segs.empty()) { SL_HIP(hipStreamSynchronize(s)); return SLIDE_OK; }
  const size_t desc_off = (used + 15) & ~(size_t)15;
  const size_t total = desc_off + segs.size() * sizeof(Seg);
  if (total > h_cap) {
    size_t nc = h_cap ? h_cap : (1u << 16);
    while (nc < total) nc *= 2;
    if (h_pin) SL_HIP(hipHostFree(h_pin));
    SL_HIP(hipHostMalloc(reinterpret_cast<void**>(&h_pin), nc, hipHostMallocDefault));
    h_cap = nc;
  }
  if (total > d_cap) {
    if (d_stage) { SL_HIP(hipStreamSynchronize(s)); SL_HIP(hipFree(d_stage)); d_stage = nullptr; }
    SL_HIP(hipMalloc(reinterpret_cast<void**>(&d_stage), h_cap));
    d_cap = h_cap;
  }
  if (segs.size() <= 16) {
    launch_gather_args(d_stage, segs.data(), (int)segs.size(), s);      // (the per-frame read-backs: the descriptors travel as kernel arguments)
  } else {
    std::memcpy(h_pin + desc_off, segs.data(), segs.size() * sizeof(Seg));
    SL_HIP(hipMemcpyAsync(d_stage + desc_off, h_pin + desc_off, segs.size() * sizeof(Seg), hipMemcpyHostToDevice, s));
    launch_gather(d_stage, (unsigned)desc_off, (int)segs.size(), s);
  }
  SL_HIP(hipMemcpyAsync(h_pin, d_stage, used, hipMemcpyDeviceToHost, s));
  SL_HIP(hipStreamSynchronize(s));
  for (size_t i = 0; i < segs.size(); ++i) std::memcpy(host_dst[i], h_pin + segs[i].off, segs[i].bytes);
  segs.clear();
  host_dst.clear();
  used = 0;
  return SLIDE_OK;
}

template <class T>
static int up_tail(DevArr<T>& d, const std::vector<T>& h, size_t old_n, size_t per, hipStream_t s) {
  const size_t n = h.size();
  if (d.ensure(std::max<size_t>(n, 1), old_n * per, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  return d.upload(h.data() + old_n * per, old_n * per, n - old_n * per, s);
}
// lists -> device CSR, from the lowest list that changed since the last upload on (HostGraph::CsrMirror)
static int up_csr(CsrMirror& M, DevArr<int>& dptr, DevArr<int>& dval, const std::vector<std::vector<int>>& lists, hipStream_t s) {
  const int n = (int)lists.size();
  int from = std::min(M.dirty_from, (int)M.ptr.size() - 1);
  from = std::max(0, std::min(from, n));
  M.ptr.resize((size_t)from + 1);
  M.val.resize((size_t)M.ptr[from]);
  M.off0 = M.val.size();
  for (int i = from; i < n; ++i) {
    M.val.insert(M.val.end(), lists[i].begin(), lists[i].end());
    M.ptr.push_back((int)M.val.size());
  }
  M.dirty_from = n;
  if (dptr.ensure(M.ptr.size(), (size_t)from, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (dval.ensure(std::max<size_t>(M.val.size(), 1), M.off0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (dptr.upload(M.ptr.data() + from, (size_t)from, M.ptr.size() - (size_t)from, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  return dval.upload(M.val.data() + M.off0, M.off0, M.val.size() - M.off0, s);
}

int HostGraph::upload_new() {
  thread_capture_mode_local();
  hipStream_t s = stream;
  if (!topo_dirty && uploaded_once) return SLIDE_OK;      // nothing was added since the last upload: values and topology are resident
  if (ub.begin() != SLIDE_OK) return SLIDE_ERR_HIP;
  struct BatchGuard { ~BatchGuard() { UploadBatch::current = nullptr; } } batch_guard;      // an error return abandons the batch
  const size_t Pn = h_pose_val.size() / 12, Ln = h_lm_type.size();
  const size_t npr = h_pr_pose.size(), nbt = h_bt_i.size(), nlf = h_lf_type.size();
#define UP(dev, host, oldn, per) \
  if (up_tail(dev, host, oldn, per, s) != SLIDE_OK) return SLIDE_ERR_HIP
  UP(d_pose_val, h_pose_val, up_P, 12);
  if (d_pose_delta.ensure(std::max<size_t>(6 * Pn, 1), 6 * up_P, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (Pn > up_P) SL_HIP(hipMemsetAsync(d_pose_delta.d + 6 * up_P, 0, 6 * (Pn - up_P) * sizeof(double), s));
  if (d_pose_est.ensure(std::max<size_t>(12 * Pn, 1), 12 * up_P, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  UP(d_lm_val, h_lm_val, up_L, 15);
  UP(d_lm_type, h_lm_type, up_L, 1);
  if (d_lm_delta.ensure(std::max<size_t>(9 * Ln, 1), 9 * up_L, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (Ln > up_L) SL_HIP(hipMemsetAsync(d_lm_delta.d + 9 * up_L, 0, 9 * (Ln - up_L) * sizeof(double), s));
  if (d_lm_est.ensure(std::max<size_t>(15 * Ln, 1), 15 * up_L, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  UP(d_pr_pose, h_pr_pose, up_pr, 1);
  UP(d_pr_z, h_pr_z, up_pr, 12);
  UP(d_pr_sigma, h_pr_sigma, up_pr, 6);
  // (the buffers of the last linearisation: an incremental update keeps what nothing touched — HostGraph::lin_gen counts their re-allocations)
  const size_t lin_caps0 = d_pr_r.cap + d_bt_r.cap + d_bt_J0.cap + d_jbuf.cap + d_ebuf.cap + d_lm_Hinv.cap + d_lm_g.cap + d_pose_H.cap + d_pose_g.cap;
  if (d_pr_r.ensure(std::max<size_t>(6 * npr, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  UP(d_bt_i, h_bt_i, up_bt, 1);
  UP(d_bt_j, h_bt_j, up_bt, 1);
  UP(d_bt_z, h_bt_z, up_bt, 12);
  UP(d_bt_sigma, h_bt_sigma, up_bt, 6);
  if (d_bt_r.ensure(std::max<size_t>(6 * nbt, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_bt_J0.ensure(std::max<size_t>(36 * nbt, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  const size_t ngh = h_gh_pose.size();
  UP(d_gh_pose, h_gh_pose, up_gh, 1);
  UP(d_gh_slot, h_gh_slot, up_gh, 1);
  UP(d_gh_first, h_gh_first, up_gh, 1);
  UP(d_gh_z, h_gh_z, up_gh, 12);
  UP(d_gh_sigma, h_gh_sigma, up_gh, 6);
  if (d_gh_r.ensure(std::max<size_t>(6 * ngh, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_gh_J.ensure(std::max<size_t>(36 * ngh, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  UP(d_lf_type, h_lf_type, up_lf, 1);
  UP(d_lf_pose, h_lf_pose, up_lf, 1);
  UP(d_lf_lm, h_lf_lm, up_lf, 1);
  UP(d_lf_slot, h_lf_slot, up_lf, 1);
  UP(d_lf_joff, h_lf_joff, up_lf, 1);
  UP(d_lf_eoff, h_lf_eoff, up_lf, 1);
  UP(d_br_z, h_br_z, up_br, 4);
  UP(d_cu_z, h_cu_z, up_cu, 15);
  UP(d_cu_sigma, h_cu_sigma, up_cu, 9);
  UP(d_cy_z, h_cy_z, up_cy, 7);
#undef UP
  if (d_jbuf.ensure(std::max<int64_t>(jbuf_used, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_ebuf.ensure(std::max<int64_t>(ebuf_used, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_lm_Hinv.ensure(std::max<size_t>(81 * Ln, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_lm_g.ensure(std::max<size_t>(9 * Ln, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_lm_Hacc.ensure(std::max<size_t>(54 * Ln, 1), 0, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_lm_t.ensure(std::max<size_t>(9 * Ln, 1), 0, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_pose_H.ensure(std::max<size_t>(36 * Pn, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_pose_g.ensure(std::max<size_t>(6 * Pn, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_pr_r.cap + d_bt_r.cap + d_bt_J0.cap + d_jbuf.cap + d_ebuf.cap + d_lm_Hinv.cap + d_lm_g.cap + d_pose_H.cap + d_pose_g.cap != lin_caps0) ++lin_gen;
  if (up_csr(csr_lm, d_lm_ptr, d_lm_fids, lm_fids, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (up_csr(csr_pose, d_pose_ptr, d_pose_fids, pose_fids, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  {
    // the two per-entry tables of the pose CSR follow its rewritten tail
    const std::vector<int>& val = csr_pose.val;
    const size_t o0 = csr_pose.off0;
    hc_pose_lms.resize(val.size());
    hc_pose_ed.resize(val.size());
    for (size_t i = o0; i < val.size(); ++i) {
      hc_pose_lms[i] = h_lf_lm[val[i]];
      const int ty = h_lm_type[h_lf_lm[val[i]]];
      const int dim = ty == VT_POINT ? 3 : (ty == VT_CUBE ? 9 : 7);
      hc_pose_ed[i] = ((long long)h_lf_eoff[val[i]] << 4) | dim;
    }
    if (d_pose_lms.ensure(std::max<size_t>(val.size(), 1), o0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_pose_lms.upload(hc_pose_lms.data() + o0, o0, val.size() - o0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_pose_ed.ensure(std::max<size_t>(val.size(), 1), o0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_pose_ed.upload(hc_pose_ed.data() + o0, o0, val.size() - o0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  }
  // k_schur keeps a landmark -> slot table (2 B per landmark) and a pose adjacency bitmap (1 bit per pose) in dynamic LDS
  if ((Ln + 7) / 8 * 8 * sizeof(short) + ((Pn + 31) / 32 + 1) * sizeof(unsigned) > 120 * 1024) {
    (void)ub.flush(s);
    g_last_error = "Schur LDS lookup capacity exceeded (about 60000 landmarks in one graph)";
    return SLIDE_ERR_CAPACITY;
  }
  if (up_csr(csr_bt, d_pose_bt_ptr, d_pose_bt, pose_bt, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  {
    const size_t f0 = std::min<size_t>((size_t)std::max(lm_first_from, 0), std::min(up_L, Ln));      // (new landmarks and every touched one)
    if (d_lm_first.ensure(std::max<size_t>(Ln, 1), f0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_lm_first.upload(h_lm_first.data() + f0, f0, Ln - f0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    lm_first_from = 1 << 30;
  }
  // (the batch stays open across the profile below: its two small arrays ride in the same copy; flushed right after)
  // exact joint step: this robot's border = its shared landmarks and the lambda coordinates of its relative-pose factors
  const bool arrow_now = arrow_on();
  int nbr_new = 0;
  if (arrow_now) {
    const int ns = (int)h_sh_lid.size(), m = h_sep_off.back();
    const bool lam_on = lam_total > 0 && h_gh_gid.size() == h_gh_pose.size();
    if (lam_total > 0 && !lam_on) { g_last_error = "exact joint step: slide_graph_set_ghost_ids must name every ghost factor of the graph"; return SLIDE_ERR_INVALID; }
    h_lm_bord.assign(std::max<size_t>(Ln, 1), -1);
    h_sep_map.assign(std::max(m + lam_total, 1), -1);
    h_gh_bord.assign(std::max<size_t>(h_gh_pose.size(), 1), -1);
    // border items: this robot's shared landmarks and the lambda coordinates of its relative-pose factors, ordered by the first block
    // column of the band in which their coupling row is non-zero (the first observing key frame) — the rows that are still all-zero at a
    // block column are then a SUFFIX of the border, which the steps and the border product skip (the maps below carry the permutation)
    // Nested dissection of the own pose chain (CholBatch::set_segments): cut the chain at n_seg - 1 places; the separator behind a cut at
    // pose q0 is [q0, q1) with q1 = 1 + the furthest pose any pose before q0 couples to (a landmark both observe, a relative-pose factor):
    // nothing couples across it.  Its poses become the FIRST border rows (whole tiles: the second level factors them as a dense system).
    segs.clear();
    h_pose_sep.assign(std::max<size_t>(Pn, 1), -1);
    nsep = nsep_dim = n_sep_poses = 0;
    const int want_seg = batch ? batch->segments() : 1;
    std::vector<std::pair<int, int>> seg_cuts;        // the windows [q0, q1) of poses behind the segments (all but the last)
    std::vector<int> reach;
    if (want_seg >= 2 && Pn >= 64) {
      std::vector<int> lm_last(Ln, -1);
      reach.assign(Pn, 0);
      for (size_t f = 0; f < nlf; ++f) lm_last[h_lf_lm[f]] = std::max(lm_last[h_lf_lm[f]], h_lf_pose[f]);
      for (size_t p = 0; p < Pn; ++p) reach[p] = (int)p;
      for (size_t f = 0; f < nlf; ++f) reach[h_lf_pose[f]] = std::max(reach[h_lf_pose[f]], lm_last[h_lf_lm[f]]);
      for (size_t b = 0; b < nbt; ++b) {
        const int lo = std::min(h_bt_i[b], h_bt_j[b]), hi = std::max(h_bt_i[b], h_bt_j[b]);
        reach[lo] = std::max(reach[lo], hi);
      }
      std::vector<int> run(Pn);                       // furthest reach of the poses 0 .. p
      for (size_t p = 0; p < Pn; ++p) run[p] = std::max(reach[p], p ? run[p - 1] : 0);
      std::vector<std::pair<int, int>> cuts;
      int seg_start = 0;
      bool ok = true;
      std::vector<Seg> tmp;
      for (int i = 1; i < want_seg && ok; ++i) {
        const int q0 = (int)(Pn * (size_t)i / want_seg);
        if (q0 <= seg_start + 8) continue;
        const int q1 = run[q0 - 1] + 1;
        if (q1 <= q0 || q1 + 8 >= (int)Pn) continue;      // (nothing couples across q0: no separator needed — or no room for a segment behind it)
        const Seg sg{6 * seg_start / NB, (6 * q0 + NB - 1) / NB};
        if (!tmp.empty() && sg.t0 < tmp.back().t1) { ok = false; break; }      // (a separator narrower than a tile: segments would share one)
        tmp.push_back(sg);
        cuts.emplace_back(q0, q1);
        seg_start = q1;
      }
      if (ok && !cuts.empty()) {
        const Seg last{6 * seg_start / NB, (int)((6 * Pn + NB - 1) / NB)};
        if (last.t0 >= tmp.back().t1) {
          tmp.push_back(last);
          segs = tmp;
          seg_cuts = cuts;
          for (const auto& c : cuts)
            for (int q = c.first; q < c.second; ++q) { h_pose_sep[q] = 6 * n_sep_poses; ++n_sep_poses; }
          nsep_dim = 6 * n_sep_poses;
          nsep = (nsep_dim + NB - 1) / NB;
        }
      }
    }
    struct Item { int cb, kind, id, dim, goff; };
    std::vector<Item> items;
    std::vector<char> taken((size_t)std::max(m, 0), 0);      // (a caller's own layout with overlapping slots would add two landmarks into the same separator coordinates)
    for (int i = 0; i < ns; ++i) {
      const int lid = h_sh_lid[i];
      if (lid < 0 || (size_t)lid >= Ln) continue;
      const int dim = lm_dim(h_lm_type[lid]);      // (the slots' coordinates may be laid out in any order: separator_offsets orders them along the robots' adjacency)
      if (h_sep_off[i] + dim > m) { g_last_error = "separator offsets do not match the landmark classes of the shared slots"; return SLIDE_ERR_INVALID; }
      for (int k = 0; k < dim; ++k) {
        if (taken[(size_t)h_sep_off[i] + k]) { g_last_error = "separator offsets: the coordinate ranges of two shared slots overlap"; return SLIDE_ERR_INVALID; }
        taken[(size_t)h_sep_off[i] + k] = 1;
      }
      int fp = 1 << 30;
      for (int f : lm_fids[lid]) fp = std::min(fp, h_lf_pose[f]);
      items.push_back(Item{fp == (1 << 30) ? 0 : 6 * fp / NB, 0, lid, dim, h_sep_off[i]});
    }
    if (lam_on)
      for (size_t q = 0; q < h_gh_pose.size(); ++q) {
        for (int k = 0; k < 6; ++k)
          if (h_sep_map[m + 6 * h_gh_gid[q] + k] == -2) { g_last_error = "exact joint step: two ghost factors of one graph name the same measurement"; return SLIDE_ERR_INVALID; }
        for (int k = 0; k < 6; ++k) h_sep_map[m + 6 * h_gh_gid[q] + k] = -2;
        items.push_back(Item{6 * h_gh_pose[q] / NB, 1, (int)q, 6, m + 6 * h_gh_gid[q]});
      }
    std::stable_sort(items.begin(), items.end(), [](const Item& x, const Item& y) { return x.cb < y.cb; });
    int o = nsep * NB;                                 // (the separator poses' rows come first, in whole tiles)
    for (const Item& it : items) {
      if (it.kind == 0) h_lm_bord[it.id] = o; else h_gh_bord[it.id] = o;
      for (int k = 0; k < it.dim; ++k) h_sep_map[it.goff + k] = o + k;
      o += it.dim;
    }
    nbr_new = (o + NB - 1) / NB;
    h_bfirst.assign(nbr_new + 1, 0);
    for (int t = nsep; t < nbr_new; ++t) h_bfirst[t] = 1 << 30;      // (separator-pose rows: always active, 0)
    o = nsep * NB;
    for (const Item& it : items) {
      for (int t = o / NB; t <= (o + it.dim - 1) / NB; ++t) h_bfirst[t] = std::min(h_bfirst[t], it.cb);
      o += it.dim;
    }
    for (int t = 0; t < nbr_new; ++t) if (h_bfirst[t] == (1 << 30)) h_bfirst[t] = 0;
    for (int t = 1; t < nbr_new; ++t) h_bfirst[t] = std::max(h_bfirst[t], h_bfirst[t - 1]);      // (non-decreasing by construction; kept so by force)
    // per-segment activity of the border rows (see host_graph.hpp)
    seg_ord.clear(); seg_sfirst.clear(); seg_tab.clear();
    if (!segs.empty()) {
      const int NS = (int)segs.size(), INF = 1 << 30;
      std::vector<std::vector<int>> sf(NS, std::vector<int>(nbr_new + 1, INF));
      auto seg_of = [&](int col) { for (int q = 0; q < NS; ++q) if (col >= segs[q].t0 && col < segs[q].t1) return q; return -1; };
      auto touch = [&](int t_lo, int t_hi, int pose) {
        if (pose < 0 || (size_t)pose >= Pn || h_pose_sep[pose] >= 0) return;      // (a cut's pose: that coupling lives in the border block)
        const int c = 6 * pose / NB, q = seg_of(c);
        if (q < 0) return;
        for (int t = t_lo; t <= t_hi; ++t) sf[q][t] = std::min(sf[q][t], c);
      };
      int o2 = nsep * NB;
      for (const Item& it : items) {
        const int t_lo = o2 / NB, t_hi = (o2 + it.dim - 1) / NB;
        if (it.kind == 0) { for (int f : lm_fids[it.id]) touch(t_lo, t_hi, h_lf_pose[f]); }
        else touch(t_lo, t_hi, h_gh_pose[it.id]);
        o2 += it.dim;
      }
      // the cuts' poses: window w lies between segment w and segment w + 1; before it, the poses whose reach crosses its start couple to
      // it (contiguous up to the cut); behind it, the segment's first block columns do
      int nb4 = 0;
      for (size_t w = 0; w < seg_cuts.size(); ++w) {
        const int q0 = seg_cuts[w].first, q1 = seg_cuts[w].second;
        const int t_lo = 6 * nb4 / NB, t_hi = (6 * (nb4 + q1 - q0) - 1) / NB;
        int pf = q0;
        for (int pz = q0 - 1; pz >= 0 && 6 * pz / NB >= segs[w].t0; --pz) if (reach[pz] >= q0) pf = pz;
        if (pf < q0) for (int t = t_lo; t <= t_hi; ++t) sf[w][t] = std::min(sf[w][t], std::max(6 * pf / NB, segs[w].t0));
        if (w + 1 < segs.size()) for (int t = t_lo; t <= t_hi; ++t) sf[w + 1][t] = std::min(sf[w + 1][t], segs[w + 1].t0);
        nb4 += q1 - q0;
      }
      seg_tab.push_back(NS);
      for (int q = 0; q < NS; ++q) seg_tab.push_back(segs[q].t1);
      std::vector<int> flat_ord;
      for (int q = 0; q < NS; ++q) {
        sf[q][nbr_new] = segs[q].t0;                  // the right-hand-side row rides from the segment's first block column
        std::vector<int> ord(nbr_new);
        for (int t = 0; t < nbr_new; ++t) ord[t] = t;
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return sf[q][a] < sf[q][b]; });
        std::vector<int> sorted(nbr_new + 1);
        for (int j = 0; j < nbr_new; ++j) sorted[j] = sf[q][ord[j]];
        sorted[nbr_new] = segs[q].t0;
        seg_ord.push_back(ord); seg_sfirst.push_back(sorted);
        flat_ord.insert(flat_ord.end(), ord.begin(), ord.end());
        seg_tab.insert(seg_tab.end(), sf[q].begin(), sf[q].end());
      }
      {
        // and, behind it, per block column of the band the set of border tile rows some segment works on there (k_border_apply reads
        // only those): T, then T x (low, high) words of a 64-bit mask — T = 0: more than 64 border tile rows, no masks
        const int Tb = (int)((6 * Pn + NB - 1) / NB);
        seg_tab_head = (int)seg_tab.size();
        if (nbr_new <= 64) {
          seg_tab.push_back(Tb);
          for (int c = 0; c < Tb; ++c) {
            unsigned long long m = 0;
            for (int q = 0; q < NS; ++q)
              if (c < segs[q].t1)
                for (int t = 0; t < nbr_new; ++t)
                  if (sf[q][t] <= c && c >= segs[q].t0) m |= 1ull << t;
            seg_tab.push_back((int)(unsigned)(m & 0xffffffffull));
            seg_tab.push_back((int)(unsigned)(m >> 32));
          }
        } else seg_tab.push_back(0);
      }
      if (d_seg_ord.ensure(std::max<size_t>(flat_ord.size(), 1), 0, s) != SLIDE_OK || d_seg_tab.ensure(seg_tab.size(), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
      if (!flat_ord.empty()) SL_HIP(hipMemcpyAsync(d_seg_ord.d, flat_ord.data(), flat_ord.size() * sizeof(int), hipMemcpyHostToDevice, s));
      SL_HIP(hipMemcpyAsync(d_seg_tab.d, seg_tab.data(), seg_tab.size() * sizeof(int), hipMemcpyHostToDevice, s));
      SL_HIP(hipStreamSynchronize(s));      // (pageable temporaries)
    }
    if (d_pose_sep.ensure(h_pose_sep.size(), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    SL_HIP(hipMemcpyAsync(d_pose_sep.d, h_pose_sep.data(), h_pose_sep.size() * sizeof(int), hipMemcpyHostToDevice, s));
    if (nsep > 0) {
      if (d_Ld2.ensure((size_t)nsep * NB * NB, 0, s) != SLIDE_OK || d_Winv2.ensure((size_t)nsep * 1024, 0, s) != SLIDE_OK ||
          d_yv2.ensure((size_t)nsep * NB, 0, s) != SLIDE_OK || d_dp2.ensure((size_t)nsep * NB, 0, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
    }
    if (d_lm_bord.ensure(h_lm_bord.size(), 0, s) != SLIDE_OK || d_sep_map.ensure(h_sep_map.size(), 0, s) != SLIDE_OK ||
        d_bfirst.ensure(h_bfirst.size(), 0, s) != SLIDE_OK || d_gh_bord.ensure(h_gh_bord.size(), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    SL_HIP(hipMemcpyAsync(d_gh_bord.d, h_gh_bord.data(), h_gh_bord.size() * sizeof(int), hipMemcpyHostToDevice, s));
    SL_HIP(hipMemcpyAsync(d_lm_bord.d, h_lm_bord.data(), h_lm_bord.size() * sizeof(int), hipMemcpyHostToDevice, s));
    SL_HIP(hipMemcpyAsync(d_sep_map.d, h_sep_map.data(), h_sep_map.size() * sizeof(int), hipMemcpyHostToDevice, s));
    SL_HIP(hipMemcpyAsync(d_bfirst.d, h_bfirst.data(), h_bfirst.size() * sizeof(int), hipMemcpyHostToDevice, s));
    SL_HIP(hipStreamSynchronize(s));      // (host vectors that may change right after)
    if (nbr_new > 0) {
      if (d_bord.ensure((size_t)(nbr_new + 1) * NB * nbr_new * NB, 0, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
      // bord0: what the assembly writes — the same positions in every pass of one topology, so it is zeroed HERE (whenever the border's
      // layout may have changed) and never per pass
      if (d_bord0.ensure((size_t)(nbr_new + 1) * NB * nbr_new * NB, 0, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
      SL_HIP(hipMemsetAsync(d_bord0.d, 0, (size_t)(nbr_new + 1) * NB * nbr_new * NB * sizeof(double), s));
      if (d_xloc.ensure((size_t)nbr_new * NB, 0, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
      SL_HIP(hipMemsetAsync(d_xloc.d, 0, (size_t)nbr_new * NB * sizeof(double), s));
    }
  }
  nbr = nbr_new;
  // dense reduced system
  const int T = (int)((6 * Pn + NB - 1) / NB);
  bool fresh_S = false;
  if (T > Tcap || nbr_new != nbr_alloc || (arrow_now && T != arrow_T)) {
    fresh_S = true;
    if (T > Tcap) {
      int nc = Tcap ? Tcap : 4;
      while (nc < T) nc = nc + nc / 2 + 1;
      Tcap = nc;
    }
    nbr_alloc = nbr_new;
    arrow_T = T;
    ++S_gen;                 // (a resident factor does not survive the re-allocation)
    const size_t ld = (size_t)(Tcap + nbr_alloc + 1) * NB;      // band rows, border rows (exact joint step), the right-hand-side tile row
    // S is rewritten by every Schur pass, so nothing is carried over; zero once so the never-written
    // strict upper tiles and the idle rows of the RHS tile hold finite values.
    if (d_S.ensure_exact(ld * (size_t)Tcap * NB, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_pcg.ensure((size_t)PCG_VEC_COUNT * Tcap * NB, 0, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_pcg_scal.ensure(8, 0, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_Ld.ensure((size_t)Tcap * NB * NB, 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_Winv.ensure((size_t)Tcap * 1024, 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_cctr.ensure((size_t)Tcap + 2, 0, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_yv.ensure((size_t)Tcap * NB, 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    // (the last solve's solution survives the re-allocation: the bounded back-substitution compares against it)
    if (d_dp.ensure((size_t)Tcap * NB, std::min(d_dp.cap, (size_t)std::max(wf_T, 0) * NB), s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_dp_prev.ensure((size_t)Tcap * NB, 0, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;
    joint_Tcap = 0;      // (the joint-solve buffers follow S's leading dimension)
  }
  // Buffers of the joint solve (the saved system, the f32 factor copy, the chain tables): only for graphs that take part in one —
  // a streaming replica never pays for them (three more allocations of S's size at every growth step were 100 ms spikes there)
  {
    const bool want_joint = pcg_iters > 0 || (batch && batch->pcg() > 0);
    if (want_joint && joint_Tcap != Tcap) {
      const size_t ld = (size_t)(Tcap + nbr_alloc + 1) * NB;
      if (d_S0.ensure_exact(ld * (size_t)Tcap * NB, s) != SLIDE_OK) return SLIDE_ERR_HIP;
      if (d_L32.ensure_exact((size_t)Tcap * (Tcap - 1) / 2 * NB * NB + 4, s) != SLIDE_OK) return SLIDE_ERR_HIP;
      if (d_ctab.ensure_exact((size_t)4 * Tcap * NB * NB, s) != SLIDE_OK) return SLIDE_ERR_HIP;
      joint_Tcap = Tcap;
    }
  }
  // Profile of the reduced pose system at tile level.  reach(p) = the last pose coupled to pose p (a landmark both observe, a
  // relative-pose factor); block column c reaches the tile row of the farthest reach of its poses; the running maximum over c is
  // closed under the fill of the factorisation (eliminating column c fills rows <= prof[c] of columns <= prof[c] only).
  {
    const std::vector<int>& reach = h_reach;      // (kept current by merge_pending: no pass over all factors per update)
    const bool dense = force_dense;      // measurement aid (slide_graph_set_dense_profile / SLIDE_CHOL_DENSE=1): ignore the structure
    std::vector<int> prof(T), first(T);
    for (int c = 0; c < T; ++c) prof[c] = dense ? T - 1 : c;
    for (size_t p = 0; p < Pn && !dense; ++p) {
      const int rt = (6 * reach[p] + 5) / NB;
      const int ta = (int)(6 * p) / NB, tb = (int)(6 * p + 5) / NB;
      prof[ta] = std::max(prof[ta], rt);
      prof[tb] = std::max(prof[tb], rt);
    }
    for (int c = 1; c < T; ++c) prof[c] = std::max(prof[c], prof[c - 1]);
    for (int r = 0, c = 0; r < T; ++r) {
      while (prof[c] < r) ++c;
      first[r] = c;
    }
    if (prof != h_prof) {
      // S outside the profile must hold zeros.  A freshly allocated S does; when T grows, the old right-hand-side row (one row of
      // the old last tile row + 1) becomes a matrix row and is cleared; when a tile LEAVES the profile (never on a growing graph) the
      // used part of S is cleared as a whole.
      const int Told = (int)h_prof.size();
      bool shrink = T < Told;
      for (int c = 0; !shrink && c < std::min(T, Told); ++c) shrink = prof[c] < h_prof[c];
      const size_t ld = (size_t)(Tcap + nbr_alloc + 1) * NB;
      if (!fresh_S && d_S.d) {
        if (shrink) ++S_gen;
        if (shrink) SL_HIP(hipMemsetAsync(d_S.d, 0, ld * (size_t)std::max(T, Told) * NB * sizeof(double), s));
        else if (T > Told && Told > 0)      // (a graph with a border is re-allocated whenever T changes: fresh_S)
          SL_HIP(hipMemset2DAsync(d_S.d + (size_t)Told * NB, ld * sizeof(double), 0, sizeof(double), (size_t)Told * NB, s));
      }
      h_prof = prof;
      h_first = first;
      h_first.push_back(0);      // (one more entry: the right-hand-side row reaches every column — the catch-up product of an incremental update)
      if (d_prof.ensure(std::max<size_t>(T, 1), 0, s) != SLIDE_OK || d_first.ensure(std::max<size_t>(T + 1, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
      if (T) {      // (into the open upload batch: staged at once, no copy or synchronisation of their own)
        if (d_prof.upload(h_prof.data(), 0, T, s) != SLIDE_OK || d_first.upload(h_first.data(), 0, T + 1, s) != SLIDE_OK) return SLIDE_ERR_HIP;
      }
      ++prof_ver;
    }
  }
  if (ub.flush(s) != SLIDE_OK) return SLIDE_ERR_HIP;   // (the host temporaries were copied into the pinned staging buffer)
  topo_dirty = false;
  uploaded_once = true;
  up_P = Pn; up_L = Ln; up_pr = npr; up_bt = nbt; up_lf = nlf; up_gh = ngh;
  up_br = h_br_z.size() / 4; up_cu = h_cu_z.size() / 15; up_cy = h_cy_z.size() / 7;

  G.P = (int)Pn; G.L = (int)Ln;
  G.pose_val = d_pose_val.d; G.pose_delta = d_pose_delta.d; G.pose_est = d_pose_est.d;
  G.lm_type = d_lm_type.d; G.lm_val = d_lm_val.d; G.lm_delta = d_lm_delta.d; G.lm_est = d_lm_est.d;
  G.n_prior = (int)npr; G.pr_pose = d_pr_pose.d; G.pr_z = d_pr_z.d; G.pr_sigma = d_pr_sigma.d; G.pr_r = d_pr_r.d;
  G.n_between = (int)nbt; G.bt_i = d_bt_i.d; G.bt_j = d_bt_j.d; G.bt_z = d_bt_z.d; G.bt_sigma = d_bt_sigma.d;
  G.bt_r = d_bt_r.d; G.bt_J0 = d_bt_J0.d;
  G.n_ghost = (int)ngh; G.gh_pose = d_gh_pose.d; G.gh_slot = d_gh_slot.d; G.gh_first = d_gh_first.d; G.gh_z = d_gh_z.d;
  G.gh_sigma = d_gh_sigma.d; G.gh_r = d_gh_r.d; G.gh_J = d_gh_J.d;
  G.n_gslots = (int)h_gslot_pose.size(); G.ghost_val = d_ghost_val.d; G.gslot_pose = d_gslot_pose.d;
  G.n_lf = (int)nlf; G.lf_type = d_lf_type.d; G.lf_pose = d_lf_pose.d; G.lf_lm = d_lf_lm.d; G.lf_slot = d_lf_slot.d;
  G.lf_joff = d_lf_joff.d; G.lf_eoff = d_lf_eoff.d;
  G.br_z = d_br_z.d; G.cu_z = d_cu_z.d; G.cu_sigma = d_cu_sigma.d; G.cy_z = d_cy_z.d;
  G.jbuf = d_jbuf.d; G.ebuf = d_ebuf.d;
  G.lm_ptr = d_lm_ptr.d; G.lm_fids = d_lm_fids.d; G.pose_ptr = d_pose_ptr.d; G.pose_fids = d_pose_fids.d; G.pose_lms = d_pose_lms.d; G.pose_ed = d_pose_ed.d;
  G.pose_bt_ptr = d_pose_bt_ptr.d; G.pose_bt = d_pose_bt.d;
  G.adj_words = (int)((Pn + 31) / 32 + 1);
  if (d_pose_adj.ensure(std::max<size_t>(Pn * (size_t)G.adj_words, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  G.pose_adj = d_pose_adj.d;
  G.lm_Hacc = d_lm_Hacc.d; G.lm_t = d_lm_t.d; G.n_slots = (int)h_sh_lid.size(); G.sh_lid = d_sh_lid.d; G.sh_owner = d_sh_owner.d;
  G.lm_Hinv = d_lm_Hinv.d; G.lm_g = d_lm_g.d; G.pose_H = d_pose_H.d; G.pose_g = d_pose_g.d;
  G.S = d_S.d; G.ld = (Tcap + nbr_alloc + 1) * NB; G.T = T; G.Ld = d_Ld.d; G.Winv = d_Winv.d; G.yv = d_yv.d; G.dp = d_dp.d; G.chol_ctr = d_cctr.d;
  G.prof = d_prof.d; G.first = d_first.d; G.prof_ver = prof_ver;
  {
    int band = 0;
    for (int c = 0; c < (int)h_prof.size(); ++c) band = std::max(band, h_prof[c] - c);
    G.schur_split = band <= 8 ? 1 : 2;
  }
  G.S0 = d_S0.d; G.save_S0 = 0; G.pcg = d_pcg.d; G.pcg_scal = d_pcg_scal.d;
  G.pcg_tol2 = (batch ? batch->pcg_tolerance() : pcg_tol) * (batch ? batch->pcg_tolerance() : pcg_tol);
  G.arrow = arrow_now ? 1 : 0; G.nbr = nbr; G.lm_bord = arrow_now ? d_lm_bord.d : nullptr; G.bord = arrow_now ? d_bord.d : nullptr; G.bord0 = arrow_now ? d_bord0.d : nullptr;
  G.ldb = (nbr + 1) * NB;
  G.gh_bord = (arrow_now && lam_total > 0) ? d_gh_bord.d : nullptr;
  G.seg_tab = nullptr;
  if (arrow_now && !seg_tab.empty() && nbr > 0 && !getenv("SLIDE_FULL_CLEAR")) {
    // the per-pass clear of the border rows is restricted to what the segments work on; whatever an earlier layout left elsewhere goes now
    G.seg_tab = d_seg_tab.d;
    SL_HIP(hipMemset2DAsync(d_S.d + (size_t)T * NB, (size_t)(Tcap + nbr_alloc + 1) * NB * sizeof(double), 0, (size_t)(nbr + 1) * NB * sizeof(double), (size_t)T * NB, s));
  }
  G.pose_sep = (arrow_now && nsep > 0) ? d_pose_sep.d : nullptr; G.nsep = arrow_now ? nsep : 0; G.nsep_dim = arrow_now ? nsep_dim : 0;
  // the segments' own profiles (their rows end where the separator begins: what lay beyond moved into the border)
  seg_prof.clear(); seg_prof_off.clear();
  if (arrow_now && nsep > 0) {
    std::vector<int> flat;
    for (const Seg& sg : segs) {
      std::vector<int> pv(sg.t1 - sg.t0);
      for (int c = sg.t0; c < sg.t1; ++c) pv[c - sg.t0] = std::min(h_prof[c], sg.t1 - 1) - sg.t0;
      seg_prof_off.push_back(flat.size());
      flat.insert(flat.end(), pv.begin(), pv.end());
      seg_prof.push_back(pv);
    }
    if (d_seg_prof.ensure(std::max<size_t>(flat.size(), 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    SL_HIP(hipMemcpyAsync(d_seg_prof.d, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice, s));
    SL_HIP(hipStreamSynchronize(s));
  }
  G.status = d_status.d;
  G.lm_first = d_lm_first.d; G.col0 = 0; G.pose0 = 0;
  G.chart = P.pose_chart;
  G.bearing_sigma = P.bearing_range_sigma; G.cyl_sigma = P.cylinder_sigma; G.numdiff_delta = P.numdiff_delta;
  launch_pose_adj(G, s);             // the topology changed: rebuild the pose adjacency of the Schur assembly
  G.sp_idx = nullptr; G.sp_pairs = nullptr; G.sp_w = 0;
  if (arrow_now && !getenv("SLIDE_SCHUR_WALK")) {
    const int rc = build_schur_pairs(s);
    if (rc != SLIDE_OK) return rc;
  }
  return sync_lm_slot();
}
// Pair lists of the Schur assembly.  k_schur's block (pi, pj) is the sum over the pairs (factor x of pose pi, factor y of pose pj) on one
// landmark of F_x E_y^T; walking pose pi's list and looking every landmark up in pose pj's costs ~20 probes per block for ~2 hits.  Between
// the passes of a batched job the topology does not change, so the hits are listed once, in the order the walk finds them (x ascending
// in pi's list, y ascending in pj's).  Separator landmarks are left out (H_ll^-1 = 0 there: their F is zero).  Only inside a monotone
// profile whose strips are at most 256 poses wide; otherwise (and on every streaming update, which changes the topology) k_schur walks.
int HostGraph::build_schur_pairs(hipStream_t s) {
  const int Pn = (int)pose_fids.size();
  if (Pn == 0 || h_prof.empty() || !G.prof) return SLIDE_OK;
  auto p_end = [&](int pj) { return std::min(Pn, ((h_prof[(6 * pj + 5) / NB] + 1) * NB + 5) / 6); };
  int W = 1;
  for (int pj = 0; pj < Pn; ++pj) W = std::max(W, p_end(pj) - pj);
  if (W > 256) return SLIDE_OK;
  std::vector<int> idx((size_t)2 * Pn * W, 0);
  std::vector<long long> pairs;
  auto ed_of = [&](int f) {
    const int ty = h_lm_type[h_lf_lm[f]];
    return ((long long)h_lf_eoff[f] << 4) | (ty == VT_POINT ? 3 : (ty == VT_CUBE ? 9 : 7));
  };
  const bool have_bord = !h_lm_bord.empty();
  std::vector<std::vector<std::pair<int, int>>> rows(W);      // per d = pi - pj: (x, y) factor ids
  for (int pj = 0; pj < Pn; ++pj) {
    const int pe = p_end(pj);
    for (auto& r : rows) r.clear();
    // every factor y of pose pj: the other factors x of its landmark on poses pi in [pj, pe)
    for (int y : pose_fids[pj]) {
      const int l = h_lf_lm[y];
      if (have_bord && (size_t)l < h_lm_bord.size() && h_lm_bord[l] >= 0) continue;
      for (int x : lm_fids[l]) {
        const int pi = h_lf_pose[x];
        if (pi >= pj && pi < pe) rows[pi - pj].emplace_back(x, y);
      }
    }
    for (int d = 0; d < pe - pj; ++d) {
      auto& r = rows[d];
      if (r.empty()) continue;
      // the walk's order: x in the order of pose pi's list (landmark id, factor id), then y in the order of pose pj's list
      std::sort(r.begin(), r.end(), [&](const std::pair<int, int>& a, const std::pair<int, int>& b) {
        const int la = h_lf_lm[a.first], lb = h_lf_lm[b.first];
        if (la != lb) return la < lb;
        if (a.first != b.first) return a.first < b.first;
        return a.second < b.second;
      });
      idx[2 * ((size_t)pj * W + d)] = (int)(pairs.size() / 2);
      idx[2 * ((size_t)pj * W + d) + 1] = (int)r.size();
      for (const auto& xy : r) { pairs.push_back(ed_of(xy.first)); pairs.push_back(ed_of(xy.second)); }
    }
  }
  if (pairs.size() / 2 > (size_t)0x7fffffff) return SLIDE_OK;
  if (d_sp_idx.ensure(idx.size(), 0, s) != SLIDE_OK || d_sp_pairs.ensure(std::max<size_t>(pairs.size(), 2), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  SL_HIP(hipMemcpyAsync(d_sp_idx.d, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, s));
  if (!pairs.empty()) SL_HIP(hipMemcpyAsync(d_sp_pairs.d, pairs.data(), pairs.size() * sizeof(long long), hipMemcpyHostToDevice, s));
  SL_HIP(hipStreamSynchronize(s));      // (pageable temporaries)
  G.sp_idx = d_sp_idx.d; G.sp_pairs = d_sp_pairs.d; G.sp_w = W;
  return SLIDE_OK;
}
int HostGraph::sync_lm_slot() {
  if (h_sh_lid.empty()) {            // no shared slots (single-robot graphs, replicas): nothing reads the table
    G.lm_slot = nullptr;
    return SLIDE_OK;
  }
  const size_t Ln = h_lm_type.size();
  std::vector<int> slot(std::max<size_t>(Ln, 1), -1);
  for (size_t i = 0; i < h_sh_lid.size(); ++i)
    if (h_sh_lid[i] >= 0 && (size_t)h_sh_lid[i] < Ln) slot[h_sh_lid[i]] = (int)i;
  if (d_lm_slot.ensure(slot.size(), 0, stream) != SLIDE_OK) return SLIDE_ERR_HIP;
  SL_HIP(hipMemcpyAsync(d_lm_slot.d, slot.data(), slot.size() * sizeof(int), hipMemcpyHostToDevice, stream));
  SL_HIP(hipStreamSynchronize(stream));      // (a pageable temporary)
  G.lm_slot = d_lm_slot.d;
  return SLIDE_OK;
}

int HostGraph::enqueue_iteration(bool lookahead, bool skip_relin, int c_d, int wf_cd) {
  if (wf_cd < 0) wf_cd = c_d;
  hipStream_t s = stream;
  static const char* kNames[] = {"relin", "linearize", "landmark_reduce", "pose_reduce", "schur_assemble", "chol_step",
                                 "chol_catchup", "chol_extract_y", "chol_bwd", "backsub", "estimate"};
  int id[11];
  for (int i = 0; i < 11; ++i) id[i] = prof.id_of(kNames[i]);
#define STAGE(i, call) do { prof.begin(id[i], s); call; prof.end(s); } while (0)
  if (!skip_relin) STAGE(0, launch_relin(G, s));
  STAGE(1, launch_linearize(G, s));
  STAGE(2, launch_landmark(G, 0, s));
  STAGE(3, launch_pose(G, s));
  STAGE(4, launch_schur(G, s));
  (void)lookahead;
  if (c_d > 0 && c_d < G.T) {
    // Incremental re-factorisation: block columns < c_d keep the factor of the last solve (the assembly left them alone, G.col0).  The
    // re-assembled trailing tiles (i, j >= c_d) and the right-hand-side row first catch up with those columns' panels,
    //     S(i, j) -= sum_{k < c_d} L(i, k) L(j, k)^T        (k_border_syrk on the view: "border rows" = tile rows c_d .. T, K = the columns < c_d,
    //                                                        first[] skips what lies outside the profile),
    // then the steps run on the trailing sub-matrix as a system of its own (same kernels, shifted base pointers and profile).
    CholSystem cu{};
    cu.S = G.S; cu.ld = G.ld; cu.T = c_d; cu.nbr = G.T - c_d; cu.ldb = G.ld;
    cu.bord = G.S + (size_t)c_d * NB * G.ld + (size_t)c_d * NB; cu.bfirst = G.first + c_d;
    STAGE(6, launch_border_syrk(&cu, 1, s));
    std::vector<int> pv(G.T - c_d);
    for (int c = c_d; c < G.T; ++c) pv[c - c_d] = h_prof[c] - c_d;
    double* Sv = G.S + (size_t)c_d * NB * G.ld + (size_t)c_d * NB;
    for (int k = 0; k < G.T - c_d; ++k)
      STAGE(5, launch_chol_step(Sv, G.ld, k, G.T - c_d, G.Ld + (size_t)(c_d + k) * NB * NB, G.Winv + (size_t)(c_d + k) * 1024, G.status, G.chol_ctr, nullptr,
                                pv.data(), s));
  } else if (c_d < G.T) {
    for (int k = 0; k < G.T; ++k)
      STAGE(5, launch_chol_step(G.S, G.ld, k, G.T, G.Ld + (size_t)k * NB * NB, G.Winv + (size_t)k * 1024, G.status, G.chol_ctr, nullptr, h_prof.data(), s));
  }
  // bounded back-substitution (iSAM2's wildfire threshold; set_wildfire): on an incremental update only, and only below the first
  // re-factored block column — there a block's factor column and forward-substituted right-hand side are the last solve's, so its
  // solution moves only through the blocks above it (bwd_chain_body<.., true>)
  // (wf_cd: the first DIRTY block column; it is c_d on an incremental update, and also bounds the substitution of an update that has
  // to re-factor everything — after a re-allocation of S — although nothing below it changed)
  const int wf_Tp = (skip_relin && wildfire_thr > 0.0 && wf_cd > 0) ? std::min(std::min(wf_T, wf_cd), G.T) : 0;
  if (wf_Tp > 0) {
    STAGE(7, launch_chol_extract_y(G.S, G.ld, G.T, G.yv, G.dp, G.status, s, 0, 0, d_dp_prev.d));
    STAGE(8, launch_chol_bwd_all(G.S, G.ld, G.T, G.Ld, G.Winv, G.yv, G.dp, G.status, G.prof, s, d_dp_prev.d, wildfire_thr, std::min(wf_T, G.T), wf_Tp));
  } else {
    STAGE(7, launch_chol_extract_y(G.S, G.ld, G.T, G.yv, G.dp, G.status, s));
    STAGE(8, launch_chol_solve_bwd(CholSystem{G.S, G.ld, G.T, G.Ld, G.Winv, G.yv, G.dp, G.status, nullptr, h_prof.data(), G.prof, G.first, d_ctab.d}, s));
  }
  STAGE(9, launch_backsub(G, 0, s));
  if (skip_relin && G.lm_first) {
    // (streaming update: also predicts the next update's relinearisation, status[5], and — when run_update asked for it — packs the
    // closing read-back in its last workgroup)
    STAGE(10, launch_estimate_predict(G, s, ei_final_pose, ei_final_out));
    if (ei_final_out) ei_final_done = true;
  }
  else STAGE(10, launch_estimate(G, s));
#undef STAGE
  return SLIDE_OK;
}

int HostGraph::run_update(double relin_thr, int iterations) {
  thread_capture_mode_local();
  hipStream_t s = stream;
  if (G.P == 0) return SLIDE_OK;
  G.relin_thr = relin_thr;
  if (!status_clean) SL_HIP(hipMemsetAsync(d_status.d, 0, 8 * sizeof(int), s));      // (k_final_pack of the last update left them at zero otherwise)
  status_clean = false;
  cache_pose = -1;
  // Replaying a captured hipGraph removes the host launch cost (~250 launches + event traffic per pass) once the
  // SAME resident graph is solved again (batch Gauss-Newton, repeated solve() without new factors).
  const bool same_as_prev = have_prev && std::memcmp(&G_prev, &G, sizeof(GraphDev)) == 0;
  static const bool env_graph = !(getenv("SLIDE_NO_GRAPH") && getenv("SLIDE_NO_GRAPH")[0] == '1');
  // two-stream look-ahead is implemented and parity-tested but measured SLOWER than the linear graph on MI355X
  // (the latency-critical diag+panel blocks queue behind the flood of update workgroups): opt-in only.
  static const bool env_look = getenv("SLIDE_LOOKAHEAD") && getenv("SLIDE_LOOKAHEAD")[0] == '1';
  const bool graph_ok = env_graph && !prof.on && G.T > 4;
  bool use_graph = graph_ok && (iterations > 1 || same_as_prev || (gexec && std::memcmp(&G_cap, &G, sizeof(GraphDev)) == 0));
  G_prev = G;
  have_prev = true;
  if (use_graph && !(gexec && std::memcmp(&G_cap, &G, sizeof(GraphDev)) == 0)) {
    if (gexec) { (void)hipGraphExecDestroy(gexec); gexec = nullptr; }
    hipGraph_t graph = nullptr;
    std::lock_guard<std::mutex> cap(g_capture_mtx);
    SL_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_iteration(env_look);
    const hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc != SLIDE_OK || e != hipSuccess || graph == nullptr) {
      (void)hipGetLastError();
      use_graph = false;
    } else {
      const hipError_t ei = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ei != hipSuccess) { gexec = nullptr; (void)hipGetLastError(); use_graph = false; }
      else G_cap = G;
    }
  }
  // Incremental re-factorisation (the streaming path: one update after a few new factors).  iSAM2 re-eliminates only the part of the
  // Bayes tree the new factors and the relinearised variables touch (ISAM2::update, graph.cpp:260-272); on the banded reduced system of
  // a pose chain that is "the block columns from the first dirty one on".  The lowest dirty pose = min(host: poses of the factors
  // merged since the last solve and every pose observing one of their landmarks; device: the same for the variables k_relin moves,
  // status[6]) — known only after k_relin, hence one extra (32-byte) read-back before the rest of the update is enqueued.
  static const bool env_inc = !(getenv("SLIDE_NO_INCREMENTAL") && getenv("SLIDE_NO_INCREMENTAL")[0] == '1');
  const bool try_inc = env_inc && inc_enabled && iterations == 1 && !use_graph && !batch && !force_dense && factor_valid && fact_gen == S_gen && relin_thr > 0.0 && G.T > 2;
  const bool wf_full_path = !try_inc && env_inc && wildfire_thr > 0.0 && iterations == 1 && !use_graph && !batch && !force_dense && wf_T > 0 && relin_thr > 0.0;
  static const bool env_pred = !(getenv("SLIDE_NO_PREDICT") && getenv("SLIDE_NO_PREDICT")[0] == '1');
  const bool use_pred = env_pred && pred_valid && pred_thr == relin_thr;
  if (d_final.ensure(18, 0, s, true) != SLIDE_OK) return SLIDE_ERR_HIP;      // 16 doubles of the closing read-back + the arrival counter of the fused pack (zero)
  ei_final_done = false;
  static const bool env_fuse_final = !(getenv("SLIDE_NO_FUSED_FINAL") && getenv("SLIDE_NO_FUSED_FINAL")[0] == '1');
  if ((try_inc || wf_full_path) && env_fuse_final) { ei_final_out = d_final.d; ei_final_pose = (int)G.P - 1; }
  else { ei_final_out = nullptr; ei_final_pose = -1; }
  if (try_inc) {
    launch_relin(G, s);
    int pmin = dirty_min_pose;
    if (use_pred) {
      // no read-back: the last update's k_estimate_predict already said which variables k_relin moves now (delta has not changed
      // since) and from which pose on their blocks change.  (Factors merged since then only add to dirty_min_pose: a relinearised
      // pose's new partners / landmarks are poses and first observers that dirty_min_pose already covers.)
      pmin = std::min(pmin, pred_pose);
      ++n_pred_used;
    } else {
      int s0[8];
      SL_HIP(hipMemcpyAsync(s0, d_status.d, 8 * sizeof(int), hipMemcpyDeviceToHost, s));
      SL_HIP(hipStreamSynchronize(s));
      if (s0[6] > 0) pmin = std::min(pmin, G.P - s0[6]);
    }
    int c_d = pmin >= G.P ? G.T : (6 * std::max(pmin, 0)) / NB;      // (nothing dirty: no step at all, the substitutions are simply repeated)
    c_d = std::min(c_d, G.T);
    G.col0 = c_d * NB;
    // the linearisation, the landmark sums / Schur records and the poses' own blocks of everything BEFORE the first dirty pose are the
    // last solve's (nothing they depend on moved — that is what "dirty" means); the kernels skip them when those buffers still hold them
    static const bool env_skip = !(getenv("SLIDE_NO_LIN_SKIP") && getenv("SLIDE_NO_LIN_SKIP")[0] == '1');
    G.pose0 = (env_skip && lin_solved_gen == lin_gen) ? std::max(0, std::min(pmin, (int)G.P)) : 0;
    const int rc = enqueue_iteration(false, true, c_d);
    G.col0 = 0;
    G.pose0 = 0;
    if (rc != SLIDE_OK) return rc;
    if (c_d > 0) ++n_inc; else ++n_full;
    last_cd = c_d;
  } else if (wf_full_path) {
    // an update that cannot keep any factor column (S was re-allocated, the system is still tiny, ...) but has a previous solution:
    // everything is re-factored, the back-substitution is bounded below the first dirty block column all the same (the rule of
    // bwd_chain_body<.., true> does not care who recomputed the unchanged columns)
    launch_relin(G, s);
    int pmin = dirty_min_pose;
    if (use_pred) {
      pmin = std::min(pmin, pred_pose);
      ++n_pred_used;
    } else {
      int s0[8];
      SL_HIP(hipMemcpyAsync(s0, d_status.d, 8 * sizeof(int), hipMemcpyDeviceToHost, s));
      SL_HIP(hipStreamSynchronize(s));
      if (s0[6] > 0) pmin = std::min(pmin, G.P - s0[6]);
    }
    const int cd = std::min(pmin >= G.P ? G.T : (6 * std::max(pmin, 0)) / NB, G.T);
    const int rc = enqueue_iteration(false, true, 0, cd);
    if (rc != SLIDE_OK) return rc;
    ++n_full;
    last_cd = 0;
  } else {
    for (int it = 0; it < iterations; ++it) {
      if (use_graph) SL_HIP(hipGraphLaunch(gexec, s));
      else {
        const int rc = enqueue_iteration(false);
        if (rc != SLIDE_OK) return rc;
      }
    }
    ++n_full;
    last_cd = 0;
  }
  // the closing read-back in one piece: status words + the newest pose's estimate (what a frame returns: HostGraph::get_pose12 serves it
  // from the cache) — k_final_pack leaves the status words at zero for the next update
  int st[8];
  pred_valid = false;
  {
    double fin[16];
    const int newest = (int)G.P - 1;
    if (!ei_final_done) launch_final_pack(G, newest, d_final.d, s);      // (otherwise k_estimate_predict's last workgroup packed them)
    ei_final_out = nullptr; ei_final_pose = -1; ei_final_done = false;
    SL_HIP(hipMemcpyAsync(fin, d_final.d, sizeof(fin), hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    std::memcpy(st, fin, sizeof(st));
    std::memcpy(cache_pose12, fin + 4, 12 * sizeof(double));
    cache_pose = newest;
    status_clean = true;
  }
  SL_HIP(hipGetLastError());
  if (prof.on) prof.collect();
  last_relin = st[2];
  factor_valid = false;
  {
    const int rc = decode_status(st);
    if (rc != SLIDE_OK) { cache_pose = -1; return rc; }
  }
  if (iterations == 1 && !use_graph && !batch && G.lm_first && relin_thr > 0.0) {
    // (the streaming paths ran k_estimate_predict: st[5] = P - lowest pose the next relinearisation changes, 0: nothing moves)
    pred_pose = st[5] > 0 ? (int)G.P - st[5] : (1 << 30);
    pred_thr = relin_thr;
    pred_valid = (try_inc || wf_full_path);
  }
  factor_valid = true;
  fact_gen = S_gen;
  lin_solved_gen = lin_gen;         // (every linearisation buffer holds this solve's state)
  dirty_min_pose = 1 << 30;
  wf_T = G.T;                       // (dp holds this solve's solution for every block column)
  last_wf_kept = st[3];
  n_wf_kept += st[3];
  return SLIDE_OK;
}

// ---- one robot per GPU ---------------------------------------------------------------------------------
// Slots are a global, rank-independent enumeration of the landmarks seen by more than one robot; slot i maps
// to this rank's landmark (cls[i], idx[i]) or to none (cls[i] < 0).
int HostGraph::set_shared(const int32_t* cls, const int64_t* idx, const int32_t* owner, int n_slots) {
  topo_dirty = true;      // the slot tables are part of the device view upload_new assembles
  int rc = merge_pending();
  if (rc != SLIDE_OK) return rc;
  rc = upload_new();
  if (rc != SLIDE_OK) return rc;
  h_sh_lid.assign(n_slots, -1);
  h_sh_owner.assign(n_slots, 0);
  for (int i = 0; i < n_slots; ++i) {
    if (cls[i] < 0) continue;
    const int lid = lm_lid(cls[i], (uint64_t)idx[i]);
    if (lid < 0) { g_last_error = "set_shared: landmark is not in the graph"; return SLIDE_ERR_INVALID; }
    h_sh_lid[i] = lid;
    h_sh_owner[i] = owner[i] ? 1 : 0;
  }
  if (d_sh_lid.ensure(std::max(n_slots, 1), 0, stream) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_sh_owner.ensure(std::max(n_slots, 1), 0, stream) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_sh_lid.upload(h_sh_lid.data(), 0, n_slots, stream) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_sh_owner.upload(h_sh_owner.data(), 0, n_slots, stream) != SLIDE_OK) return SLIDE_ERR_HIP;
  SL_HIP(hipStreamSynchronize(stream));
  G.n_slots = n_slots; G.sh_lid = d_sh_lid.d; G.sh_owner = d_sh_owner.d;
  return sync_lm_slot();
}

// Distributed Gauss-Newton pass = phase 0, all-reduce(buf: 54/slot), phase 1, all-reduce(buf: 9/slot), phase 2.
//   0: relinearise (threshold 0), linearise, per-landmark partial sums, pack H_ll / g_l of the shared slots
//   1: unpack the summed blocks, invert H_ll, pose reduce, Schur, Cholesky, pose solve, t_l, pack t_l
//   2: unpack summed t_l, landmark back-substitution, estimate
//  10: pack the owner's landmark values (15/slot)      11: unpack them (every rank adopts the owner's value)
// Every robot solves its own reduced pose system with the GLOBAL landmark blocks (block-Jacobi over robots on
// the Schur complement, exact gradient): the fixed point is the joint optimum.
bool HostGraph::arrow_on() const {
  return batch && batch->is_arrow() && !h_sh_lid.empty() && h_sep_off.size() == h_sh_lid.size() + 1;
}
int HostGraph::set_ghost_ids(const int32_t* ids, int n, int n_total) {
  if (n < 0 || n_total < 0 || (n > 0 && !ids)) return SLIDE_ERR_INVALID;
  for (int i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= n_total) { g_last_error = "set_ghost_ids: index outside the job's list of relative-pose measurements"; return SLIDE_ERR_INVALID; }
  h_gh_gid.assign(ids, ids + n);
  lam_total = 6 * n_total;
  topo_dirty = true;
  return SLIDE_OK;
}
int HostGraph::set_separator(const int32_t* off, int n) {
  if (n < 0 || (n > 0 && !off)) return SLIDE_ERR_INVALID;
  for (int i = 0; i + 1 < n; ++i)
    if (off[i] < 0 || off[i] > off[n - 1]) { g_last_error = "set_separator: an offset lies outside the separator (the last entry is its dimension)"; return SLIDE_ERR_INVALID; }
  h_sep_off.assign(off, off + n);
  topo_dirty = true;
  return SLIDE_OK;
}
void HostGraph::set_pcg(int iters, double tol) {
  pcg_iters = iters < 0 ? 0 : iters;
  pcg_tol = tol > 0.0 ? tol : 0.0;
  topo_dirty = true;                      // (the joint-solve buffers are allocated by upload_new on demand)
  for (auto& pg : phase_graph)            // the captured phase 1 depends on it
    if (pg.exec) { (void)hipGraphExecDestroy(pg.exec); pg.exec = nullptr; }
}
int HostGraph::sync_self() {
  if (have_self && std::memcmp(&G_self, &G, sizeof(GraphDev)) == 0) return SLIDE_OK;
  if (d_Gself.ensure(1, 0, stream) != SLIDE_OK) return SLIDE_ERR_HIP;
  SL_HIP(hipMemcpyAsync(d_Gself.d, &G, sizeof(GraphDev), hipMemcpyHostToDevice, stream));
  SL_HIP(hipStreamSynchronize(stream));       // (&G is host memory that may change right after)
  G_self = G;
  have_self = true;
  return SLIDE_OK;
}

int HostGraph::enqueue_phase(int phase, double* d_buf) {
  pred_valid = false; status_clean = false; cache_pose = -1;      // (outside the streaming update: nothing it left behind can be relied on)
  hipStream_t s = stream;
  const bool joint = pcg_iters > 0 && G.n_slots > 0 && !batch;      // un-batched joint solve: phases 31 / 32 / 33 follow phase 1
  if (phase == 0) {
    launch_relin(G, s);
    launch_linearize(G, s);
    launch_landmark(G, 1, s);
    launch_shared_pack(G, 0, d_buf, s);
  } else if (phase == 1 || phase == 3 || phase == 4) {      // 3 / 4: the parts of phase 1 before / after the factor + solve
    if (phase != 4) {
      launch_shared_unpack(G, 0, d_buf, s);
      launch_landmark(G, 2, s);
      launch_pose(G, s);
      launch_schur(G, s);
    }
    if (phase == 1) {
      if (joint) SL_HIP(hipMemcpyAsync(G.S0, G.S, (size_t)G.ld * G.T * NB * sizeof(double), hipMemcpyDeviceToDevice, s));
      const int rc = factor_and_solve(s);
      if (rc != SLIDE_OK) return rc;
      if (joint) {
        launch_pcg_init(d_Gself.d, &G, 1, s);
        launch_pcg_tl(d_Gself.d, &G, 1, &d_buf, PCG_VEC_U, s);
        return SLIDE_OK;
      }
    }
    if (phase != 3) {
      launch_backsub(G, 1, s);
      launch_shared_pack(G, 1, d_buf, s);
    }
  } else if (phase == 31) {          // after the exchange of t_l: w = S u, partial dots -> d_buf[0 .. 1]
    launch_pcg_matvec_dots(d_Gself.d, &G, 1, &d_buf, 1, s);
  } else if (phase == 32 || phase == 33) {     // after the exchange of the dots: the updates; 32: next u, t_l -> d_buf; 33 (last): dp = x, t_l(dp) -> d_buf
    launch_pcg_update(d_Gself.d, &G, 1, &d_buf, 1, s);
    if (phase == 33) {
      launch_pcg_finish(d_Gself.d, &G, 1, s);
      launch_backsub(G, 1, s);
      launch_shared_pack(G, 1, d_buf, s);
    } else {
      const size_t nT = (size_t)G.T * NB;
      const CholSystem cs{G.S, G.ld, G.T, G.Ld, G.Winv, G.yv, G.dp, G.status, d_L32.d, h_prof.data(), G.prof, G.first, d_ctab.d};
      const double* in = G.pcg + PCG_VEC_R * nT;
      double* out = G.pcg + PCG_VEC_Y * nT;
      double* nxt = G.pcg + PCG_VEC_U * nT;
      launch_chain_batch(&cs, 1, &in, &out, true, true, true, &nxt, s);
      in = out;
      out = nxt;
      launch_chain_batch(&cs, 1, &in, &out, false, true, true, nullptr, s);
      launch_pcg_tl(d_Gself.d, &G, 1, &d_buf, PCG_VEC_U, s);
    }
  } else {
    launch_shared_unpack(G, 1, d_buf, s);
    launch_backsub(G, 2, s);
    launch_estimate(G, s);
  }
  return SLIDE_OK;
}

// The launch sequence of a phase (0, 1, 2; 3 / 4 = phase 1 before / after the factor + solve) is replayed as a hipGraph while the
// resident graph and the exchange buffer stay the same (every pass of a distributed Gauss-Newton run): ~90 launches per pass otherwise
int HostGraph::launch_phase(int phase, double* d_buf) {
  pred_valid = false; status_clean = false; cache_pose = -1;      // (outside the streaming update: nothing it left behind can be relied on)
  hipStream_t s = stream;
  static const bool env_graph = !(getenv("SLIDE_NO_GRAPH") && getenv("SLIDE_NO_GRAPH")[0] == '1');
  PhaseGraph& pg = phase_graph[phase];
  static const int env_mask = getenv("SLIDE_PHASE_GRAPH_MASK") ? atoi(getenv("SLIDE_PHASE_GRAPH_MASK")) : 31;
  bool use_graph = env_graph && !prof.on && G.T > 4 && ((env_mask >> phase) & 1) && !(batch && phase == 1);   // (the batch spans streams)
  if (use_graph && !(pg.exec && pg.buf == d_buf && std::memcmp(&pg.G, &G, sizeof(GraphDev)) == 0)) {
    if (pg.exec) { (void)hipGraphExecDestroy(pg.exec); pg.exec = nullptr; }
    hipGraph_t graph = nullptr;
    std::lock_guard<std::mutex> cap(g_capture_mtx);
    SL_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_phase(phase, d_buf);
    const hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc != SLIDE_OK || e != hipSuccess || graph == nullptr) {
      (void)hipGetLastError();
      use_graph = false;
    } else {
      const hipError_t ei = hipGraphInstantiate(&pg.exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ei != hipSuccess) { pg.exec = nullptr; (void)hipGetLastError(); use_graph = false; }
      else { pg.G = G; pg.buf = d_buf; }
    }
  }
  if (use_graph) SL_HIP(hipGraphLaunch(pg.exec, s));
  else {
    const int rc = enqueue_phase(phase, d_buf);
    if (rc != SLIDE_OK) return rc;
  }
  return SLIDE_OK;
}

int HostGraph::dist_phase(int phase, double* d_buf) {
  pred_valid = false; status_clean = false; cache_pose = -1;      // (outside the streaming update: nothing it left behind can be relied on)
  hipStream_t s = stream;
  factor_valid = false;      // (the phases move linearisation points and factor into S on their own schedule)
  if (phase >= 0 && phase <= 2) {
    if (phase == 0) {
      int rc = merge_pending();
      if (rc != SLIDE_OK) return rc;
      rc = upload_new();
      if (rc != SLIDE_OK) return rc;
      G.relin_thr = 0.0;
      if (pcg_iters > 0 && (rc = sync_self()) != SLIDE_OK) return rc;
      SL_HIP(hipMemsetAsync(d_status.d, 0, 8 * sizeof(int), s));     // (outside the captured sequence, as in run_update)
    }
    {
      const int rc = launch_phase(phase, d_buf);
      if (rc != SLIDE_OK) return rc;
    }
    if (phase == 2) {
      int st[8];
      SL_HIP(hipMemcpyAsync(st, d_status.d, 8 * sizeof(int), hipMemcpyDeviceToHost, s));
      SL_HIP(hipStreamSynchronize(s));
      SL_HIP(hipGetLastError());
      return decode_status(st);
    }
  } else if (phase >= 31 && phase <= 33) {
    if (!(pcg_iters > 0 && G.n_slots > 0) || batch || !have_self) { g_last_error = "dist_phase 31-33: the joint solve is not set up (set_pcg, phases 0 and 1 first)"; return SLIDE_ERR_INVALID; }
    const int rc = enqueue_phase(phase, d_buf);
    if (rc != SLIDE_OK) return rc;
  } else if (phase == 10) {
    // commit every variable (theta <- theta (+) delta, delta <- 0), then publish the owners' values
    G.relin_thr = 0.0;
    launch_relin(G, s);
    if (G.P) SL_HIP(hipMemsetAsync(G.pose_delta, 0, 6 * (size_t)G.P * sizeof(double), s));
    if (G.L) SL_HIP(hipMemsetAsync(G.lm_delta, 0, 9 * (size_t)G.L * sizeof(double), s));
    launch_shared_pack(G, 2, d_buf, s);
  } else if (phase == 11) {
    launch_shared_unpack(G, 2, d_buf, s);
    launch_estimate(G, s);
  } else if (phase == 20) {
    // current estimates of the owned ghost poses (12 doubles per slot, zeros elsewhere) for an all-reduce(sum)
    int rc = merge_pending();
    if (rc != SLIDE_OK) return rc;
    rc = upload_new();
    if (rc != SLIDE_OK) return rc;
    launch_estimate(G, s);
    launch_ghost_exchange(G, 0, d_buf, s);
  } else if (phase == 21) {
    launch_ghost_exchange(G, 1, d_buf, s);
  } else {
    return SLIDE_ERR_INVALID;
  }
  SL_HIP(hipStreamSynchronize(s));
  SL_HIP(hipGetLastError());
  return SLIDE_OK;
}

// One distributed Gauss-Newton pass of a graph whose batch holds EVERY robot of the job (all on this GPU): phases 0 / 1 / 2 with
// the two exchanges as device-side sums between the batch's buffers — stream-ordered, one host synchronisation at the end.
int HostGraph::dist_pass_local(double* d_buf) {
  pred_valid = false; status_clean = false; cache_pose = -1;      // (outside the streaming update: nothing it left behind can be relied on)
  if (!batch) { g_last_error = "dist_pass_local: the graph is in no batch"; return SLIDE_ERR_INVALID; }
  hipStream_t s = stream;
  int rc = merge_pending();
  if (rc != SLIDE_OK) return rc;
  rc = upload_new();
  if (rc != SLIDE_OK) return rc;
  G.relin_thr = 0.0;
  SL_HIP(hipMemsetAsync(d_status.d, 0, 8 * sizeof(int), s));
  if ((rc = launch_phase(0, d_buf)) != SLIDE_OK) return rc;
  if ((rc = batch->all_reduce(batch_slot, d_buf, 54 * G.n_slots, s)) != SLIDE_OK) return rc;
  if ((rc = launch_phase(3, d_buf)) != SLIDE_OK) return rc;
  if ((rc = factor_and_solve(s)) != SLIDE_OK) return rc;
  if ((rc = launch_phase(4, d_buf)) != SLIDE_OK) return rc;
  if ((rc = batch->all_reduce(batch_slot, d_buf, 9 * G.n_slots, s)) != SLIDE_OK) return rc;
  if ((rc = launch_phase(2, d_buf)) != SLIDE_OK) return rc;
  int st[8];
  SL_HIP(hipMemcpyAsync(st, d_status.d, 8 * sizeof(int), hipMemcpyDeviceToHost, s));
  SL_HIP(hipStreamSynchronize(s));
  SL_HIP(hipGetLastError());
  return decode_status(st);
}

// getPoseCovariance graph.cpp:314-323: marginal covariance of one pose at the linearisation point of the last solve
int HostGraph::pose_covariance(int robot, uint64_t idx, double* cov36) {
  if (!robot_ok(robot)) return SLIDE_ERR_INVALID;
  auto it = key2pose.find(pose_key(robot, idx));
  if (it == key2pose.end() || (size_t)it->second >= up_P) return SLIDE_MISSING;
  if (!factor_valid || G.T == 0) { g_last_error = "pose_covariance: no factorisation yet (call solve first)"; return SLIDE_ERR_INVALID; }
  hipStream_t s = stream;
  const size_t nT = (size_t)G.T * NB;
  if (d_covY.ensure(6 * nT + 36, 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  SL_HIP(hipMemsetAsync(d_covY.d, 0, (6 * nT + 36) * sizeof(double), s));
  const int row0 = 6 * it->second;
  const double one = 1.0;
  for (int c = 0; c < 6; ++c)
    SL_HIP(hipMemcpyAsync(d_covY.d + (size_t)c * nT + row0 + c, &one, sizeof(double), hipMemcpyHostToDevice, s));
  launch_pose_covariance(G.S, G.ld, G.T, G.Ld, G.Winv, d_covY.d, row0, d_covY.d + 6 * nT, s);
  SL_HIP(hipMemcpyAsync(cov36, d_covY.d + 6 * nT, 36 * sizeof(double), hipMemcpyDeviceToHost, s));
  SL_HIP(hipStreamSynchronize(s));
  SL_HIP(hipGetLastError());
  return SLIDE_OK;
}

int HostGraph::solve() {
  int rc = merge_pending();
  if (rc != SLIDE_OK) return rc;
  rc = upload_new();
  if (rc != SLIDE_OK) return rc;
  return run_update(P.relinearize_threshold, 1);
}
int HostGraph::gauss_newton(int iterations) {
  int rc = merge_pending();
  if (rc != SLIDE_OK) return rc;
  rc = upload_new();
  if (rc != SLIDE_OK) return rc;
  return run_update(0.0, iterations);
}

int HostGraph::get_pose12(int robot, uint64_t idx, double* out12) {
  SE3 I{eye3(), V3{0, 0, 0}};
  to12(I, out12);
  if (!robot_ok(robot)) return SLIDE_ERR_INVALID;
  auto it = key2pose.find(pose_key(robot, idx));
  if (it == key2pose.end() || (size_t)it->second >= up_P) return SLIDE_MISSING;
  if (it->second == cache_pose) {      // (the newest key frame right after its update: came back with the status words)
    std::memcpy(out12, cache_pose12, 12 * sizeof(double));
    return SLIDE_OK;
  }
  SL_HIP(hipMemcpyAsync(out12, d_pose_est.d + 12 * (size_t)it->second, 12 * sizeof(double), hipMemcpyDeviceToHost, stream));
  SL_HIP(hipStreamSynchronize(stream));
  return SLIDE_OK;
}
int HostGraph::lm_lid(int cls, uint64_t idx) const {
  auto it = key2lm.find(lm_key(cls, idx));
  if (it == key2lm.end() || (size_t)it->second >= up_L) return -1;
  return it->second;
}
int HostGraph::get_landmark(int cls, uint64_t idx, double* out) {
  const int n = cls == SLIDE_CLS_CYLINDER ? 7 : (cls == SLIDE_CLS_CUBE ? 15 : 3);
  for (int i = 0; i < n; ++i) out[i] = 0.0;
  const int lid = lm_lid(cls, idx);
  if (lid < 0) return SLIDE_MISSING;
  SL_HIP(hipMemcpyAsync(out, d_lm_est.d + 15 * (size_t)lid, n * sizeof(double), hipMemcpyDeviceToHost, stream));
  SL_HIP(hipStreamSynchronize(stream));
  return SLIDE_OK;
}
void HostGraph::stats(int64_t* o) const {
  o[0] = (int64_t)up_P; o[1] = (int64_t)up_L; o[2] = (int64_t)(up_pr + up_bt + up_lf); o[3] = last_relin;
  o[4] = (int64_t)G.T * NB;
}
int64_t HostGraph::rejected() const { return n_rejected; }
// 2 x NonlinearFactorGraph::error at the CURRENT estimate: theta <- theta (+) delta, relinearise, sum r^T r
int HostGraph::chi2(double* out4) {
  pred_valid = false; status_clean = false; cache_pose = -1;      // (outside the streaming update: nothing it left behind can be relied on)
  int rc = merge_pending();
  if (rc != SLIDE_OK) return rc;
  rc = upload_new();
  if (rc != SLIDE_OK) return rc;
  for (int i = 0; i < 4; ++i) out4[i] = 0.0;
  if (G.P == 0) return SLIDE_OK;
  hipStream_t s = stream;
  G.relin_thr = 0.0;
  launch_relin(G, s);                 // theta <- theta (+) delta ...
  SL_HIP(hipMemsetAsync(G.pose_delta, 0, 6 * (size_t)G.P * sizeof(double), s));      // ... and delta <- 0: the next update must not apply it again
  if (G.L) SL_HIP(hipMemsetAsync(G.lm_delta, 0, 9 * (size_t)G.L * sizeof(double), s));
  launch_linearize(G, s);
  launch_estimate(G, s);
  if (d_covY.ensure(8, 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  launch_chi2(G, d_covY.d, s);
  SL_HIP(hipMemcpyAsync(out4, d_covY.d, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  SL_HIP(hipStreamSynchronize(s));
  SL_HIP(hipGetLastError());
  factor_valid = false;
  return SLIDE_OK;
}
int HostGraph::get_tile_profile(int* out, int cap) {
  int rc = merge_pending();
  if (rc == SLIDE_OK) rc = upload_new();
  if (rc != SLIDE_OK) return rc < 0 ? rc : -rc;
  const int T = (int)h_prof.size();
  for (int c = 0; c < T && c < cap; ++c) out[c] = h_prof[c];
  return T;
}
int HostGraph::get_border_profile(int* out, int cap) {
  int rc = merge_pending();
  if (rc == SLIDE_OK) rc = upload_new();
  if (rc != SLIDE_OK) return rc < 0 ? rc : -rc;
  if (!arrow_on()) return 0;
  for (int i = 0; i < nbr && i < cap; ++i) out[i] = h_bfirst[i];
  return nbr;
}
// the per-segment activity table of the border rows (seg_tab, host_graph.hpp): returns its length, 0 when the band is not cut
int HostGraph::get_segment_table(int* out, int cap) {
  int rc = merge_pending();
  if (rc == SLIDE_OK) rc = upload_new();
  if (rc != SLIDE_OK) return rc < 0 ? rc : -rc;
  if (!arrow_on() || nsep <= 0 || seg_tab.empty()) return 0;
  for (int i = 0; i < seg_tab_head && i < cap; ++i) out[i] = seg_tab[i];      // (the per-column masks behind it are the kernels' business)
  return seg_tab_head;
}
int HostGraph::get_segments(int* out, int cap) {
  int rc = merge_pending();
  if (rc == SLIDE_OK) rc = upload_new();
  if (rc != SLIDE_OK) return rc < 0 ? rc : -rc;
  if (!arrow_on() || nsep <= 0) return 1;
  for (size_t i = 0; i < segs.size() && 2 * (int)i + 1 < cap; ++i) { out[2 * i] = segs[i].t0; out[2 * i + 1] = segs[i].t1; }
  if (2 * (int)segs.size() < cap) out[2 * segs.size()] = n_sep_poses;
  return (int)segs.size();
}
void HostGraph::set_dense_profile(bool on) {
  if (force_dense == on) return;
  force_dense = on;
  topo_dirty = true;
}
int HostGraph::pcg_stats(double* out8) {
  for (int i = 0; i < 8; ++i) out8[i] = 0.0;
  if (!d_pcg_scal.d) return SLIDE_OK;
  SL_HIP(hipMemcpyAsync(out8, d_pcg_scal.d, 8 * sizeof(double), hipMemcpyDeviceToHost, stream));
  SL_HIP(hipStreamSynchronize(stream));
  return SLIDE_OK;
}

}  // namespace sl
