// Host side of the device-resident factor graph: the SemanticFactorGraph seam of the reference
// (backend/sloam/include/factorgraph/graph.h:70-121, src/factorgraph/graph.cpp) re-designed for one
// MI355X: factors and variables are appended to SoA arrays in HBM, the host keeps only keys,
// topology (CSR lists) and the pending fgraph / fvalues queues (graph.h:150-151).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/slide_gpu.h"
#include "graph_dev.hpp"
#include "kernels.hpp"
#include "sl_math.hpp"

namespace sl {

extern thread_local std::string g_last_error;
void thread_capture_mode_local();      // (host_graph.hip) once per host thread: stream-capture mode thread-local
bool hip_ok(hipError_t e, const char* what);
int decode_status(const int* st8);      // status words of a pass -> SLIDE_OK / SLIDE_ERR_NOT_SPD / SLIDE_ERR_RUNTIME (+ g_last_error)
#define SL_HIP(x)                                     \
  do {                                                \
    if (!::sl::hip_ok((x), #x)) return SLIDE_ERR_HIP; \
  } while (0)

// Host -> device uploads of one graph update, batched: while a batch is open (thread-local), DevArr::upload copies into one
// pinned staging buffer instead of issuing a hipMemcpyAsync per array (~35 small copies per frame otherwise, each a few
// microseconds of host and copy-engine time); flush() sends the buffer with ONE copy and scatters it on the device.
struct UploadBatch {
  struct Seg { unsigned long long dst; unsigned off, bytes; };
  std::vector<Seg> segs;
  unsigned char* h_pin = nullptr; size_t h_cap = 0, used = 0;
  unsigned char* d_stage = nullptr; size_t d_cap = 0;
  hipEvent_t ev = nullptr;
  bool in_flight = false;
  static thread_local UploadBatch* current;
  ~UploadBatch();
  int begin();                                     // waits for the previous flush's copy before the pinned buffer is reused
  int reserve(size_t need);
  int add(void* dst, const void* src, size_t bytes);
  int flush(hipStream_t s);                        // closes the batch
};

// Device -> host results of one step, batched the same way: gathered on the device, ONE copy into pinned memory, then
// handed out to their host destinations.  run() synchronises the stream.
struct DownloadBatch {
  struct Seg { unsigned long long src; unsigned off, bytes; };
  std::vector<Seg> segs;
  std::vector<void*> host_dst;
  unsigned char* h_pin = nullptr; size_t h_cap = 0, used = 0;
  unsigned char* d_stage = nullptr; size_t d_cap = 0;
  ~DownloadBatch();
  void add(void* host, const void* dev, size_t bytes);     // bytes: multiple of 4
  int run(hipStream_t s);
};

// host mirror of a device CSR (HostGraph: landmark -> factors, pose -> factors, pose -> relative-pose factors)
struct CsrMirror {
  std::vector<int> ptr{0}, val;
  int dirty_from = 0;      // lists from this index on changed since the last upload (lists beyond the mirror are new anyway)
  size_t off0 = 0;         // (set by the upload) first value entry it rewrote
  void touch(int i) { if (i < dirty_from) dirty_from = i; }
};
// growable device array; contents below `used` survive a growth
template <class T>
struct DevArr {
  T* d = nullptr;
  size_t cap = 0;
  ~DevArr() { if (d) (void)hipFree(d); }
  int ensure(size_t n, size_t used, hipStream_t s, bool zero_new = false) {
    if (n <= cap) return SLIDE_OK;
    size_t nc = cap ? cap : 1024;
    while (nc < n) nc *= 2;
    T* nd = nullptr;
    SL_HIP(hipMalloc(&nd, nc * sizeof(T)));
    if (zero_new) SL_HIP(hipMemsetAsync(nd, 0, nc * sizeof(T), s));
    if (d && used) SL_HIP(hipMemcpyAsync(nd, d, used * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (d) {
      SL_HIP(hipStreamSynchronize(s));
      SL_HIP(hipFree(d));
    }
    d = nd;
    cap = nc;
    return SLIDE_OK;
  }
  // exactly n elements (the big per-system buffers: the caller's own growth policy, no rounding up on top of it); contents dropped
  int ensure_exact(size_t n, hipStream_t s, bool zero_new = false) {
    if (d) {
      SL_HIP(hipStreamSynchronize(s));
      SL_HIP(hipFree(d));
      d = nullptr;
      cap = 0;
    }
    if (n == 0) return SLIDE_OK;
    SL_HIP(hipMalloc(&d, n * sizeof(T)));
    cap = n;
    if (zero_new) SL_HIP(hipMemsetAsync(d, 0, n * sizeof(T), s));
    return SLIDE_OK;
  }
  int upload(const T* h, size_t off, size_t count, hipStream_t s) {
    if (count == 0) return SLIDE_OK;
    if (UploadBatch::current && sizeof(T) % 4 == 0) return UploadBatch::current->add(d + off, h, count * sizeof(T));
    SL_HIP(hipMemcpyAsync(d + off, h, count * sizeof(T), hipMemcpyHostToDevice, s));
    return SLIDE_OK;
  }
};

struct Profiler {
  bool on = false;
  struct Rec { int id; hipEvent_t a, b; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  std::vector<std::string> names;
  std::vector<double> ms;
  std::vector<int64_t> count;
  int id_of(const char* name);
  void begin(int id, hipStream_t s);
  void end(hipStream_t s);
  void collect();     // after a stream sync
  void reset();
  ~Profiler();
};

struct PendVar {
  uint64_t key;
  int type;
  double val[15];
};
constexpr int PF_GHOST = 100;
struct PendFac {
  int type;          // 0 prior, 1 between, FT_BR, FT_CUBE, FT_CYL, PF_GHOST (ghost between: k1 = slot, z[12] = local-first flag)
  uint64_t k0, k1;
  double z[15];
  double sigma[9];
};

// The dense factor + solve of several graphs that share a GPU as ONE launch sequence (launch_chol_batch): every graph's thread
// arrives with its system assembled on its own stream, the last one enqueues the batched launches on the batch's stream behind
// all of them, and every graph's stream continues behind the batch.  All joined graphs must solve in lockstep (the distributed
// pass does); a rendezvous that is not completed within 60 s returns SLIDE_ERR_RUNTIME.
class HostGraph;
class CholBatch {
 public:
  explicit CholBatch(int n);
  ~CholBatch();
  int n_slots() const { return n; }
  int factor_solve(int slot, const GraphDev& G, hipStream_t s);
  int all_reduce(int slot, double* d_buf, int count, hipStream_t s);     // sum over the joined graphs' buffers, stream-ordered
  // Ownership: a batch does not own its graphs and a graph does not own its batch.  ~HostGraph leaves its batch (detach), ~CholBatch
  // sends every joined graph back to its own launches; either invalidates the captured pass.  Lock order: pass_mtx -> graph -> mtx.
  void set_graph(int slot, HostGraph* g);
  void detach(HostGraph* g);
  // One distributed pass of ALL joined graphs from one host thread: the phases of every robot on its own stream, forked from and
  // joined to the batch's stream around the two device-side exchanges and the batched factor + solve, captured once and replayed
  // as ONE hipGraph per pass.  bufs[i]: exchange buffer of the graph in slot i.
  int pass_all(double* const* d_bufs);
  int pass_part(double* const* d_bufs, int part);       // the same pass cut at its exchanges (multi-GPU jobs): parts 0, 1, 2 and, with the joint solve, 10, 11, 12
  // Joint solve: after the factorisations, `iters` PCG iterations on the global reduced system (pcg_kernels.hip); 0 = each robot's own
  // block solve only (block-Jacobi over robots).  Changing it invalidates the captured launch sequences.
  void set_pcg(int iters, double tol = 0.0);
  int pcg() const { return pcg_iters; }
  double pcg_tolerance() const { return pcg_tol; }
  // Exact joint step ("arrow"): the shared landmarks stay as the separator of the joint graph (graph_dev.hpp).  sep_buf: the caller's
  // device buffer of sep_buffer_len(m) doubles in which part 0 of a cut pass leaves this GPU's partial sum of the separator system
  // (packed: lower tile columns only) for the cross-GPU all-reduce, and from which part 2 takes the sum; null: no exchange (the job is
  // this process alone).  Changing it invalidates the captured launch sequences.
  int set_arrow(bool on, double* sep_buf, long long sep_len);
  int set_separator_profile(const int32_t* prof, int n);
  int set_separator_blocks(int Ta, int Tb, int used_a, int used_b);      // nested dissection of the separator system: two leaf blocks + top block (zeros: off)
  // A job that spans GPUs, its ranks split in two halves along the dissection: this rank OWNS leaf `leaf` (0 / 1; -1: none, the default) —
  // it factors that leaf only, and of the separator system only the top block crosses between the halves (cut pass: [20] 0 1 2).
  // leader: the one rank of its half that adds the leaf's Schur complement to the top block's sum.
  int set_separator_owner(int leaf, bool leader);
  static void sep_segment(int ms, int lam, int Ta, int Tb, int which, long long* off, long long* len);      // packed buffer: leaf a | leaf b | top block + lambdas
  // Nested dissection of every robot's own pose chain in exact joint passes: `n` segments per robot factored side by side, the windows
  // of poses between them (as wide as the band) eliminated at a second level (graph_dev.hpp pose_sep).  1: off.
  void set_segments(int n_seg);
  int segments() const { return n_seg; }      // tile profile of the separator system's landmark part (n = its tile columns), or none: dense
  bool is_arrow() const { return arrow; }
  // packed exchange layout of the separator system: ms landmark coordinates + lam lambda coordinates (6 per inter-robot relative-pose factor)
  // doubles of it a cut pass exchanges: a dissected layout (Ta, Tb tile columns in its leaves) leaves the zero block between the leaves out
  static long long sep_exchange_len(int ms, int lam, int Ta, int Tb) { return sep_buffer_len(ms, lam) - (long long)NB * NB * Ta * Tb; }
  static long long sep_buffer_len(int ms, int lam = 0) { const long long Tt = (ms + NB - 1) / NB + (lam + NB - 1) / NB; return (long long)NB * NB * Tt * (Tt + 3) / 2; }
  hipStream_t pass_stream();                             // the stream the passes run on (created on first use)
  int profile_pass(double* const* d_bufs, double* ms_steps, int* n_launches);
  int profile_arrow(double* const* d_bufs, double* out6, int* n_sep_steps);

 private:
  int n;
  std::mutex mtx;                        // rendezvous state + the graphs[] table
  std::mutex pass_mtx;                   // pass_all / profile_pass (the captured pass and its device tables)
  bool pass_dirty = false;               // graphs[] changed since the pass was captured (under mtx)
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long generation = 0;
  int gen_status = SLIDE_OK;
  std::vector<CholSystem> sys;
  std::vector<double*> bufs;
  std::vector<HostGraph*> graphs;
  hipGraphExec_t pass_exec = nullptr;
  hipGraphExec_t part_exec[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};     // parts 0, 1, 2, 10, 11, 12 (exact joint step: 0 and 2 only), 20 (ghost refresh)
  int enqueue_ghost_refresh(double* const* d_bufs, int part);
  int last_part = -1;                    // the part of a cut pass that ran last (-1: none / a whole pass): pass_part checks the order
  int pcg_iters = 0;
  double pcg_tol = 0.0;
  bool arrow = false;
  int n_seg = 1;
  std::vector<CholSystem> seg_sys, l2_sys;          // the segments of all joined graphs as systems of their own (views); the second-level systems
  std::vector<int> l2_graph;                        // l2_sys[i] belongs to graphs[l2_graph[i]]
  int* d_ctr2 = nullptr;
  int* d_pair_tickets = nullptr;                    // 12 x CHOL_STEP_BATCH_MAX ints (zero between launches): the pair kernel's per-system counters — one block per launch sequence
                                                    // of the segments (8), the second level (8), the leaves (9), the own leaf (10), the top block (11)
  int* pair_tickets(int block) const { return d_pair_tickets ? d_pair_tickets + block * CHOL_STEP_BATCH_MAX : nullptr; }
  // left-looking persistent factorisations (k_chol_ll): one launch per level instead of one per block column — the segments of all
  // joined graphs, the bands' second level, the separator's leaves (both / the own one of a rank that owns a leaf), its top block.
  // SLIDE_CHOL_LL=0: the step kernels, one launch per block column (round 3's path)
  CholLLPlan *ll_seg = nullptr, *ll_l2 = nullptr, *ll_leaves = nullptr, *ll_leaf_own[2] = {nullptr, nullptr}, *ll_top = nullptr;
  std::vector<const int*> seg_hord;                 // host copies of the segments' border-row orders (CholSystem::ord)
  void free_ll_band_plans();
  void free_ll_sep_plans();
  void sep_leaf_systems(CholSystem* lv) const;      // the two leaf blocks of a dissected separator as systems (views of sepS)
  CholSystem sep_top_system() const;                // its top block (the lambdas' rows as border)
  int* d_syrk_jobs = nullptr;                       // border product: (system << 20 | ib << 10 | jb) of every lower tile + right-hand-side row, longest sum first
  int n_syrk_jobs = 0, syrk_jobs_cap = 0, syrk_lds_pad = 0;
  int* d_l2_jobs = nullptr; int n_l2_jobs = 0, l2_jobs_cap = 0;      // the same for the border products of the bands' second level
  // separator system of the exact joint step: m coordinates, Ts tile columns; factored by the un-batched step kernels on the pass's stream
  double* sepS = nullptr; long long sep_len = 0;          // the system in the factorisation's layout (owned)
  double* sep_x = nullptr; long long sep_x_len = 0;       // the caller's exchange buffer (packed layout), or null: no exchange
  int sep_m = 0, sep_Ts = 0;
  double *sep_Ld = nullptr, *sep_Winv = nullptr, *sep_yv = nullptr, *sep_dp = nullptr;
  int *sep_status = nullptr, *sep_ctr = nullptr, *d_sep_off = nullptr;
  int sep_cap = 0;
  std::vector<int> h_sep_prof; int* d_sep_prof = nullptr; bool sep_prof_on = false;
  // nested dissection of the separator system (set_separator_blocks): two leaf blocks of sep_leafT[0], sep_leafT[1] tile columns that no
  // robot couples, factored side by side as views of sepS with the top block's rows (and the lambda rows) as their border, then the top
  // block; sep_used[b]: coordinates of leaf b that carry a slot (the rest of its last tile is padding with a unit diagonal)
  int sep_leafT[2] = {0, 0}, sep_used[2] = {0, 0};
  int sep_owner = -1; bool sep_leader = true;
  std::vector<int> h_leaf_prof[2];                  // the leaves' own profiles (relative to the view)
  int* d_leaf_prof = nullptr;                       // both, one after the other (backward substitutions of the views)
  int* d_sep_tmask = nullptr;                       // per (virtual) tile of the separator: which joined graphs hold coordinates of it (k_sep_gather)
  int* sep_ctr2 = nullptr; int* d_sep_jobs = nullptr; int n_sep_jobs = 0; double* sep_scratch = nullptr; int sep_ks = 1;
  // a WHOLE pass over a dissected separator takes the same arithmetic as two ranks owning a leaf each: per-half partial sums of the top
  // block (sepS's own top block: the first half's, sep_top2 / sep_bord2: the other's), each minus its leaf's Schur complement, added —
  // so that 1, 2, 4 and 8 ranks give the same bits (SURVEY 7 hard part 5).  sep_mask_b: the joined graphs that hold leaf b.
  double *sep_top2 = nullptr, *sep_bord2 = nullptr; int sep_ld2 = 0; unsigned sep_mask_b = 0;
  int* d_lam_jobs2 = nullptr; int n_lam_jobs2 = 0;      // the same for the lambda block
  int* d_sep_jobs2 = nullptr; int n_sep_jobs2 = 0;      // the top block's product over BOTH leaves as two systems of one launch (system << 20 | ib << 10 | jb)
  bool sep_dissected() const { return sep_leafT[0] > 0 && sep_leafT[1] > 0; }
  // lambda coordinates of the inter-robot relative-pose factors: border rows of the separator system, their own small system
  int sep_lam = 0, sep_nl = 0, lam_cap = -1;
  double *lam_scratch = nullptr;      // partial products of the separator's own border product (split K)
  double *sep_bord = nullptr, *lamS = nullptr, *lam_Ld = nullptr, *lam_Winv = nullptr, *lam_yv = nullptr, *lam_dp = nullptr;
  int *lam_status = nullptr, *lam_ctr = nullptr;
  SepLayout sep_layout() const {
    SepLayout Y{sepS, sep_bord, sep_x, sep_Ts, sep_nl, sep_m, sep_lam, {0, 0, 0, 0}, 0, 0};
    if (sep_dissected()) {
      Y.gap[0] = sep_used[0]; Y.gap[1] = sep_leafT[0] * NB;
      Y.gap[2] = sep_leafT[0] * NB + sep_used[1]; Y.gap[3] = (sep_leafT[0] + sep_leafT[1]) * NB;
      Y.hTa = sep_leafT[0]; Y.hTL = sep_leafT[0] + sep_leafT[1];
    }
    return Y;
  }
  int prepare_separator();
  int enqueue_arrow(double* const* d_bufs, int part, hipEvent_t e0, hipEvent_t e1);
  hipEvent_t prof_ev[7] = {};       // (profile_arrow: [6] = behind the robots' border product, before the bands' second level)
  void free_separator();
  int enqueue_pcg_head(double* const* d_bufs);                   // r = b, u = M^-1 b, t_l(u) packed + local sum
  int enqueue_pcg_mid(double* const* d_bufs, bool whole);                    // w = S u, partial dots + local sum
  int enqueue_pcg_tail(double* const* d_bufs, bool last, bool whole);        // alpha, beta, updates; then u = M^-1 r, t_l(u) | dp = x
  int save_systems();                                            // S -> S0 before the factorisation (joint solve only)
  std::vector<GraphDev> pass_G;
  std::vector<double*> pass_bufs;
  hipEvent_t ev_fork = nullptr;
  GraphDev* d_Gs = nullptr;              // the joined graphs' device views, for the kernels batched over blockIdx.z
  int capture_pass(double* const* d_bufs, int part, hipGraphExec_t* exec);
  int prepare_pass();
  int begin_pass(double* const* d_bufs, bool* same);
  int end_pass();
  int enqueue_pass(double* const* d_bufs, hipEvent_t e0, hipEvent_t e1, int part);
  std::vector<GraphDev> hG;
  int rendezvous(int slot, hipStream_t s, bool reduce, int count);
  std::vector<hipEvent_t> ev_in;
  hipEvent_t ev_out = nullptr;
  hipStream_t master = nullptr;
  hipStream_t aux[8] = {};               // further streams of the grouped factorisation (the groups' launches overlap on the GPU)
  hipEvent_t ev_aux0 = nullptr, ev_aux1[8] = {};
  int* d_ctr = nullptr;
  int last_groups = 1;                   // launch sequences factor_all used last
  int* d_status_all = nullptr;           // the joined graphs' status words, gathered by the last node of a pass
  int ctr_cap = 0;
  int factor_all(hipEvent_t after);      // the batched factor + solve of all joined systems, in one to four overlapping launch sequences
};

class HostGraph {
  friend class CholBatch;
 public:
  explicit HostGraph(const slide_params_t& p);
  ~HostGraph();
  int init();

  // SemanticFactorGraph API (graph.cpp)
  int set_prior(int robot, const double* pose7);
  int add_keypose_between(int robot, uint64_t from, uint64_t to, const double* rel7, const double* est7);
  int add_between_sigma(uint64_t k0, uint64_t k1, const SE3& rel, const double* sigma6);
  int add_loop_closure(const double* rel7, uint64_t i1, int r1, uint64_t i2, int r2);
  int add_relative_meas(const double* rel7, uint64_t i1, int r1, uint64_t i2, int r2);
  int add_relative_meas_ghost(const double* rel7, uint64_t idx, int robot, int slot, bool local_first);
  int set_ghosts(const int32_t* own_robot, const int64_t* own_idx, int n_slots);
  int pose_covariance(int robot, uint64_t idx, double* cov36);
  void join_batch(CholBatch* b, int slot);      // takes the graph's lock itself (never while the batch's is held the other way round)
  int dist_pass_local(double* d_buf);     // one distributed pass when every robot of the job is in this graph's batch (no host syncs inside)
  int add_point_landmark(uint64_t idx, const double* xyz);
  int add_range_bearing(int robot, uint64_t pose_idx, uint64_t lm_idx, const double* bearing, double range);
  int add_cube(int robot, uint64_t pose_idx, uint64_t cube_idx, const SE3& pose, const SE3& cube_world, const double* scale,
               bool exists);
  int add_cylinder(int robot, uint64_t pose_idx, uint64_t cyl_idx, const SE3& pose, const double* root, const double* ray,
                   double radius, bool exists);
  int solve();                       // one iSAM2-equivalent update
  int gauss_newton(int iterations);  // batch GN iterations (threshold 0)
  int get_pose12(int robot, uint64_t idx, double* out12);   // SLIDE_MISSING + identity when absent
  int get_landmark(int cls, uint64_t idx, double* out);
  int lm_lid(int cls, uint64_t idx) const;                   // -1 when not merged yet
  // one-robot-per-GPU mode: shared-landmark slots + the three phases of a distributed GN pass
  int set_shared(const int32_t* cls, const int64_t* idx, const int32_t* owner, int n_slots);
  int dist_phase(int phase, double* d_buf);
  int pcg() const { return pcg_iters; }
  double pcg_tolerance() const { return pcg_tol; }
  void set_pcg(int iters, double tol = 0.0);      // un-batched passes: dist_phase 1 prepares the joint solve, 31 / 32 / 33 run it (0 = block solves only)
  // exact joint step (batched passes only): offsets of the shared slots' tangent coordinates in the separator system, n_slots + 1 ints,
  // the same on every rank (slide_graph_set_separator)
  int set_separator(const int32_t* off, int n);
  // exact joint step: ids[i] = index of this graph's i-th ghost factor in the job's list of n_total inter-robot relative-pose
  // measurements: the factor enters through six separator coordinates of its own ("lambda", graph_dev.hpp gh_bord)
  int set_ghost_ids(const int32_t* ids, int n, int n_total);
  void stats(int64_t* out5) const;
  int64_t rejected() const;
  int chi2(double* out4);                 // sum of squared whitened residuals at the current estimate: total, priors, betweens, landmark factors
  void set_incremental(bool on) { inc_enabled = on; }
  void incremental_stats(int64_t* out4) const { out4[0] = n_inc; out4[1] = n_full; out4[2] = last_cd; out4[3] = G.T; }
  // iSAM2's wildfire threshold on the back-substitution of a streaming update (ISAM2GaussNewtonParams::wildfireThreshold, 1e-3 in the
  // reference's GTSAM 4.0.3 build; graph.cpp:15-18, 260-272).  0 (the default here): every update solves the linear system exactly.
  void set_wildfire(double thr) { wildfire_thr = thr > 0.0 ? thr : 0.0; }
  void wildfire_stats(int64_t* out2) const { out2[0] = n_wf_kept; out2[1] = last_wf_kept; }      // blocks kept: all updates / the last one
  int get_tile_profile(int* out, int cap);
  int get_border_profile(int* out, int cap);
  int get_segments(int* out, int cap);
  int get_segment_table(int* out, int cap);          // number of segments of the band in exact joint passes (1: not cut); out[2 i], out[2 i + 1] = tile range of segment i; out[2 n] = separator poses   // nbr (>= 0) or a negative error; out[i] = first block column of border tile row i, i < min(nbr, cap)     // T (>= 0) or a negative error; out[c] = prof[c] for c < min(T, cap)
  void set_dense_profile(bool on);        // ignore the structure of the reduced system (measurement aid)
  int pcg_stats(double* out8);            // scalars of the last joint solve: gamma_old, alpha_old, alpha, beta, first gamma, last gamma               // entries merge_pending refused since creation

  static uint64_t pose_key(int robot, uint64_t idx);
  static uint64_t lm_key(int cls, uint64_t idx);

  slide_params_t P;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;              // trailing updates of the Cholesky look-ahead
  std::vector<hipEvent_t> ev_dp, ev_upd;
  std::mutex mtx;
  Profiler prof;
  GraphDev G{};

 private:
  int merge_pending();
  int64_t n_rejected = 0;                 // factors / variables refused by merge_pending since creation (slide_graph_stats)
  int upload_new();
  int run_update(double relin_thr, int iterations);
  int enqueue_iteration(bool lookahead, bool skip_relin = false, int c_d = 0, int wf_cd = -1);      // one GN / iSAM2-equivalent pass on `stream`; c_d > 0: block columns below it keep their factor
  // Incremental re-factorisation (iSAM2's "re-eliminate only the affected top of the tree", graph.cpp:260-272): the lowest pose whose
  // rows of the reduced system the factors merged since the last solve change; the generation of S the resident factor belongs to
  int dirty_min_pose = 1 << 30;
  std::vector<int> h_lm_first;          // per landmark: lowest pose index observing it
  DevArr<int> d_lm_first;
  unsigned long long S_gen = 0, fact_gen = ~0ull;
  int64_t n_inc = 0, n_full = 0, last_cd = 0;
  double wildfire_thr = 0.0;
  DevArr<double> d_dp_prev;            // the last solve's reduced solution (bounded back-substitution: k_chol_extract_y keeps it)
  int wf_T = 0;                        // block columns of it that are valid (0: none — first solve, or the buffers were re-allocated)
  // streaming updates without a read-back in the middle (round 5): the last update predicted which variables THIS one relinearises
  // (k_estimate_predict -> status[5]), its closing read-back brought the status words and the newest pose's estimate in one copy
  unsigned long long lin_gen = 1, lin_solved_gen = 0;      // re-allocations of the linearisation buffers (their contents are dropped) / the value at the last successful solve
  bool pred_valid = false;             // pred_pose / pred_thr describe the delta the device holds right now
  int pred_pose = 1 << 30;             // lowest pose whose blocks the next relinearisation changes (1 << 30: none)
  double pred_thr = -1.0;
  bool status_clean = false;           // the status words are zero (k_final_pack left them so)
  DevArr<double> d_final;              // 16 doubles: k_final_pack's output
  int cache_pose = -1;                 // pose index whose estimate cache_pose12 holds (-1: none)
  double cache_pose12[12];
  int64_t n_pred_used = 0;
  int64_t n_wf_kept = 0, last_wf_kept = 0;
  bool inc_enabled = true;

  std::vector<PendVar> pend_vars;
  std::vector<PendFac> pend_facs;
  std::unordered_map<uint64_t, int> key2pose, key2lm;
  // merged host mirrors (initial values; theta itself lives in HBM)
  std::vector<double> h_pose_val, h_lm_val;
  std::vector<int> h_lm_type;
  std::vector<int> h_pr_pose; std::vector<double> h_pr_z, h_pr_sigma;
  std::vector<int> h_bt_i, h_bt_j; std::vector<double> h_bt_z, h_bt_sigma;
  std::vector<int> h_gh_pose, h_gh_slot, h_gh_first; std::vector<double> h_gh_z, h_gh_sigma;   // ghost-between factors
  std::vector<int> h_gslot_pose;
  std::vector<int> h_lf_type, h_lf_pose, h_lf_lm, h_lf_slot;
  std::vector<int64_t> h_lf_joff, h_lf_eoff;
  std::vector<double> h_br_z, h_cu_z, h_cu_sigma, h_cy_z;
  int64_t jbuf_used = 0, ebuf_used = 0;
  std::vector<std::vector<int>> lm_fids, pose_fids, pose_bt;
  // host mirrors of their device CSR forms.  A streaming update touches the lists of the newest pose and of the few landmarks it
  // observes (the youngest ones: landmark ids grow with the first observation), so only the tail from the lowest touched list on is
  // rebuilt and uploaded (round 5: the three full rebuilds + the two per-factor tables were ~0.1 ms of host work and 350 KB of
  // upload per frame at 625 poses, growing with the graph).
  CsrMirror csr_lm, csr_pose, csr_bt;
  double* ei_final_out = nullptr; int ei_final_pose = -1; bool ei_final_done = false;      // run_update -> enqueue_iteration: the closing pack fused into k_estimate_predict
  std::vector<int> hc_pose_lms;           // per entry of the pose CSR: the factor's landmark
  std::vector<long long> hc_pose_ed;      // ... and its E record (offset << 4 | tangent dimension)
  int lm_first_from = 0;                  // h_lm_first changed from this landmark on since the last upload
  std::vector<int> h_lm_last, h_reach;    // per landmark the last observing pose; per pose the last pose it couples to (the profile's input)
  size_t up_P = 0, up_L = 0, up_pr = 0, up_bt = 0, up_lf = 0, up_br = 0, up_cu = 0, up_cy = 0;
  int last_relin = 0;
  bool topo_dirty = true, uploaded_once = false;      // upload_new has work only after merge_pending consumed something

  DevArr<double> d_pose_val, d_pose_delta, d_pose_est, d_lm_val, d_lm_delta, d_lm_est;
  DevArr<int> d_lm_type;
  DevArr<int> d_pr_pose; DevArr<double> d_pr_z, d_pr_sigma, d_pr_r;
  DevArr<int> d_bt_i, d_bt_j; DevArr<double> d_bt_z, d_bt_sigma, d_bt_r, d_bt_J0;
  DevArr<int> d_gh_pose, d_gh_slot, d_gh_first, d_gslot_pose; DevArr<double> d_gh_z, d_gh_sigma, d_gh_r, d_gh_J, d_ghost_val;
  size_t up_gh = 0;
  DevArr<int> d_lf_type, d_lf_pose, d_lf_lm, d_lf_slot;
  DevArr<int64_t> d_lf_joff, d_lf_eoff;
  DevArr<long long> d_pose_ed, d_sp_pairs;
  DevArr<int> d_sp_idx;                   // pair lists of the Schur assembly (GraphDev::sp_idx), rebuilt by upload_new for graphs in an exact joint batch
  int build_schur_pairs(hipStream_t s);
  DevArr<unsigned> d_pose_adj;
  DevArr<double> d_br_z, d_cu_z, d_cu_sigma, d_cy_z, d_jbuf, d_ebuf;
  DevArr<int> d_lm_ptr, d_lm_fids, d_pose_ptr, d_pose_fids, d_pose_lms, d_pose_bt_ptr, d_pose_bt;
  DevArr<double> d_lm_Hinv, d_lm_g, d_pose_H, d_pose_g, d_lm_Hacc, d_lm_t;
  DevArr<int> d_sh_lid, d_sh_owner;
  std::vector<int> h_sh_lid, h_sh_owner;
  DevArr<double> d_S, d_Ld, d_Winv, d_yv, d_dp;
  DevArr<double> d_S0, d_pcg, d_pcg_scal;              // joint solve (pcg_kernels.hip)
  DevArr<int> d_lm_slot;                               // landmark -> shared slot or -1
  int sync_lm_slot();                                  // rebuilds it from h_sh_lid (upload_new, set_shared)
  DevArr<int> d_prof, d_first;                         // tile-level profile of the reduced system (graph_dev.hpp), host copies h_prof / h_first
  std::vector<int> h_prof, h_first;
  int prof_ver = 0;
  int joint_Tcap = 0;                                   // Tcap the joint-solve buffers (S0, L32, ctab) are allocated for; 0 = not allocated
  bool force_dense = getenv("SLIDE_CHOL_DENSE") && getenv("SLIDE_CHOL_DENSE")[0] == '1';
  DevArr<double> d_ctab;                               // explicit inverses of the diagonal blocks (k_chain_tables)
  DevArr<float> d_L32;                                 // packed f32 copy of the factor: the joint solve's preconditioner streams this
  DevArr<GraphDev> d_Gself;                             // this graph's view on the device, for the kernels that take an array of views
  GraphDev G_self{};
  bool have_self = false;
  int pcg_iters = 0;
  double pcg_tol = 0.0;
  // exact joint step: this robot's border = its shared landmarks in slot order
  std::vector<int> h_sep_off;                          // global offsets (n_slots + 1) or empty
  std::vector<int> h_lm_bord, h_sep_map, h_bfirst;     // landmark -> border offset; global separator coordinate -> border coordinate; first column block per border tile row
  // nested dissection of the own pose chain (exact joint passes; CholBatch::set_segments)
  struct Seg { int t0, t1; };                          // tile columns [t0, t1) of a segment
  std::vector<Seg> segs;
  std::vector<int> h_pose_sep;                         // pose -> border offset of a separator pose, or -1
  std::vector<std::vector<int>> seg_prof;              // per segment: its profile in its own numbering (host; plan_step)
  DevArr<int> d_pose_sep, d_seg_prof;
  std::vector<size_t> seg_prof_off;
  // per segment: which border tile rows are non-zero in it and from which block column on (a shared landmark seen only from the first
  // half of the trajectory has no coupling to the second segment's poses; the poses of a cut couple to the few block columns before it and
  // to the segment behind it).  seg_ord[s]: the border tile rows in the order of that first column (the rows still all-zero at a block
  // column are then a suffix again, per segment); seg_sfirst[s]: the first columns in that order (plan_step); seg_tab: what the border
  // product reads — nseg, the segments' last block columns + 1, then per segment the first column of every tile row + the right-hand side
  std::vector<std::vector<int>> seg_ord, seg_sfirst;
  std::vector<int> seg_tab;
  int seg_tab_head = 0;                               // ints of seg_tab in front of the per-column activity masks
  DevArr<int> d_seg_ord, d_seg_tab;
  int nsep = 0, nsep_dim = 0, n_sep_poses = 0;
  DevArr<double> d_Ld2, d_Winv2, d_yv2, d_dp2;
  std::vector<int> h_gh_gid, h_gh_bord;                // ghost factor -> index in the job's relative-pose list; -> border offset of its lambda coordinates
  int lam_total = 0;
  DevArr<int> d_lm_bord, d_sep_map, d_bfirst, d_gh_bord;
  DevArr<double> d_bord, d_bord0, d_xloc;
  int nbr = 0, nbr_alloc = -1, arrow_T = -1;
  bool arrow_on() const;                               // the batch runs exact joint passes and this graph has shared slots + separator offsets
  int sync_self();
  DevArr<int> d_cctr;
  DevArr<double> d_covY;
  bool factor_valid = false;            // S / Ld / Winv hold the factor of the system of the last solve
  DevArr<int> d_status;
  UploadBatch ub;
  CholBatch* batch = nullptr;           // shared factor + solve with the other graphs of this GPU (not owned)
  int batch_slot = 0;
  int factor_and_solve(hipStream_t s);  // the Cholesky part of a pass: own launches, or the batch's
  int Tcap = 0;
  // hipGraph of one pass, captured when the same resident graph is solved repeatedly (kernel arguments are
  // baked in, so any change of counts / pointers / threshold invalidates it)
  hipGraphExec_t gexec = nullptr;
  GraphDev G_cap{};
  GraphDev G_prev{};
  struct PhaseGraph { hipGraphExec_t exec = nullptr; GraphDev G{}; double* buf = nullptr; };
  PhaseGraph phase_graph[5];             // captured launch sequences of dist_phase 0 / 1 / 2 and of phase 1's halves (3, 4)
  int enqueue_phase(int phase, double* d_buf);
  int launch_phase(int phase, double* d_buf);      // hipGraph replay of enqueue_phase when nothing changed
  bool have_prev = false;
};

}  // namespace sl
