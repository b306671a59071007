// Launch wrappers of the HIP kernels (definitions in *_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "graph_dev.hpp"

namespace sl {

// solver_kernels.hip
void init_solver_kernels();   // one-time kernel attributes (call outside stream capture)
void launch_relin(const GraphDev& G, hipStream_t s);
void launch_linearize(const GraphDev& G, hipStream_t s);
void launch_landmark(const GraphDev& G, int mode, hipStream_t s);      // mode: 0 fused, 1 accumulate, 2 finish from sums
void launch_pose(const GraphDev& G, hipStream_t s);
void launch_schur(const GraphDev& G, hipStream_t s);
void launch_backsub(const GraphDev& G, int mode, hipStream_t s);       // mode: 0 fused, 1 t_l only, 2 delta_l from t_l
void launch_shared_pack(const GraphDev& G, int what, double* buf, hipStream_t s);
void launch_shared_unpack(const GraphDev& G, int what, const double* buf, hipStream_t s);
void launch_estimate(const GraphDev& G, hipStream_t s);
void launch_estimate_predict(const GraphDev& G, hipStream_t s, int pose = -1, double* out17 = nullptr);      // out17 (or null): the closing pack of k_final_pack by the last workgroup — 16 doubles + an arrival counter (zero before the launch)
void launch_final_pack(const GraphDev& G, int pose, double* out16, hipStream_t s);      // out16: 8 status ints (then cleared) | pose_est of `pose` (12 doubles)
void launch_chi2(const GraphDev& G, double* out4, hipStream_t s);   // sum of squared whitened residuals at the last linearisation point
void launch_pose_adj(const GraphDev& G, hipStream_t s);        // pose adjacency bitmap of the Schur assembly (topology only)
void launch_scatter(const void* stage, unsigned desc_off, int nseg, hipStream_t s);
void launch_gather(void* stage, unsigned desc_off, int nseg, hipStream_t s);
void launch_gather_args(void* stage, const void* segs, int nseg, hipStream_t s);      // the same, up to 16 descriptors as kernel arguments
void launch_sum_bcast(double* const* bufs, int n, int count, hipStream_t s);   // local all-reduce(sum) of up to 8 buffers
void launch_bcast(double* const* bufs, int n, int count, hipStream_t s);       // bufs[0] -> the others
void launch_phase0_batched(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, hipStream_t s, bool pack = true);   // the per-robot phases 0 / 4 / 2 of a batched pass,
void launch_phase4_batched(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, hipStream_t s);   // blockIdx.z = robot
void launch_phase2_batched(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, hipStream_t s);
void launch_phase3_batched(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, hipStream_t s);   // blockIdx.z = robot               // local all-reduce(sum) of up to 8 buffers             // device arrays -> one staging buffer (DownloadBatch)   // staged upload -> destinations (UploadBatch)
void launch_ghost_exchange(const GraphDev& G, int what, double* buf, hipStream_t s);   // what: 0 pack owned poses, 1 adopt
void launch_ghost_exchange_batched(const GraphDev* d, int n, int n_gslots, int what, double* const* bufs, hipStream_t s);   // all robots of a batched pass in one launch
// exact joint step ("arrow", graph_dev.hpp): border rows + border block of every robot; the separator system of all shared landmarks
// gathered from the robots' border blocks (maps[i]: m ints, global separator coordinate -> robot i's border coordinate or -1); its
// solution handed back
void launch_border_assemble_batched(const GraphDev* d, const GraphDev* h, int n, hipStream_t s);
void launch_sep_extract_batched(const GraphDev* d, const GraphDev* h, int n, const int* n_sep_poses, hipStream_t s);      // separator poses out of the band (k_sep_extract_b)
void launch_sep_pose_scatter_batched(const GraphDev* d, const GraphDev* h, int n, double* const* xloc, hipStream_t s);  // their solution into dp
// Separator system: Ts tile columns of landmark coordinates (ms real) in sys (ld = (Ts + nl + 1) * NB: band, nl border row tiles = the
// coupling rows of the lam "lambda" coordinates of the inter-robot relative-pose factors, right-hand-side tile row), the lambda x lambda
// block + its right-hand-side row in bord (ldb = (nl + 1) * NB); packed: the exchange buffer (lower tile columns of the whole)
struct SepLayout { double* sys; double* bord; double* packed; int Ts, nl, ms, lam; int gap[4]; int hTa, hTL; };      // gap: two ranges [lo, hi) of landmark coordinates no slot uses (padding between the blocks of a dissected layout): unit diagonal; hTa, hTL: tile rows [hTa, hTL) of the tile columns < hTa are structurally zero (leaf b's rows under leaf a's columns) and absent from the packed layout
void launch_ghost_refresh_local(const GraphDev* d, int n, int n_gslots, hipStream_t s);      // ghost poses of a whole pass on one GPU: pack + sum + adopt in one launch
void launch_copy_pairs(const double* const* src, double* const* dst, const int* count, int n, hipStream_t s);      // up to 8 small device-to-device copies in one launch
void launch_sep_gather(const GraphDev* h, int n, const int* const* maps, const SepLayout& Y, bool packed, hipStream_t s, const int* tmask = nullptr,
                       int split_col = -1, unsigned mask_b = 0, double* sys2 = nullptr, int ld2 = 0, double* bord2 = nullptr);      // tmask: Ts + nl ints, bit r = robot r holds a coordinate of the (virtual) tile; split_col ..: per-half partial sums of the top block (k_sep_gather)
void launch_sep_top_add(const SepLayout& Y, int c0, const double* sys2, int ld2, const double* bord2, hipStream_t s);      // tile columns >= c0 of the system (and the lambda block) += the other half's partial
void launch_sep_unpack(const SepLayout& Y, hipStream_t s, int c0 = 0, int c1 = -1, bool to_packed = false);      // tile columns [c0, c1) (virtual: lambda tiles behind the landmarks'); to_packed: the reverse copy
void launch_lam_prepare(const double* bord, int nl, int lam, double* out, hipStream_t s);      // M = -(K22 - L21 L21^T), rhs = -(r2 - L21 z1)
void launch_sep_xloc(int n, const int* const* maps, int ms, int lam, const double* xs, const double* xl, double* const* xloc, hipStream_t s);
void launch_arrow_finish_batched(const GraphDev* d, const GraphDev* h, int n, const double* xs, const int* sep_off, hipStream_t s);
void launch_phase3_arrow_batched(const GraphDev* d, const GraphDev* h, int n, hipStream_t s);   // phase 3 without the exchanged sums: the robots' own H_ll

// chol_kernels.hip — blocked right-looking FP64 Cholesky of the (T*NB)^2 lower matrix S (column-major,
// leading dimension ld = (T+1)*NB; the extra row tile carries the right-hand side), tile edge NB = 64.
// ctr: T + 2 ints, zero before the first factorisation (each step clears the next step's work counter itself)
void launch_chol_step(double* S, int ld, int k, int T, double* Ld, double* Winv, int* status, int* ctr, float* L32, const int* h_prof, hipStream_t s, int nbr = 0);   // L32, h_prof: see CholSystem
void launch_chol_extract_y(const double* S, int ld, int T, double* yv, double* dp, int* status, hipStream_t s, int nbr = 0, int b0 = 0, double* prev = nullptr);   // prev: dp's old content is kept there (bounded back-substitution)   // also clears status[4], the ticket counter of launch_chol_bwd_all
// A quiet-NaN payload no solution value can equal bit for bit: the outputs of the chained substitutions are pre-filled with it and the
// workgroups poll the blocks they depend on ("flag in data").
constexpr unsigned long long CHAIN_SENTINEL = 0x7FF8DEADBEEF0BADull;
struct CholSystem { double* S; int ld, T; double* Ld; double* Winv; double* yv; double* dp; int* status;
                    float* L32;      // packed f32 copy of the factor for the joint solve's preconditioner (null: none), see bwd_chain_body
                    const int* h_prof;           // host: profile of the factor, T ints (plan_step in chol_kernels.hip), or null = dense
                    const int* prof; const int* first;      // device: the same and, per block row, the first block column that reaches it
                    double* ctab;                // 4 * T * 4096 doubles: tables of the chained substitutions in a joint-solve pass (k_chain_tables), or null
                    int nbr;                     // border row tiles between the band and the right-hand-side row (exact joint step: the separator's coupling rows), see b_decode
                    double* bord; int ldb;       // border x border block of the system ((nbr + 1) * NB rows, nbr * NB columns, column-major) — k_border_syrk
                    const double* bord_src;      // or null: the block's content BEFORE the product lives there (same layout) and k_border_syrk WRITES bord = bord_src - W W^T (GraphDev::bord0)
                    const int* bfirst;           // device, nbr + 1 ints: first block column of the band in which border tile row i can be non-zero (non-decreasing)
                    const int* h_bfirst;         // host copy (plan_step: the border rows still all-zero at a block column are skipped) or null: every row always
                    int b0;                      // tile row (in the system's own row numbering) at which its border rows start; 0: right behind the band (= T).  A SEGMENT of
                                                 // a robot's band factored as a system of its own (a view: shifted S, profile) has its border further down: b0 = T_robot - first tile
                    int kofs;                    // the view's first block column in the robot's numbering (bfirst is in that numbering)
                    const int* ord;              // device, nbr ints, or null: the j-th ACTIVE border row of this view is tile row b0 + ord[j] (a segment has its own order
                                                 // of first columns: h_bfirst is then that segment's sorted list)
                    const int* segtab; };        // device or null (border product of a segmented band): nseg, the segments' end columns, per segment the first column of
                                                 // every border tile row + the right-hand side (1 << 30: the row is zero in that segment)
constexpr int CHOL_STEP_BATCH_MAX = 32;      // systems per launch of the batched step kernel (the segments of eight robots' bands in one launch sequence)
void launch_chol_batch(const CholSystem* d, int n, int* ctr, hipStream_t s, hipEvent_t after_steps = nullptr, bool solve = true, int cu_share = 100, int* pair_tickets = nullptr);      // pair_tickets (CHOL_STEP_BATCH_MAX ints, zero, used by no other launch sequence at the same time): two block columns per launch (k_chol_pair_batched) when the systems allow it (chol_pair_supported)
bool chol_pair_supported(const CholSystem* d, int n);            // up to 8 systems, one launch per block column; solve = false: steps + extraction of y only; cu_share: percent of the CUs this launch sequence may count on (sequences running side by side)
void launch_chol_bwd_batch(const CholSystem* d, int n, hipStream_t s);       // yv -> dp of up to 8 factored systems (chained backward substitution)
// Left-looking persistent factorisation (k_chol_ll): ALL block columns of the n systems in ONE launch, the kernel boundary per block
// column replaced by flags between workgroups that start in ticket order.  The plan (task table, flags, device copies of the systems'
// parameters) depends on the systems' pointers, profiles and border tables: rebuild it when any of them changes.  h_ord[i]: host copy of
// d[i].ord (or null).  launch_chol_ll = the steps of launch_chol_batch(.., solve = false): factorisation + extraction of y.
struct CholLLPlan;
CholLLPlan* chol_ll_plan_create(const CholSystem* d, int n, const int* const* h_ord = nullptr, hipStream_t upload_stream = nullptr);
void chol_ll_plan_destroy(CholLLPlan* p);
int chol_ll_plan_tasks(const CholLLPlan* p);
int chol_ll_plan_columns(const CholLLPlan* p);
void launch_chol_ll(const CholLLPlan* p, const CholSystem* d, int n, hipStream_t s, int* trace = nullptr);      // trace: diagnostic, 16 host-pinned ints per task (ll_mark / ll_time) or null
// Exact joint step (the border of the systems = the separator's coupling rows, W^T after the steps):
void launch_border_syrk(const CholSystem* d, int n, hipStream_t s, double* scratch = nullptr, int ks = 1);          // bord(i, j) -= sum_c W^T(i, c) W^T(j, c)^T, i >= j, right-hand-side row included; scratch + ks: split K (one system)
void launch_border_syrk_jobs(const CholSystem* d, int n, const int* jobs, int njobs, int lds_pad, hipStream_t s, double* scratch = nullptr, int ks = 1, int jb_end = -1,
                             const int* ks_sys = nullptr, bool fuse_add = false);      // fuse_add (n = 2, same border): system 0's block += system 1's, each split its own way (ks_sys)      // the same, workgroups in the order of a job table (system << 20 | ib << 10 | jb), lds_pad bytes of idle LDS per workgroup (bounds the residency)
void launch_border_apply(const CholSystem* d, int n, const double* const* xloc, hipStream_t s);   // yv -= W x_loc (x_loc: nbr * NB doubles per system)
// one of the two triangular solves with the finished factors of up to 8 systems on arbitrary vectors (T * NB doubles each): out = L^-1 in
// (fwd) or L^-T in (bwd); the preconditioner of the joint solve (pcg_kernels.hip)
void launch_chain_batch(const CholSystem* d, int n, const double* const* in, double* const* out, bool fwd, bool f32, bool prepared,
                        double* const* next_out, hipStream_t s);   // needs the tables (launch_chain_tables) of this factorisation
void launch_chain_tables(const CholSystem* d, int n, hipStream_t s);
void launch_chol_bwd_all(const double* S, int ld, int T, const double* Ld, const double* Winv, double* yv, double* dp, int* status, const int* prof, hipStream_t s,
                         const double* wf_prev = nullptr, double wf_thr = 0.0, int wf_Tprev = 0, int wf_lim = -1);      // wf_*: iSAM2's wildfire bound (bwd_chain_body): blocks whose inputs changed by < wf_thr keep wf_prev; status[7] / status[3] must be 0
// marginal covariance of the pose whose first tangent row is row0 (Y: 6 * T * NB scratch doubles holding the six unit columns)
void launch_pose_covariance(const double* S, int ld, int T, const double* Ld, const double* Winv, double* Y, int row0, double* cov36,
                            hipStream_t s);
// stand-alone dense SPD solve on device buffers (used by the unit tests and the roofline bench leg)
void launch_chol_solve_bwd(const CholSystem& cs, hipStream_t s);   // yv -> dp after launch_chol_extract_y: one-workgroup substitution for narrow profiles, else the chained kernel
int chol_factor_solve(double* S, int ld, int T, double* Ld, double* Winv, double* yv, double* dp, int* status, int* ctr, hipStream_t s);

// pcg_kernels.hip — joint Gauss-Newton step of the robots of a GPU (and, through the caller's exchanges, of the job): PCG on the
// global reduced pose system with the robots' own factors as preconditioner.  d: device array of the n graphs' views, h: host copy.
enum { PCG_VEC_R = 0, PCG_VEC_U = 1, PCG_VEC_W = 2, PCG_VEC_P = 3, PCG_VEC_S = 4, PCG_VEC_X = 5, PCG_VEC_Y = 6, PCG_VEC_COUNT = 7 };
void launch_status_clear(const GraphDev* d, int n, hipStream_t s, int* x0 = nullptr, int* x1 = nullptr);      // x0 / x1: two more 8-word status blocks cleared in the same launch             // status[0..7] = 0 for every graph of the batch
void launch_status_gather(const GraphDev* d, int n, int* out, hipStream_t s, const int* x0 = nullptr, const int* x1 = nullptr);      // x0 / x1: OR-ed into graph 0's flags first  // out[8 i ..] = graph i's status words
void launch_ints_clear(int* p, int n, hipStream_t s);                          // p[0 .. n-1] = 0 (a kernel node: see launch_status_clear)
void launch_status_or(int* dst, const int* src, int n, hipStream_t s);         // dst[i] |= src[i]
void launch_pcg_init(const GraphDev* d, const GraphDev* h, int n, hipStream_t s);
void launch_pcg_tl(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, int vec, hipStream_t s);      // t_l -> bufs[i][9 slot ..], and w = S0 v in the same launch
void launch_pcg_matvec_dots(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, int nsum, hipStream_t s);     // bufs: summed t_l in, (gamma, delta) partials out (w = S0 u ran in launch_pcg_tl)
void launch_pcg_update(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, int nsum, hipStream_t s);          // bufs: summed (gamma, delta) in
void launch_pcg_finish(const GraphDev* d, const GraphDev* h, int n, hipStream_t s);

// assoc_kernels.hip
struct AssocFrameDev {
  // map of one class, resident in HBM
  const float *cx, *cy, *cz; // n each: first-seen float32 positions, SoA (K-NN gate)
  const double* model;       // cyl: 7 n (root ray radius) ; box: 3 n (xyz)
  const int32_t* label;      // n
  int n;
  int K;
  int gate;                  // 1: K-NN gate; 0: the submap is the whole map in the caller's order (stand-alone matchers)
  int Kp, cached, staged;    // LDS plan of the gate (assoc_plan)
  double thresh;
  double best_init;          // sloam.cpp:90 / :128,136 / :176,180
  int label_gate;            // 0 none (cubes), 1 skip unless equal (ellipsoids), 2 distance = 1000 (cylinders)
  int is_cyl;
  // detections of this frame (body frame) and the pose estimate
  const double* det;         // cyl: 7 per (root ray radius) ; box: pose12 (R t) per detection
  const int32_t* det_label;
  int n_det;
  // outputs
  double* det_world;         // same layout as det
  int32_t* match_sub;        // submap index or -1
  int32_t* match_map;        // map index or -1
  int32_t* submap;           // K entries: map indices nearest first (matchesMap_)
  int32_t* n_sub;            // 1
};
// LDS plan of one class for the K-NN gate: sort buffer length Kp (power of two >= min(K, n)), whether the n distance words are cached
// in LDS, dynamic LDS bytes.  false: min(K, n) exceeds ASSOC_MAX_K.  The cloud itself may be of any size.
bool assoc_plan(int n, int K, int gate, int model_stride, int* Kp, int* cached, int* staged, size_t* bytes);
void launch_assoc_frame(const AssocFrameDev* classes3, size_t lds_bytes /* max over the three classes */, const double* pose12, hipStream_t s);
int launch_assoc_sweep(const float* cx, const float* cy, const float* cz, const double* model_xyz, const int32_t* label, int n_map,
                       const double* query_pos, const double* obs_xyz, const int32_t* obs_label, int n_query,
                       int n_obs, int K, double thresh, int32_t* out_map_idx, hipStream_t s);     // -1: K beyond ASSOC_MAX_K
void launch_map_refresh(double* cyl_model, int n_cyl, const int* cyl_lid, double* cube_xyz, int n_cube, const int* cube_lid,
                        double* ell_xyz, int n_ell, const int* ell_lid, const double* lm_est, hipStream_t s);
constexpr int ASSOC_MAX_K = 16384;                 // neighbours kept by the K-NN gate (LDS sort buffer); the cloud is unbounded
constexpr int ASSOC_LDS_BUDGET = 140 * 1024;      // dynamic part; the select keeps another 16 KB of per-wave histograms (static)

// place_kernels.hip
struct PlaceDev {
  const double* ref7; int nr;
  const double* qry7; int nq;
  const double* xs; const double* ys; const double* yaws;   // candidate lattice values
  const int32_t* cell_x; const int32_t* cell_y;             // per cell indices into xs / ys
  long long n_cells; int n_yaw;
  double thr_pos, thr_dim; int ignore_dim;
  int32_t* inliers;                                         // n_cells * n_yaw
  // round 5 (k_place_sweep_b): both maps bucketed by label on the host (stable: the reference's first-hit order inside a label is kept)
  const double* rxy;       // 2 nr: (x, y) of the reference objects, bucket after bucket; null: the plain kernel
  const double* rdim;      // 3 nr: their dimensions (read when !ignore_dim)
  const double* qxy;       // 2 nq: (x, y) of the query objects, ordered by the bucket of their label
  const double* qdim;      // 3 nq
  const int32_t* qrange;   // 2 nq: [lo, hi) = the reference bucket a query object scans (empty when the reference holds no such label)
  double v_crit;           // smallest double whose correctly rounded square root is >= thr_pos: sqrt(v) < thr_pos <=> v < v_crit
};
void launch_place_sweep(const PlaceDev& P, hipStream_t s);
void launch_place_argmax(const int32_t* inliers, long long n, long long* best_idx, int32_t* best_val, hipStream_t s);
void launch_tri_prepare(const double* tri, int n, double* sdist, double* sxy, hipStream_t s);
void launch_tri_match(bool emit, const double* dm, const double* xm, int ntm, const double* dd, const double* xd, int ntd, double thr,
                      int* counts, const long long* offs, double* pts, double* diffs, hipStream_t s);
// clipper_kernels.hip — CLIPPER dense clique on the device: CSR of the symmetric affinity matrix from its dense upper triangle (count,
// host prefix sum, fill), then the whole projected-gradient solve in one persistent workgroup.  work6n: 6 n doubles (u comes back in the
// first n), out4: {F, d, gradient evaluations, outer iterations}
struct ClqSolve {
  const int* rowptr; const int* col; const double* val; int n;
  const double* u0;               // start weights
  double *u, *unew, *g, *gnew, *Mu, *Cu;      // n each (global; L2-resident)
  double tol_u, tol_F, beta, eps;
  int maxin, maxol, maxls, rescale;
  double* out;                    // [0] F, [1] d, [2] gradient evaluations, [3] outer iterations
};
void launch_clq_solve_batch(const ClqSolve* d_jobs, int n_jobs, hipStream_t s);      // one persistent workgroup per job (grid = n_jobs)
// one large problem on n_wg co-resident workgroups (cooperative launch): A.u (n) receives the result, A.Mu = 4 n doubles with A.Cu = A.Mu + n,
// priv = n_wg x 4 n doubles, bar2 = 2 ints; false when the cooperative launch is refused.  out[2] < 0 afterwards: a grid barrier timed out.
bool launch_clq_solve_coop(const ClqSolve& A, double* priv, int* bar2, int n_wg, hipStream_t s);
void launch_clq_csr_count(const double* Mup, int n, int* rowcnt, hipStream_t s);
void launch_clq_csr_fill(const double* Mup, int n, const int* rowptr, int* col, double* val, hipStream_t s);
void launch_clq_solve(const int* rowptr, const int* col, const double* val, int n, const double* u0, double* work6n, double tol_u, double tol_F,
                      double beta, double eps, int maxin, int maxol, int maxls, int rescale, double* out4, hipStream_t s);
void launch_clipper_affinity(const double* D1, const double* D2, int dim, const int32_t* A, int m, double sigma, double eps,
                             double mindist, double affinityeps, double* M, hipStream_t s);

}  // namespace sl
