// Device-resident factor-graph state (HBM, SoA per record kind) shared by the host graph object and
// the HIP kernels.  One Gauss-Newton / iSAM2-equivalent update of the reference
// (SemanticFactorGraph::solve, backend/sloam/src/factorgraph/graph.cpp:260-272) runs as:
//   relin -> linearise -> landmark reduce (H_ll, H_ll^-1, E, F, u) -> pose reduce (H_pp, g_p)
//   -> Schur assemble (dense reduced pose system, column-major lower) -> blocked FP64-MFMA Cholesky
//   (forward substitution rides along as an extra row) -> backward substitution -> landmark
//   back-substitution -> estimate = theta (+) delta.
#pragma once
#include <stdint.h>

namespace sl {

enum { VT_POSE = 0, VT_POINT = 1, VT_CUBE = 2, VT_CYL = 3 };
enum { FT_BR = 2, FT_CUBE = 3, FT_CYL = 4 };

constexpr int NB = 64;  // Cholesky tile edge

__host__ __device__ inline int lm_dim(int t) { return t == VT_POINT ? 3 : (t == VT_CUBE ? 9 : 7); }
__host__ __device__ inline int lf_rows(int t) { return t == FT_BR ? 3 : (t == FT_CUBE ? 9 : 7); }
// packed sizes (doubles) of one landmark factor's linearisation record [r | Jp (m x 6) | Jl (m x d)]
__host__ __device__ inline int lf_jsize(int t) { return t == FT_BR ? 30 : (t == FT_CUBE ? 144 : 98); }
// and of its Schur record [E (6 x d) | F (6 x d) | u (6)]
__host__ __device__ inline int lf_esize(int t) { return t == FT_BR ? 42 : (t == FT_CUBE ? 114 : 90); }

struct GraphDev {
  // ---- variables -------------------------------------------------------------------------
  int P, L;
  double* pose_val;    // 12 P   theta: R row-major (9) + t (3)
  double* pose_delta;  // 6 P
  double* pose_est;    // 12 P   theta (+) delta
  int* lm_type;        // L      VT_*
  double* lm_val;      // 15 L   POINT xyz | CUBE R t scale | CYL root ray radius
  double* lm_delta;    // 9 L
  double* lm_est;      // 15 L
  // ---- factors ---------------------------------------------------------------------------
  int n_prior;
  int* pr_pose; double* pr_z; double* pr_sigma;   // z: 12, sigma: 6
  double* pr_r;                                   // 6 whitened residual (J = diag(1/sigma))
  int n_between;
  int* bt_i; int* bt_j; double* bt_z; double* bt_sigma;   // z: 12, sigma: 6
  double* bt_r; double* bt_J0;                    // 6 ; 36 (J1 = diag(1/sigma))
  // sharded mode (one robot per GPU): relative-pose factors whose other pose lives on another rank ("ghost")
  int n_ghost;                                    // factors
  int* gh_pose; int* gh_slot; int* gh_first;      // local pose, ghost slot, 1 = local pose is the FIRST key of the Between
  double* gh_z; double* gh_sigma;                 // z: 12, sigma: 6
  double* gh_r; double* gh_J;                     // 6 whitened residual ; 36 whitened Jacobian w.r.t. the local pose
  int n_gslots; double* ghost_val; int* gslot_pose;   // 12 per slot (R row-major, t) ; local pose owning the slot or -1
  int n_lf;                                       // landmark factors, unified id space
  int* lf_type; int* lf_pose; int* lf_lm; int* lf_slot;   // slot = index into the per-type z arrays
  int64_t* lf_joff; int64_t* lf_eoff;             // offsets (doubles) into jbuf / ebuf
  double* br_z;                                   // 4 per BR factor: bearing(3), range
  double* cu_z; double* cu_sigma;                 // 15 ; 9
  double* cy_z;                                   // 7
  double* jbuf;                                   // per factor [r | Jp | Jl], whitened
  double* ebuf;                                   // per factor [E | F | u]
  // ---- topology --------------------------------------------------------------------------
  int* lm_ptr; int* lm_fids;        // landmark -> factor ids (insertion order)
  int* pose_ptr; int* pose_fids;    // pose -> factor ids sorted by (landmark id, factor id)
  int* pose_lms;                    // landmark id of every pose_fids entry (same indexing)
  long long* pose_ed;               // (lf_eoff << 4) | landmark dimension of every pose_fids entry (same indexing)
  int* pose_bt_ptr; int* pose_bt;   // pose -> (between index << 1 | role), role 1 = second key
  unsigned* pose_adj; int adj_words; // per pose j a bitmap (adj_words words) of the poses >= j sharing a landmark / relative-pose factor with it
  // ---- landmark blocks -------------------------------------------------------------------
  double* lm_Hinv;   // 81 L  (d x d used)
  double* lm_g;      // 9 L
  double* lm_Hacc;   // 54 L  packed lower H_ll (45) + g_l (9): partial sums exchanged between robots
  double* lm_t;      // 9 L   sum_f E_f^T delta_p
  int n_slots;       // shared-landmark slots of the multi-robot exchange (global, identical on every rank)
  int* sh_lid;       // slot -> local landmark id or -1
  int* sh_owner;     // slot -> 1 when this rank owns the landmark value
  // ---- pose blocks -----------------------------------------------------------------------
  double* pose_H;    // 36 P
  double* pose_g;    // 6 P   (already reduced: g_p - sum F g_l)
  // ---- dense reduced system ----------------------------------------------------------------
  double* S;         // column-major, ld rows x (T*NB) columns, lower triangle + RHS row at T*NB
  int ld;            // (T + 1) * NB
  int T;             // ceil(6 P / NB)
  double* Ld;        // T * NB*NB : per diagonal block L_kk (column-major 64x64 slot) its 16x16 sub-tiles BELOW the sub-diagonal blocks (all chol_bwd reads)
  double* Winv;      // T * 4*256 : inverses of the four 16x16 diagonal sub-blocks of every L_kk
  double* yv;        // T*NB  forward-substituted RHS
  double* dp;        // T*NB  reduced solution (delta_p = -dp)
  int* chol_ctr;     // T + 2 : work counters of the Cholesky step kernels (self-clearing)
  // profile (envelope) of the reduced system at tile level, from the topology (HostGraph::upload_new): everything outside it is
  // structurally zero in S AND in its Cholesky factor, is never written after the clear that follows a change of profile, and is
  // skipped by the assembly, the factorisation, the substitutions and the products
  const int* prof;   // T : prof[c] = last tile row of block column c inside the profile (monotone in c, >= c); null = dense
  const int* first;  // T : first[r] = first block column whose profile reaches block row r
  int prof_ver;      // bumped whenever the arrays change (launch plans captured in hipGraphs depend on their values)
  int schur_split;   // workgroups per pose column of the Schur assembly: 1 on a narrow profile (the strip is one or two chunks), else 2
  // ---- joint solve over several robots (pcg_kernels.hip) --------------------------------------
  double* S0;        // copy of S (lower triangle + padding) taken before the factorisation overwrites it: the symmetric products
  int save_S0;       // 1: the Schur assembly writes every block to S0 as well (batched passes with the joint solve; else a device copy)
  double* pcg;       // 7 vectors of T*NB doubles: r, u, w, p, s, x, y
  const int* lm_slot; // L   shared slot of a landmark or -1: the cross-robot part of the products reads the exchanged sums through it
  double* pcg_scal;  // 8    gamma_old, alpha_old, alpha, beta, first gamma, last gamma, iterations that did work
  double pcg_tol2;   // the joint solve counts as converged once gamma = r^T M^-1 r <= pcg_tol2 * (first gamma): later iterations are no-ops
  // ---- exact joint step over several robots ("arrow": the shared landmarks are the separator of the joint graph) ---------------
  int arrow;         // 1: landmarks with a shared slot are NOT eliminated into the pose system (H_ll^-1 = 0: F = u = 0); their coupling
                     //    rows E^T ride below the band as nbr border row tiles of S, their own blocks start the border x border block
  int nbr;           // border row tiles: physical tile rows T .. T + nbr - 1 of S; the right-hand-side row is tile row T + nbr
  const int* lm_bord; // L   offset of a shared landmark's tangent coordinates in this robot's border, or -1
  const int* seg_tab; // or null: the band is cut — nseg, the segments' last block columns + 1, then per segment the first block column of every
                      // border tile row + the right-hand side (HostGraph::seg_tab): what lies outside is never written and stays zero
  double* bord;      // ((nbr + 1) * NB) x (nbr * NB), column-major, ldb: border x border block (lower) + right-hand-side row at nbr * NB
  int ldb;
  double* bord0;     // the same layout: what the ASSEMBLY writes (own H_ll blocks, -g_l, the separator poses' entries, lambda rows' -I / -r) —
                     // always the same few positions of a zeroed buffer, so nothing has to be cleared per pass: the robots' border product
                     // reads it and WRITES bord = bord0 - W W^T (round 5; before, bord was cleared, filled and updated in place: 71 MB of
                     // zeros written per pass on C4)
  const int* pose_sep; // P   border offset of a SEPARATOR POSE's six coordinates, or -1 / null.  Nested dissection of the robot's own pose
                      //     chain: the poses of a window as wide as the band is (every coupling across it passes through it) are moved
                      //     out of the band into the first nsep border row tiles (k_sep_extract), the chain falls into independent
                      //     segments that are factored side by side, and the window's own system (nsep tile columns, the rest of the
                      //     border as ITS border) is eliminated at a second level — the serial chain is as long as one segment
  int nsep;           // border row tiles taken by the separator poses (padded to whole tiles; 0: no segmentation)
  int nsep_dim;       // their coordinates (6 per separator pose)
  const int* gh_bord; // n_ghost  border offset of a ghost (inter-robot relative-pose) factor's six "lambda" coordinates, or null: the
                      // factor then enters H_pp / g_p with the other pose frozen (block-Jacobi); non-null (exact joint step): it
                      // enters through the border only — rows J (its Jacobian w.r.t. the own pose), the first-key side adds -I and -r
  // ---- incremental re-factorisation of the streaming path (HostGraph::run_update) --------------------------------------------------
  const int* lm_first; // L   the first (lowest-index) pose observing a landmark: k_relin reports, in status[6], P - (the lowest pose whose
                       //     rows of the reduced system a relinearisation changes) — 0: nothing was relinearised
  int pose0;           // first DIRTY pose of an incremental update (0: everything): the factors of earlier poses, the landmarks no pose >= pose0
                       // observes and those poses' own blocks keep the last solve's linearisation, Schur records, H_pp and g_p (nothing they
                       // depend on moved: HostGraph::run_update) — the linearisation / landmark / pose kernels leave them alone
  int col0;            // columns of S below it hold the factor of the last solve and are left alone by the assembly (k_schur, k_pad_rhs);
                       // their right-hand-side entries are the forward-substituted ones of that solve (yv).  0: assemble everything.
  // ---- Schur assembly from pair lists (batched exact passes: the topology is fixed between the passes) ------------------------------
  const int* sp_idx;        // or null.  (start, count) into sp_pairs of block (row pose pj + d, column pose pj) at 2 (pj * sp_w + d), d < sp_w
  const long long* sp_pairs;// 2 per pair: pose_ed of the row pose's factor (its F record), pose_ed of the column pose's factor (its E record) —
                            // every pair of factors of the two poses on the same private landmark, in the order of the two poses' lists
  int sp_w;                 // widest strip of a pose column inside the profile, in poses
  int* status;       // [0] not-SPD flag (landmark), [1] not-SPD flag (chol), [2] #relinearised, [6] see lm_first
  // ---- parameters ------------------------------------------------------------------------
  int chart;
  double relin_thr;
  double bearing_sigma, cyl_sigma, numdiff_delta;
};

}  // namespace sl
