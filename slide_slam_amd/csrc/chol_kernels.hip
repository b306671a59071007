// Blocked right-looking FP64 Cholesky of the dense reduced pose system on gfx950 matrix cores.
//
// This is the "dense block-diagonal Schur-complement solve" of BASELINE.json's north_star: the
// reference hands the same linear algebra to GTSAM's multifrontal Cholesky (ISAM2Params::CHOLESKY,
// backend/sloam/src/factorgraph/graph.cpp:15).  Layout: S column-major, leading dimension
// ld = (T+1)*64, lower triangle of the (T*64)^2 system plus ONE extra row tile whose first row is
// the right-hand side, so the forward substitution L y = b falls out of the panel/update steps.
// One launch per block column k (k_chol_step): the look-ahead chain of column k (type-A workgroups: pending panel, diagonal
// factor, triangular solve — all on v_mfma_f64_16x16x4_f64) runs beside the trailing update of earlier panels (type-B work
// queue, rank-128 passes); then ONE launch for the backward substitution (k_chol_bwd_chain).
// MFMA operand orientation is chosen so that every global access is a 128-byte run down a column.
// f64 MFMA lane maps (cdna_hip_programming.md §3): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// D[row = (lane>>4) + 4*reg][col = lane&15].
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <utility>
#include <vector>

#include "graph_dev.hpp"
#include "kernels.hpp"

namespace sl {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int CHOL_BATCH_MAX = 8;    // systems per batched launch (robots sharing one GPU)
__device__ inline v4d mfma_f64(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

__device__ inline double rsqrt_nr(double d) {
  // 1/sqrt(d): hardware estimate + two Newton steps (full double precision).  Each step as THREE dependent operations instead of four
  // (t = d y; r = 1/2 - t (y/2); y += y r — y/2 does not wait for t): the four chained calls per 4x4 pivot block sit on the critical path
  // of every step launch (SLIDE_RSQRT_OLD: round 1-4's form, for comparison)
#ifdef SLIDE_RSQRT_OLD
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - 0.5 * d * y * y);
  y = y * (1.5 - 0.5 * d * y * y);
  return y;
#else
  double y = __builtin_amdgcn_rsq(d);
  double t = d * y, h = 0.5 * y;
  double r = __builtin_fma(-t, h, 0.5);
  y = __builtin_fma(y, r, y);
  t = d * y; h = 0.5 * y;
  r = __builtin_fma(-t, h, 0.5);
  y = __builtin_fma(y, r, y);
  return y;
#endif
}

__device__ __forceinline__ double bcast_lane(double v, int src) {   // src: compile-time lane
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

#ifdef SLIDE_STAMPS
__device__ unsigned long long g_stamps[40];
#define STAMP(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMPW(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)   // no barrier
// chained substitutions: wall-clock (100 MHz) stamps of two consecutive blocks in the middle of the chain
__device__ unsigned long long g_chain_stamps[32];
#define CSTAMP(i) do { if (threadIdx.x == 0 && (bidx == 30 || bidx == 31)) g_chain_stamps[(bidx - 30) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define CSTAMPP(i) do { if (threadIdx.x == 256 && (bidx == 30 || bidx == 31)) g_chain_stamps[(bidx - 30) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)   // the polling wave
#else
#define STAMP(i)
#define STAMPW(i)
#define CSTAMP(i)
#define CSTAMPP(i)
#endif

// ------------------------------------------------------------------------------------------------
// ONE kernel per block column k ("step kernel").  Two kinds of workgroups share the launch:
//  type A (blockIdx < nA = T - k, dispatched first): tile (i, k), i = k+1..T (T = the RHS tile).  It applies
//     the trailing update of panel k-1 to its own tile AND (redundantly) to the diagonal tile (k, k), factors
//     the diagonal tile in registers, inverts its four 16x16 diagonal sub-blocks and solves its own 64 rows
//     X = A L^-T by blocked substitution on v_mfma_f64_16x16x4_f64 — i.e. the whole "look-ahead" chain of
//     column k without any inter-workgroup dependency.  Redundant factoring costs no latency (the blocks run
//     concurrently) and keeps L_kk / its inverses out of L2 round trips.
//  type B (work queue): the trailing update in rank-128 passes — panels kb-2, kb-1 of a pair base kb go onto every tile
//     (i, j >= kb+1) in one visit, half of the pass in launch kb, half in launch kb+1 (see b_decode / launch_chol_step).
// The long type-A blocks therefore run BESIDE the type-B flood inside one launch; the next step needs only the kernel
// boundary.  MFMA accumulator layout of X_c^T (row (lane>>4) + 4r, column lane&15) is exactly the B-operand layout of
// k-step r, so chained products need no lane movement.
// ---------------- type B: trailing update with two panels, one 2x2 group of 64x64 tiles per workgroup ----------------
// A work item is one tile row of a 2x2 group: two tiles (i, j0), (i, j0 + 1); four waves per tile, each a 32x32 quadrant:
// four 16x16 accumulators initialised with C itself, 16 k-steps per panel of two 16-byte operand loads feeding four MFMAs (the loads
// run five k-steps ahead of their use).  No LDS, no barrier: waves whose tile lies outside the lower triangle leave at once.
__device__ __forceinline__ long long tri_row(long long t) {
  long long ii = (long long)floor((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (ii * (ii + 1) / 2 > t) --ii;
  while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
  return ii;
}
// Pair schedule (rank-128 trailing update): the panels of two consecutive block columns are applied in ONE pass over the
// trailing matrix, so every C tile is read and written once per two columns.  kb (even) is the base of the pair: panels
// kb-2 and kb-1 go onto the tiles (i, j >= kb+1); the first part of the item list (which holds all of tile columns kb+1,
// kb+2) runs in launch kb, the rest in launch kb+1.  Items are enumerated column-major over the 2x2 groups.  Launch kb+1
// also brings tile column kb+2 up to panel kb (rank 64, "column items"), so that every type-A column has ONE pending panel.
// The row <-> lane map of the quadrant is permuted (lane lr owns rows 2 lr, 2 lr + 1 of its 32-row half instead of lr, lr + 16)
// so that every operand / C access is 16 contiguous bytes per lane: half the vector-memory instructions of the natural map,
// which is what bounds an item next to the matrix pipe (tools/b_bench.hip: latency hiding across items gains nothing).
struct BItem {
  int i, j, pcb, ks;   // tile (i, j); first panel column block; k-steps of four columns: 16 = one panel, 32 = two
  bool ok;
};
typedef double v2d __attribute__((ext_vector_type(2)));
template <int KS>
__device__ __forceinline__ void b_quadrant(double* __restrict__ S, int ld, const BItem& it, int wq) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int ch = (wq >> 1) & 1, rh = wq & 1;      // column half, row half of the tile
  const double* pjh = S + (size_t)(it.pcb * NB + lk) * ld + (size_t)it.j * NB + 32 * ch + 2 * lr;   // rows 32 ch + 2 lr + a of panel tiles (j, pcb ..)
  const double* pih = S + (size_t)(it.pcb * NB + lk) * ld + (size_t)it.i * NB + 32 * rh + 2 * lr;   // rows 32 rh + 2 lr + b of panel tiles (i, pcb ..)
  // acc[a][b] register r of lane (lr, lk) = C[row 32 rh + 2 lr + b][column 32 ch + 2 (lk + 4 r) + a]
  double* cbh = S + (size_t)(it.j * NB + 32 * ch + 2 * lk) * ld + (size_t)it.i * NB + 32 * rh + 2 * lr;   // + (8 r + a) ld + b
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const v2d c2 = *(const v2d*)(cbh + (size_t)(8 * r + a) * ld);
      acc[a][0][r] = c2[0]; acc[a][1][r] = c2[1];
    }
#ifndef SLIDE_B_RD
#define SLIDE_B_RD 4
#endif
  constexpr int RD = SLIDE_B_RD;             // operand ring: loads run RD - 1 k-steps ahead
  v2d pa[RD], pb[RD];
#pragma unroll
  for (int pre = 0; pre < RD - 1; ++pre) {
    const size_t off = (size_t)(4 * pre) * ld;
    pa[pre] = *(const v2d*)(pjh + off);
    pb[pre] = *(const v2d*)(pih + off);
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + RD - 1 < KS) {
      const size_t off = (size_t)(4 * (ks + RD - 1)) * ld;
      pa[(ks + RD - 1) % RD] = *(const v2d*)(pjh + off);
      pb[(ks + RD - 1) % RD] = *(const v2d*)(pih + off);
    }
    __builtin_amdgcn_sched_barrier(0);      // keep the prefetch distance: no further hoisting of loads
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const double na = -pa[ks % RD][a];
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = mfma_f64(na, pb[ks % RD][b], acc[a][b]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v2d c2;
      c2[0] = acc[a][0][r]; c2[1] = acc[a][1][r];
      *(v2d*)(cbh + (size_t)(8 * r + a) * ld) = c2;
    }
}
// item g of launch k.  g < nR: item g0 + g of the rank-128 pass of pair base kb = tile row i of a 2x2 group, tiles (i, j0),
// (i, j0 + 1).  Then the column items c = g - nR of an odd launch: tiles (k+1 + 2c, k+1), (k+2 + 2c, k+1) take panel k-1.
// Eight waves: tile (wave >> 2), 32x32 quadrant (wave & 3).
// Profile (envelope) of the factor: the launches enumerate rows / columns up to a VIRTUAL size Tv <= T — real tile rows 0 .. Tv-1, the
// right-hand-side row as virtual row Tv — and map virtual row Tv to the physical tile row T.  TvB = profile of panel kb-1 (the rank-128
// pass), TvX = profile of panel k-1 (the column items); tiles beyond hold zeros and get zero contributions (launch_chol_step).
// Border (nbr > 0, the exact joint step of several robots, host_graph.hip "arrow"): nbr further row tiles ride below the profile like
// the right-hand side does — virtual rows Tv .. Tv + nbr - 1 are the physical tile rows T .. T + nbr - 1 (the coupling of the band's
// columns to the separator variables, W^T = B^T L^-T after the steps), virtual row Tv + nbr the right-hand side at physical row T + nbr.
// The steps never touch border COLUMNS: the border x border block is one product at the end (k_border_syrk).
// nbB / nbX: the border rows that are active in the pass / for the column items — the rows whose coupling starts at a later block column
// are still all-zero and form a suffix of the border (plan_step); the right-hand side is virtual row Tv + (active rows).
__device__ __forceinline__ BItem b_decode(int g, int nItems, int nR, int g0, int k, int kb, int T, int TvB, int TvX, int nP, long long nG, int wave,
                                          int nbr, int nbB, int nbX, const int* __restrict__ ord = nullptr) {
  BItem it;
  it.ok = false; it.i = it.j = it.pcb = 0; it.ks = 32;
  if (g >= nItems) return it;
  if (g < nR) {
    const long long tt = (long long)g0 + g;
    const long long t = nG - 1 - (tt >> 1);
    const long long u = tri_row(t);
    const int v = (int)(t - u * (u + 1) / 2);
    const int bi = nP - 1 - v, bj = nP - 1 - (int)u;
    it.i = kb + 1 + 2 * bi + (int)(tt & 1);
    it.j = kb + 1 + 2 * bj + (wave >> 2);
    it.pcb = kb - 2;
    it.ok = !(it.i > TvB + nbB || it.j > TvB - 1 || it.i < it.j);
    if (it.i >= TvB) it.i = (it.i - TvB < nbB) ? T + (ord ? ord[it.i - TvB] : it.i - TvB) : T + nbr;
  } else {
    const int c = g - nR;
    it.j = k + 1;
    it.i = k + 1 + 2 * c + (wave >> 2);
    it.pcb = k - 1; it.ks = 16;
    it.ok = !(it.i > TvX + nbX || it.j > TvX - 1);
    if (it.i >= TvX) it.i = (it.i - TvX < nbX) ? T + (ord ? ord[it.i - TvX] : it.i - TvX) : T + nbr;
  }
  return it;
}

// ---------------- type A: look-ahead chain of block column k for row tile it = k + 1 + ia ----------------
// The 64x64 diagonal tile D = (k, k) is handled as 16x16 sub-tiles in the MFMA accumulator layout of the transpose
// ("T-layout": lane (lr = lane&15, lk = lane>>4), register r of sub-tile (I, J) holds D[16I + lr][16J + lk + 4r]).
// In that layout register s of a sub-tile is the B operand "rows lr, columns 4s..4s+3", so
//   * the rank-4 panel step X = D[:, j..j+3] L_p^-T is ONE v_mfma_f64_16x16x4_f64 per 16 rows: the A operand "mop" is
//     the 4x4 inverse of the pivot block's Cholesky factor zero-padded to 16x4, and register 0 of the product is X in
//     the same "rows lr, four columns" form;
//   * the rank-4 update of a sub-tile is ONE MFMA with two such X registers as A and B operands.
// Only the 4x4 pivot Cholesky + inverse (four chained rsqrt) runs on the vector ALU, on wave-uniform values fetched
// with v_readlane.  That serial chain is the critical path of the whole factorisation, so the eight waves specialise:
//   wave 0 ("chain"): owns the current diagonal sub-tile (J, J); per iteration n = 4 JQ + s it factors the pivot block,
//       publishes mop and the masked X of its sub-tile (xm) through LDS and raises it_done; its own MFMAs are the
//       panel step and the in-phase update of that one sub-tile.
//   waves 1..3 ("workers"): own sub-tile row w of D.  They follow the published (mop, xm) stream one iteration behind:
//       panel step and in-phase update of (w, JQ), at the end of a 16-column phase the rank-16 updates of (w, J > JQ) —
//       the neighbours' finished sub-tiles come from LDS — and after phase w-1 they hand the finished diagonal
//       sub-tile (w, w) to the chain wave.  The update of their sub-tiles with panel k-1 (which commutes with the
//       rank-16 updates) is spread over the iterations before each sub-tile's own phase.  Worker 2 also carries the
//       identity pseudo-tile of every phase, which ends as L_JJ^-T (the 16x16 inverses of the triangular solves).
//   waves 4..7 ("panel"): own 16 rows each of the panel tile (it, k): update with panel k-1 while the factorisation
//       runs on the other wave of their SIMD, then, after the closing barrier, X = A L^-T for their rows by blocked
//       substitution with the finished sub-tiles (LDS) and inverses.
// The panel tile (k, k-1) all of them need as A operand is staged once through LDS.  All hand-offs are LDS flags between
// resident waves of one workgroup (no cycle: every wait is on an earlier iteration / phase).  Redundant factoring
// across the T-k workgroups costs no latency and keeps L_kk out of L2.
constexpr int PSTR = 80;   // LDS column stride of the staged panel tile: the four 16-lane groups of a ds_read_b64 fall in disjoint bank halves
constexpr int WT = 2;      // the worker that carries the identity pseudo-tiles
struct ALds {
  double Pk[NB][PSTR];  // pending panel tiles (k, k-NPAN .. k-1): Pk[kk][r] = L[k*NB + r][(k-NPAN)*NB + kk]   (rows of the diagonal block)
  double mop[16][64];   // iteration n: A operand of the panel MFMAs, lane image
  double xm[16][64];    // iteration n: masked X of the diagonal sub-tile = A operand of the in-phase updates, lane image
  double Lt[6][4][64];  // finished off-diagonal sub-tiles L(I, J), I > J, at index I (I - 1) / 2 + J, register, lane
  double Dh[3][4][64];  // hand-off of the diagonal sub-tile (J, J), J = 1..3, updated up to the LAST BUT ONE iteration of phase J - 1
  double Dh3[3][64];    // ... with the worker's register 3 of sub-tile (J, J - 1) at that point: the chain wave applies the last iteration itself
  double Wi[4][256];    // Wi[b][c * 16 + r] = (L_bb^-1)[r][c]
  int it_done;          // iterations published by the chain wave
  int col_done[4];      // col_done[I]: columns J of row I published in Lt
  int d_ready[4];       // d_ready[J]: Dh[J - 1] is valid
  int w_done;           // inverses Wi[0 .. w_done-1] are valid
};

// Flags between the waves of one workgroup live in LDS.  DS operations of one wave are carried out in issue order,
// so a flag written after its data is seen after it; only the compiler has to be kept from reordering (no s_waitcnt
// on the producer: a release fence would also drain the wave's outstanding global loads and stores).
// Raw DS instructions on the LDS byte offset (the low 32 bits of the generic address): a volatile access through a
// generic pointer would become a flat store with system scope followed by s_waitcnt vmcnt(0).
__device__ __forceinline__ void lds_wait(int* f, int v) {
  const unsigned off = (unsigned)(size_t)f;
  int cur;
  for (;;) {
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(cur) : "v"(off) : "memory");
    if (__builtin_amdgcn_readfirstlane(cur) >= v) break;
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ void lds_post(int* f, int v, int lane) {
  const unsigned off = (unsigned)(size_t)f;
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(off), "v"(v) : "memory");
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}
__device__ __forceinline__ constexpr int oidx(int I, int J) { return I * (I - 1) / 2 + J; }   // I > J
// The flag waits are opaque asm: without a use of the accumulators in front of them the compiler sinks the MFMAs that
// were meant to run during the wait to behind it.
__device__ __forceinline__ void pin(v4d& x) { asm volatile("" : "+v"(x)); }

// ---- wave 0 ---------------------------------------------------------------------------------------------------------
template <int NPAN>
__device__ __forceinline__ void a_chain_wave(int ia, int* status, ALds& L, int lane, v4d Dt) {
  const int lr = lane & 15, lk = lane >> 4;
  const v4d zero = v4d{0.0, 0.0, 0.0, 0.0};
  // lane predicates of the 4x4 inverse inside the 16x4 A operand: row lr < 4, column lk <= lr
  const bool s00 = lr == 0 && lk == 0, s10 = lr == 1 && lk == 0, s11 = lr == 1 && lk == 1, s20 = lr == 2 && lk == 0,
             s21 = lr == 2 && lk == 1, s22 = lr == 2 && lk == 2, s30 = lr == 3 && lk == 0, s31 = lr == 3 && lk == 1,
             s32 = lr == 3 && lk == 2, s33 = lr == 3 && lk == 3;
  if (NPAN > 0) {
    // panel k-1 on sub-tile (0, 0): four independent accumulation chains instead of one of sixteen
    v4d d1 = zero, d2 = zero, d3 = zero;
#pragma unroll
    for (int ks = 0; ks < 4 * NPAN; ++ks) {
      const double q0 = L.Pk[4 * ks + lk][lr], q1 = L.Pk[16 * NPAN + 4 * ks + lk][lr], q2 = L.Pk[32 * NPAN + 4 * ks + lk][lr],
                   q3 = L.Pk[48 * NPAN + 4 * ks + lk][lr];
      Dt = mfma_f64(-q0, q0, Dt);
      d1 = mfma_f64(-q1, q1, d1);
      d2 = mfma_f64(-q2, q2, d2);
      d3 = mfma_f64(-q3, q3, d3);
    }
    Dt = (Dt + d1) + (d2 + d3);
  }
  bool ok = true;
  STAMPW(3);
  v4d Dnext = zero;
#pragma unroll
  for (int JQ = 0; JQ < 4; ++JQ) {
    if (JQ > 0) Dt = Dnext;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (lr < lk + 4 * r) Dt[r] = 0.0;          // upper triangle: never read for a result, keep it finite
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n = 4 * JQ + s;
      const double dv = Dt[s];   // columns 4s..4s+3 of the sub-tile; element (4s+a, 4s+b) sits in lane 4s+a+16b
      const double P00 = bcast_lane(dv, 4 * s), P10 = bcast_lane(dv, 4 * s + 1), P20 = bcast_lane(dv, 4 * s + 2), P30 = bcast_lane(dv, 4 * s + 3);
      const double P11 = bcast_lane(dv, 4 * s + 1 + 16), P21 = bcast_lane(dv, 4 * s + 2 + 16), P31 = bcast_lane(dv, 4 * s + 3 + 16);
      const double P22 = bcast_lane(dv, 4 * s + 2 + 32), P32 = bcast_lane(dv, 4 * s + 3 + 32), P33 = bcast_lane(dv, 4 * s + 3 + 48);
      // a non-positive pivot only raises the flag: the NaNs it breeds stay inside this (rejected) factorisation
      const double i0 = rsqrt_nr(P00);
      const double l10 = P10 * i0, l20 = P20 * i0, l30 = P30 * i0;
      const double d1 = P11 - l10 * l10;
      const double i1 = rsqrt_nr(d1);
      const double l21 = (P21 - l20 * l10) * i1, l31 = (P31 - l30 * l10) * i1;
      const double d2 = P22 - l20 * l20 - l21 * l21;
      const double i2 = rsqrt_nr(d2);
      const double l32 = (P32 - l30 * l20 - l31 * l21) * i2;
      const double d3 = P33 - l30 * l30 - l31 * l31 - l32 * l32;
      const double i3 = rsqrt_nr(d3);
      ok = ok && (P00 > 0.0) && (d1 > 0.0) && (d2 > 0.0) && (d3 > 0.0);
      // M = L_p^-1 (lower 4x4)
      const double m10 = -l10 * i0 * i1;
      const double m21 = -l21 * i1 * i2;
      const double m20 = -(l20 * i0 + l21 * m10) * i2;
      const double m32 = -l32 * i2 * i3;
      const double m31 = -(l31 * i1 + l32 * m21) * i3;
      const double m30 = -(l30 * i0 + l31 * m10 + l32 * m20) * i3;
      double mop = 0.0;                       // selected in order of availability: the last link follows m30 directly
      mop = s00 ? i0 : mop;  mop = s11 ? i1 : mop;  mop = s10 ? m10 : mop;  mop = s22 ? i2 : mop;  mop = s21 ? m21 : mop;
      mop = s20 ? m20 : mop; mop = s33 ? i3 : mop;  mop = s32 ? m32 : mop;  mop = s31 ? m31 : mop;  mop = s30 ? m30 : mop;
      L.mop[n][lane] = mop;
      const double xd = mfma_f64(mop, Dt[s], zero)[0];          // X[row lr][column lk] of the diagonal sub-tile
      Dt[s] = xd;
      const double xm = (lr > 4 * s + 3) ? xd : 0.0;            // rows below the pivot block only
      L.xm[n][lane] = xm;
      lds_post(&L.it_done, n + 1, lane);
      if (n < 4) STAMPW(32 + n);
      if (s < 3) Dt = mfma_f64(-xm, xm, Dt);
      if (s == 3 && JQ < 3) {
        // the next diagonal sub-tile: the worker of row JQ + 1 handed it over after the last but one iteration of this phase (long
        // since: it follows one iteration behind), with its register 3 of sub-tile (JQ + 1, JQ) — the last iteration's panel step and
        // rank-4 update are two MFMAs here instead of a round trip through the worker (~0.5 us per phase, on every launch's chain)
        STAMPW(3 + 2 * JQ + 1);
        lds_wait(&L.d_ready[JQ + 1], 1);
        STAMPW(3 + 2 * JQ + 2);
        const double x3 = mfma_f64(mop, L.Dh3[JQ][lane], zero)[0];
#pragma unroll
        for (int r = 0; r < 4; ++r) Dnext[r] = L.Dh[JQ][r][lane];
        Dnext = mfma_f64(-x3, x3, Dnext);
      }
    }
  }
  STAMPW(10);
  if (!ok && lane == 0 && ia == 0) atomicOr(&status[1], 1);
}

// ---- waves 1..3 -----------------------------------------------------------------------------------------------------
template <int W, int NPAN>
__device__ __forceinline__ void a_worker_wave(int ia, double* __restrict__ Ld, double* __restrict__ Winv, ALds& L, int lane,
                                              v4d (&R)[W + 1]) {
  const int lr = lane & 15, lk = lane >> 4;
  const v4d zero = v4d{0.0, 0.0, 0.0, 0.0};
  if (NPAN > 0) {
    // the whole sub-tile row before the first iteration, in the order of need: a worker answers an iteration of the
    // chain wave in about 500 cycles but needs twice that with a quarter of a sub-tile update on top, so it is better
    // late for the first iterations (it catches up well before its hand-off) than slow in all of them
    // (a chain of dependent MFMAs fed from LDS runs at ~150 cycles per link: 2 (W + 1) independent chains, k-steps outermost)
    v4d e[W + 1];
#pragma unroll
    for (int J = 0; J <= W; ++J) e[J] = zero;
#pragma unroll
    for (int ks = 0; ks < 16 * NPAN; ks += 2) {
      const double w0 = L.Pk[4 * ks + lk][16 * W + lr], w1 = L.Pk[4 * ks + 4 + lk][16 * W + lr];
#pragma unroll
      for (int J = 0; J <= W; ++J) {
        R[J] = mfma_f64(-L.Pk[4 * ks + lk][16 * J + lr], w0, R[J]);
        e[J] = mfma_f64(-L.Pk[4 * ks + 4 + lk][16 * J + lr], w1, e[J]);
      }
    }
#pragma unroll
    for (int J = 0; J <= W; ++J) {
      R[J] += e[J];
      pin(R[J]);
    }
  }
  if (W == 1) STAMPW(37);
  v4d Wt = zero;                   // identity pseudo-tile (worker WT only)
#pragma unroll
  for (int JQ = 0; JQ < W; ++JQ) {
    if (W == WT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Wt[r] = (lr == lk + 4 * r) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n = 4 * JQ + s;
      lds_wait(&L.it_done, n + 1);
      if (W == 1 && n == 3) STAMPW(11);
      if (W == 1 && n < 3) STAMPW(16 + n);
      if (W == 2 && n < 12) STAMPW(20 + n);
      const double mop = L.mop[n][lane], xm = L.xm[n][lane];
      const double x = mfma_f64(mop, R[JQ][s], zero)[0];
      double xw = 0.0;
      if (W == WT) xw = mfma_f64(mop, Wt[s], zero)[0];
      R[JQ][s] = x;
      if (W == WT) Wt[s] = xw;
      if (s < 3) {
        R[JQ] = mfma_f64(-xm, x, R[JQ]);
        if (W == WT) Wt = mfma_f64(-xm, xw, Wt);
      }
      // the rank-16 update of the diagonal sub-tile one k-step at a time: register s of (W, JQ) is final from here on
      R[W] = mfma_f64(-R[JQ][s], R[JQ][s], R[W]);
#pragma unroll
      for (int J = 0; J <= W; ++J) pin(R[J]);
      if (W == WT) pin(Wt);
      if (JQ == W - 1 && s == 2) {
        // hand-off of the own diagonal sub-tile one iteration early: everything but the last iteration's rank-4 update is in it, and
        // register 3 of (W, JQ) as it stands goes along — the chain wave finishes both itself (a_chain_wave)
#pragma unroll
        for (int r = 0; r < 4; ++r) L.Dh[W - 1][r][lane] = R[W][r];
        L.Dh3[W - 1][lane] = R[JQ][3];
        lds_post(&L.d_ready[W], 1, lane);
        if (W == 1) STAMPW(13);
        if (W == 2) STAMPW(36);
      }
    }
    // column block JQ of row W is final
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      L.Lt[oidx(W, JQ)][r][lane] = R[JQ][r];
      if (ia == 0) Ld[(size_t)(16 * JQ + lk + 4 * r) * NB + 16 * W + lr] = R[JQ][r];
    }
    lds_post(&L.col_done[W], JQ + 1, lane);
    if (W == 1) STAMPW(12);
    // rank-16 updates of the other sub-tiles right of this phase
#pragma unroll
    for (int J = JQ + 1; J < W; ++J) {
      lds_wait(&L.col_done[J], JQ + 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) R[J] = mfma_f64(-L.Lt[oidx(J, JQ)][ks][lane], R[JQ][ks], R[J]);
    }
    if (W == WT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = (lk + 4 * r >= lr) ? Wt[r] : 0.0;      // Wt lane (lr, lk) reg r = (L_JQ,JQ^-1)[lk + 4r][lr]
        L.Wi[JQ][lr * 16 + lk + 4 * r] = v;
        if (ia == 0) Winv[(size_t)JQ * 256 + lr * 16 + lk + 4 * r] = v;
      }
      lds_post(&L.w_done, JQ + 1, lane);
    }
  }
  static_assert(W >= 1 && W <= 3, "worker index");
  if (W == WT) {
    // identity pseudo-tiles of the phases after the own ones
#pragma unroll
    for (int JQ = WT; JQ < 4; ++JQ) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Wt[r] = (lr == lk + 4 * r) ? 1.0 : 0.0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int n = 4 * JQ + s;
        lds_wait(&L.it_done, n + 1);
        const double mop = L.mop[n][lane], xm = L.xm[n][lane];
        const double xw = mfma_f64(mop, Wt[s], zero)[0];
        Wt[s] = xw;
        if (s < 3) Wt = mfma_f64(-xm, xw, Wt);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = (lk + 4 * r >= lr) ? Wt[r] : 0.0;
        L.Wi[JQ][lr * 16 + lk + 4 * r] = v;
        if (ia == 0) Winv[(size_t)JQ * 256 + lr * 16 + lk + 4 * r] = v;
      }
      lds_post(&L.w_done, JQ + 1, lane);
    }
  }
}

// ---- panel rows: 16 rows of tile (it, k), T-layout: Tq[b][r] = A[16 q + lr][16b + lk + 4r] ----------------------------------
template <int NPAN>
__device__ __forceinline__ void panel_load(const double* __restrict__ S, int ld, int k, int it, int q, int lr, int lk,
                                           v4d (&Tq)[4], double (&tb)[16]) {
  const double* tcol = S + (size_t)(k * NB) * ld + (size_t)it * NB + 16 * q + lr;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) Tq[b][r] = tcol[(size_t)(16 * b + lk + 4 * r) * ld];
  if (NPAN > 0) {
    const double* pi = S + (size_t)((k - NPAN) * NB) * ld + (size_t)it * NB + 16 * q + lr;   // own rows of panel tiles (it, k-NPAN .. k-1)
#pragma unroll
    for (int ks = 0; ks < 16 * NPAN; ++ks) tb[ks] = pi[(size_t)(4 * ks + lk) * ld];
  }
}
// Column block p of the own rows: pending update from panel k-1, then X_p = (A_p - sum_{c<p} X_c L(p,c)^T) L_pp^-T as soon
// as phase p of the factorisation has delivered L(p, c) and the inverse of L_pp — only the last block is left when the
// chain wave finishes.  Nothing before the first phase is through: until then the factor waves are busy with panel k-1
// themselves and this wave would only compete for the matrix pipe and the LDS.
template <int NPAN, bool EARLY>
__device__ __forceinline__ void panel_rows(double* __restrict__ S, int ld, int k, int it, int q, int gate, int lane, ALds& L, v4d (&Tq)[4],
                                           const double (&tb)[16], float* __restrict__ L32t) {
  const int lr = lane & 15, lk = lane >> 4;
  double* tcol = S + (size_t)(k * NB) * ld + (size_t)it * NB + 16 * q + lr;
  float* pcol = L32t ? L32t + lk * NB + 16 * q + lr : nullptr;      // packed f32 copy of the tile (element (row, col) at col * 64 + row)
  v4d xt[4], t[4];
  // order: | T(0) x(0) T(1) t(1) | x(1) T(2) t(2) | x(2) T(3) t(3) | x(3): after the last phase only the four MFMAs with the
  // last inverse are left.  t(b) = A_b - sum_{c<b} X_c L(b,c)^T needs phase b-1, x(b) = t(b) L_bb^-T the inverse of phase b.
  // This wave shares its SIMD with a worker that answers the chain wave with two or three MFMAs per iteration: the
  // throughput work here goes in bursts of four MFMAs with a pause after each, so the matrix pipe is free half the time.
#define PANEL_YIELD() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(1); __builtin_amdgcn_sched_barrier(0); } while (0)
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (b == 0) {
      if (NPAN > 0 && EARLY) {
        if (gate == 1) lds_wait(&L.d_ready[1], 1);   // wave 5 / 6 share their SIMD with worker 1 / 2: not before that one's hand-off
        if (gate == 2) lds_wait(&L.d_ready[2], 1);   // (worker 3 has two phases of slack: wave 7 starts at once)
        // the whole pending update at once, while the factor waves are in their own: the phases that follow then see
        // only the short substitution bursts of this wave on their SIMD
#pragma unroll
        for (int ks = 0; ks < 16 * NPAN; ++ks) {
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) Tq[bb] = mfma_f64(-L.Pk[4 * ks + lk][16 * bb + lr], tb[ks], Tq[bb]);
        }
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) pin(Tq[bb]);
      }
      lds_wait(&L.it_done, 4);
      if (NPAN > 0 && !EARLY) {
#pragma unroll
        for (int ks = 0; ks < 16 * NPAN; ++ks) {
          Tq[0] = mfma_f64(-L.Pk[4 * ks + lk][lr], tb[ks], Tq[0]);
          if ((ks & 3) == 3) PANEL_YIELD();
        }
      }
      t[0] = Tq[0];
    }
    lds_wait(&L.w_done, b + 1);
    v4d x = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) x = mfma_f64(L.Wi[b][(4 * s4 + lk) * 16 + lr], t[b][s4], x);   // (L_bb^-1)[lr][4 s4 + lk]
    xt[b] = x;
#pragma unroll
    for (int r = 0; r < 4; ++r) tcol[(size_t)(16 * b + lk + 4 * r) * ld] = x[r];
    if (pcol) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pcol[(16 * b + 4 * r) * NB] = (float)x[r];
    }
    if (b < 3) {
      PANEL_YIELD();
      if (NPAN > 0 && !EARLY) {
#pragma unroll
        for (int ks = 0; ks < 16 * NPAN; ++ks) {
          Tq[b + 1] = mfma_f64(-L.Pk[4 * ks + lk][16 * (b + 1) + lr], tb[ks], Tq[b + 1]);
          if ((ks & 3) == 3) PANEL_YIELD();
        }
      }
      lds_wait(&L.col_done[b + 1], b + 1);
      v4d tn = Tq[b + 1];
#pragma unroll
      for (int c = 0; c <= b; ++c) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) tn = mfma_f64(-L.Lt[oidx(b + 1, c)][s4][lane], xt[c][s4], tn);   // -L[16(b+1) + lr][16c + lk + 4 s4]
        if (b < 2 || c < b) PANEL_YIELD();        // (not before the closing x(3))
      }
      t[b + 1] = tn;
      pin(t[b + 1]);
    }
  }
#undef PANEL_YIELD
}

template <int NPAN>
__device__ __forceinline__ void step_type_a_impl(double* __restrict__ S, int ld, int k, int ia, int it, int half, double* __restrict__ Ld,
                                                 double* __restrict__ Winv, int* status, ALds& L, float* __restrict__ L32t) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..7
  const int lr = lane & 15, lk = lane >> 4;
  STAMP(0);
  // ---- prologue: every global load of this workgroup is issued here, in one wave of traffic ----
  if (NPAN > 0) {
    // one eighth of the pending panel tiles (k, k-NPAN .. k-1) per wave -> LDS
    constexpr int NC = 8 * (NPAN > 0 ? NPAN : 1);
    const double* pq = S + (size_t)((k - NPAN) * NB + NC * wave) * ld + (size_t)k * NB + lane;
    double stage[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) stage[c] = pq[(size_t)c * ld];
#pragma unroll
    for (int c = 0; c < NC; ++c) L.Pk[NC * wave + c][lane] = stage[c];
  }
  if (tid == 0) L.it_done = 0;
  if (tid < 4) { L.col_done[tid] = 0; L.d_ready[tid] = 0; }
  if (tid == 4) L.w_done = 0;
  if (wave == 1) STAMPW(38);
  if (wave < 4) {
    // ---------------- factor waves: own sub-tile row of D ----------------
    const double* dcol = S + (size_t)(k * NB) * ld + (size_t)k * NB + 16 * wave + lr;
    v4d R[4];
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) R[J][r] = (J <= wave) ? dcol[(size_t)(16 * J + lk + 4 * r) * ld] : 0.0;
    v4d Tq[4];
    double tb[16];
    if (wave == 1 && half < 0) panel_load<NPAN>(S, ld, k, it, 0, lr, lk, Tq, tb);
    __syncthreads();
    if (wave == 1) STAMPW(39);
    if (wave == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);   // ahead of the panel wave sharing the SIMD
    if (wave == 0) {
      a_chain_wave<NPAN>(ia, status, L, lane, R[0]);
    } else if (wave == 1) {
      v4d R1[2] = {R[0], R[1]};
      a_worker_wave<1, NPAN>(ia, Ld, Winv, L, lane, R1);
      if (half < 0) panel_rows<NPAN, false>(S, ld, k, it, 0, 0, lane, L, Tq, tb, L32t);      // idle from here on otherwise
    } else if (wave == 2) {
      v4d R2[3] = {R[0], R[1], R[2]};
      a_worker_wave<2, NPAN>(ia, Ld, Winv, L, lane, R2);
    } else {
      a_worker_wave<3, NPAN>(ia, Ld, Winv, L, lane, R);
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    STAMP(1);
    STAMP(2);
  } else {
    // ---------------- panel waves 5..7: rows 16..63 of tile (it, k); rows 0..15 go to worker 1 (idle after its
    // hand-off), so that the chain wave has its SIMD to itself (wave 4 only keeps the barriers company) ----------------
    // rows of the panel tile: 16 (wave - 4) .. ; with two workgroups per tile (half = 0 / 1) waves 5, 6 take rows 32 half + 0 / 16,
    // wave 7 and worker 1 none
    const int q = half < 0 ? wave - 4 : (wave == 5 ? 2 * half : (wave == 6 ? 2 * half + 1 : -1));
    const bool rows = half < 0 ? q > 0 : q >= 0;
    v4d Tq[4];
    double tb[16];
    if (rows) panel_load<NPAN>(S, ld, k, it, q, lr, lk, Tq, tb);
    __syncthreads();
    if (rows) panel_rows<NPAN, true>(S, ld, k, it, q, wave - 4, lane, L, Tq, tb, L32t);
    STAMPW(14);
    __syncthreads();
    STAMP(1);
    STAMP(2);
  }
}
// L32 (or null): packed f32 copy of the factor's off-diagonal tiles, written along with the panel (the preconditioner of the joint solve
// streams it, see bwd_chain_body): tile (i, k), k < i < T, at 4096 * (k (T-1) - k (k-1) / 2 + i - k - 1)
__device__ __forceinline__ void step_type_a(double* __restrict__ S, int ld, int k, int T, int TvA, int ia, int half, double* __restrict__ Ld,
                                            double* __restrict__ Winv, int* status, ALds& L, float* __restrict__ L32, int nbr, int nbA,
                                            const int* __restrict__ ord = nullptr) {
  const int vi = k + 1 + ia - TvA;      // rows k+1 .. TvA-1 of the profile, then the nbA active border rows, then the right-hand-side row
  const int it = vi < 0 ? k + 1 + ia : (vi < nbA ? T + (ord ? ord[vi] : vi) : T + nbr);
  float* L32t = (L32 && it < T) ? L32 + ((size_t)k * (T - 1) - (size_t)k * (k - 1) / 2 + (it - k - 1)) * (NB * NB) : nullptr;
  if (k > 0) step_type_a_impl<1>(S, ld, k, ia, it, half, Ld, Winv, status, L, L32t);
  else step_type_a_impl<0>(S, ld, k, ia, it, half, Ld, Winv, status, L, L32t);
}

// One launch per block column k.  Workgroups 0 .. T-k-1 are type A (column k with its pending panel k-1); every workgroup of
// the launch (the type-A ones after their chain, too, when a_joins) then draws items from the work queue ctr[k]: rank-128 items
// g0 <= g < g1 of the pair base kb, then nX column items — so the ~T-k long type-A chains and the flood balance by themselves at
// the one workgroup per CU the register-heavy kernel gets.  ctr[k+1] is cleared here for the next launch (ctr[0..1] start at 0).
__global__ __launch_bounds__(512) void k_chol_step(double* __restrict__ S, int ld, int k, int T, double* __restrict__ Ld,
                                                   double* __restrict__ Winv, int* status, int* __restrict__ ctr, int kb, int nP,
                                                   int g0, int g1, int nX, int a_joins, int a_split, float* __restrict__ L32,
                                                   int TvA, int TvB, int TvX, int nbr, int nbA, int nbB, int nbX) {
  __shared__ ALds L;
  __shared__ int s_g;
  const int nA = (TvA - k + nbA) << a_split;      // a_split: two type-A workgroups per tile, 32 panel rows each (A-bound launches: the CUs are there)
  if (blockIdx.x == 0 && threadIdx.x == 0) ctr[k + 1] = 0;
  if ((int)blockIdx.x < nA) {
    step_type_a(S, ld, k, T, TvA, (int)blockIdx.x >> a_split, a_split ? (int)(blockIdx.x & 1) : -1, Ld, Winv, status, L, L32, nbr, nbA);
    if (!a_joins) return;        // the queue workers are through before the chain is: an item taken now would only add a tail
  }
  const int nR = g1 - g0, nItems = nR + nX;
  if (nItems <= 0) return;
  const long long nG = (long long)nP * (nP + 1) / 2;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_g = atomicAdd(&ctr[k], 1);
    __syncthreads();
    const int g = __builtin_amdgcn_readfirstlane(s_g);
    if (g >= nItems) break;
    const BItem it = b_decode(g, nItems, nR, g0, k, kb, T, TvB, TvX, nP, nG, wave, nbr, nbB, nbX);
    if (!it.ok) continue;
    if (it.ks == 32) b_quadrant<32>(S, ld, it, wave & 3);
    else b_quadrant<16>(S, ld, it, wave & 3);
  }
}

// Several factorisations (the robots that share a GPU) in ONE launch per block column: the type-A workgroups of every system,
// then one work queue over the type-B items of all of them.  Same device code as k_chol_step, per-system parameters by value.
struct CholBatchArgs {
  int n;
  double* S[CHOL_STEP_BATCH_MAX]; int ld[CHOL_STEP_BATCH_MAX]; int T[CHOL_STEP_BATCH_MAX];
  double* Ld[CHOL_STEP_BATCH_MAX]; double* Winv[CHOL_STEP_BATCH_MAX]; int* status[CHOL_STEP_BATCH_MAX];
  float* L32[CHOL_STEP_BATCH_MAX];
  int TvA[CHOL_STEP_BATCH_MAX], TvB[CHOL_STEP_BATCH_MAX], TvX[CHOL_STEP_BATCH_MAX];      // virtual sizes of the step (profile), see b_decode
  int nP[CHOL_STEP_BATCH_MAX], g0[CHOL_STEP_BATCH_MAX], g1[CHOL_STEP_BATCH_MAX], nX[CHOL_STEP_BATCH_MAX], a_split[CHOL_STEP_BATCH_MAX];
  int nbr[CHOL_STEP_BATCH_MAX];             // border row tiles below the profile (b_decode)
  int nbA[CHOL_STEP_BATCH_MAX], nbB[CHOL_STEP_BATCH_MAX], nbX[CHOL_STEP_BATCH_MAX];      // ... of which active for column k / the pair's pass / the column items
  int B0[CHOL_STEP_BATCH_MAX];              // tile row at which the border rows start (T, or further down for a segment view: CholSystem::b0)
  const int* ord[CHOL_STEP_BATCH_MAX];      // order of the active border rows of a view (CholSystem::ord) or null
  int a_base[CHOL_STEP_BATCH_MAX + 1];      // prefix sums of the type-A workgroup counts
  int b_base[CHOL_STEP_BATCH_MAX + 1];      // prefix sums of the type-B item counts
};
// XCD-aware numbering of the type-A workgroups (round 5; cdna_hip_programming.md T1): blocks b and b + 8 share an XCD (round-robin
// placement, observed), and the type-A workgroups of ONE system all read the same diagonal tile and pending panel tile — numbered
// consecutively they sit on eight different XCDs and every L2 fetches those tiles for itself, all at the start of the launch.  The
// bijective remap below gives the blocks of one XCD a run of consecutive logical numbers, so a system's workgroups share (mostly) one
// L2.  Speed only: any placement is correct.  SLIDE_CHOL_XCD=1 turns it on; off by default: the C4 pass does not move (2.134 vs 2.134 ms).
__device__ __forceinline__ int xcd_logical(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
__global__ __launch_bounds__(512) void k_chol_step_batched(CholBatchArgs A, int k, int kb, int* __restrict__ ctr, int a_joins, int xcd_map) {
  __shared__ ALds L;
  __shared__ int s_g;
  if (blockIdx.x == 0 && threadIdx.x == 0) ctr[k + 1] = 0;
  int bid = (int)blockIdx.x;
  if (xcd_map && bid < A.a_base[A.n]) bid = xcd_logical(bid, A.a_base[A.n]);
  if (bid < A.a_base[A.n]) {
    int r = 0;
    while (bid >= A.a_base[r + 1]) ++r;
    const int local = bid - A.a_base[r], sp = A.a_split[r];
    step_type_a(A.S[r], A.ld[r], k, A.B0[r], A.TvA[r], local >> sp, sp ? (local & 1) : -1, A.Ld[r] + (size_t)k * NB * NB,
                A.Winv[r] + (size_t)k * 1024, A.status[r], L, A.L32[r], A.nbr[r], A.nbA[r], A.ord[r]);
    if (!a_joins) return;
  }
  const int nItems = A.b_base[A.n];
  if (nItems <= 0) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_g = atomicAdd(&ctr[k], 1);
    __syncthreads();
    const int g = __builtin_amdgcn_readfirstlane(s_g);
    if (g >= nItems) break;
    int r = 0;
    while (g >= A.b_base[r + 1]) ++r;
    const int gl = g - A.b_base[r], nR = A.g1[r] - A.g0[r];
    const long long nG = (long long)A.nP[r] * (A.nP[r] + 1) / 2;
    const BItem it = b_decode(gl, nR + A.nX[r], nR, A.g0[r], k, kb, A.B0[r], A.TvB[r], A.TvX[r], A.nP[r], nG, wave, A.nbr[r], A.nbB[r], A.nbX[r], A.ord[r]);
    if (!it.ok) continue;
    if (it.ks == 32) b_quadrant<32>(A.S[r], A.ld[r], it, wave & 3);
    else b_quadrant<16>(A.S[r], A.ld[r], it, wave & 3);
  }
}


// ------------------------------------------------------------------------------------------------
// PAIR kernel (round 5): TWO block columns k, k + 1 (k even) per launch — the chain-bound launch train of an exact joint pass halved.
// A launch of the step kernels above is one 64-column diagonal chain (~6.3 us) wrapped in ~3 us of dispatch gap, ~1.5 us of loads and
// staging and ~1.8 us of closing.  Here a type-A workgroup carries, beside its own 32 rows of tile row i >= k + 2, the WHOLE sub-diagonal
// tile (k + 1, k) and the diagonal block (k + 1, k + 1) redundantly (as it already does D_kk), so that the chain of column k + 1 starts
// inside the same launch the moment the last rows of L(k + 1, k) exist: no cross-workgroup hop, no kernel boundary.  Right-looking kept:
// the launch's work queue applies the panels of the PREVIOUS pair (k - 2, k - 1) to the tiles (i, j >= k + 2) in one rank-128 pass; the
// pair's own columns take those two panels as PENDING panels inside the type-A workgroups.
// Twelve waves (wave w runs on SIMD w & 3):
//   0        chain wave: the 4x4 pivot chain of D_kk, then of D(k+1, k+1) (alone on SIMD 0: waves 4 and 8 only stage and leave)
//   1, 2, 3  workers: sub-tile rows 1..3 of the diagonal block of the column at hand (a_worker_wave's protocol, flags offset per column)
//   5, 9, 6, 7   "sub" waves q = 0..3: rows 16 q .. of tile (k + 1, k): pending panels, X = A L_kk^-T pipelined with the chain's phases; every
//            finished 16-column block goes to LDS as lane images (Xs) for the other waves; the own sub-tile ROW q of D(k+1, k+1) is brought
//            up to date on the side (panels k-2, k-1 from memory, the fresh panel k block by block) and handed to factor wave q (D2) —
//            after the chain of column k only the last block's four MFMAs and the hand-over of sub-tile (0, 0) lie before the next chain
//   10, 11   "row" waves: 16 rows each of the workgroup's half of tile row i: column k like panel_rows, then tile (i, k + 1): its pending
//            panels k-2, k-1 (operands of tile (k+1, c) straight from memory), the fresh panel from Xs, X = A L(k+1,k+1)^-T with the
//            second chain's phases.
// LDS: the staged pending tiles (k, k-2), (k, k-1) (80 KB; once every wave is through with them — pend_done — their first 20 KB carry D2),
// Xs (32 KB), the factor waves' exchange areas of ALds.  All flags count upwards over the two columns (no reset, no barrier between them).
constexpr int PAIR_THREADS = 768;
struct PLds {
  double Pk[2 * NB][PSTR];  // Pk[kk][r] = L[k*NB + r][(k-2)*NB + kk]: pending panel tiles (k, k-2), (k, k-1); later D2 (see above)
  double Xs[16][4][64];     // Xs[ks][J][lane (lr, lk)] = X[16 J + lr][4 ks + lk], X = L(k+1, k)
  double mop[16][64];
  double xm[16][64];
  double Lt[6][4][64];
  double Dh[3][4][64];
  double Dh3[3][64];
  double Wi[4][256];
  int it_done;              // 1..16 column k, 17..32 column k+1
  int col_done[4];          // 1..4, then 5..8
  int d_ready[4];           // 1, then 2
  int w_done;               // 1..4, then 5..8
  int pend_done;            // waves that are through with Pk
  int p1_done;              // sub / row waves that are through with column k's Lt / Wi
  int xs_done[4];           // xs_done[q]: 16-column blocks of rows 16 q .. of X published in Xs
  int d2_ready[4];          // d2_ready[q]: sub-tile row q of D(k+1, k+1) is in D2
  int sub_loaded;           // sub waves whose loads of tile (k+1, k) have come back
  int store_flag;           // 0: not drawn yet; 1: another workgroup stores L(k+1, k); 2: this one does (the system's last ticket)
};
// Bounded flag wait (every wave of the pair kernel reaches its end whatever happens): ~0.2 s of polling, then the wait is counted in
// g_pair_timeouts (slide_debug_pair_timeouts: the tests assert 0) and the wave goes on with whatever is there.
__device__ int g_pair_timeouts;
__device__ __forceinline__ void lds_wait_p(int* f, int v) {
  const unsigned off = (unsigned)(size_t)f;
  int cur;
  for (int spin = 0; spin < (1 << 21); ++spin) {
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(cur) : "v"(off) : "memory");
    if (__builtin_amdgcn_readfirstlane(cur) >= v) return;
    __builtin_amdgcn_s_sleep(1);
  }
  if ((threadIdx.x & 63) == 0) atomicAdd(&g_pair_timeouts, 1);
}
__device__ __forceinline__ void lds_add1(int* f, int lane) {
  const unsigned off = (unsigned)(size_t)f;
  const int one = 1;
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(off), "v"(one) : "memory");
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}
__device__ __forceinline__ constexpr int didx(int I, int J) { return I * (I + 1) / 2 + J; }   // I >= J
typedef double (*D2Ptr)[4][64];
// global accesses of the pair kernel: wave-uniform base (scalar registers) + one 32-bit lane offset in bytes, so that the dozens of
// loads a wave keeps in flight share ONE address register (with 64-bit per-lane addresses they do not fit the 168 registers of
// three waves per SIMD)
// (the empty asm pins the base in a scalar register pair: without it the compiler, short of scalar registers, keeps whole families of
// 64-bit per-lane addresses alive — a tile's load addresses until its stores — and spills them)
__device__ __forceinline__ double ldu(const double* ubase, unsigned boff) {
  asm("" : "+s"(ubase));
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(ubase) + boff);
}
__device__ __forceinline__ void stu(double* ubase, unsigned boff, double v) {
  asm("" : "+s"(ubase));
  *reinterpret_cast<double*>(reinterpret_cast<char*>(ubase) + boff) = v;
}
__device__ __forceinline__ D2Ptr pair_d2(PLds& L) { return reinterpret_cast<D2Ptr>(&L.Pk[0][0]); }

// ---- wave 0 (a_chain_wave with the flags of column PH) ---------------------------------------------------------------------------------
template <int NPAN, int PH>
__device__ __forceinline__ void p_chain_wave(bool report, int* status, PLds& L, int lane, v4d Dt) {
  const int lr = lane & 15, lk = lane >> 4;
  const v4d zero = v4d{0.0, 0.0, 0.0, 0.0};
  const bool s00 = lr == 0 && lk == 0, s10 = lr == 1 && lk == 0, s11 = lr == 1 && lk == 1, s20 = lr == 2 && lk == 0,
             s21 = lr == 2 && lk == 1, s22 = lr == 2 && lk == 2, s30 = lr == 3 && lk == 0, s31 = lr == 3 && lk == 1,
             s32 = lr == 3 && lk == 2, s33 = lr == 3 && lk == 3;
  if (NPAN > 0) {
    v4d d1 = zero, d2 = zero, d3 = zero;
#pragma unroll
    for (int ks = 0; ks < 4 * NPAN; ++ks) {
      const double q0 = L.Pk[4 * ks + lk][lr], q1 = L.Pk[16 * NPAN + 4 * ks + lk][lr], q2 = L.Pk[32 * NPAN + 4 * ks + lk][lr],
                   q3 = L.Pk[48 * NPAN + 4 * ks + lk][lr];
      Dt = mfma_f64(-q0, q0, Dt);
      d1 = mfma_f64(-q1, q1, d1);
      d2 = mfma_f64(-q2, q2, d2);
      d3 = mfma_f64(-q3, q3, d3);
    }
    Dt = (Dt + d1) + (d2 + d3);
  }
  if (PH == 0) lds_add1(&L.pend_done, lane);
  bool ok = true;
  v4d Dnext = zero;
#pragma unroll
  for (int JQ = 0; JQ < 4; ++JQ) {
    if (JQ > 0) Dt = Dnext;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (lr < lk + 4 * r) Dt[r] = 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n = 4 * JQ + s;
      const double dv = Dt[s];
      const double P00 = bcast_lane(dv, 4 * s), P10 = bcast_lane(dv, 4 * s + 1), P20 = bcast_lane(dv, 4 * s + 2), P30 = bcast_lane(dv, 4 * s + 3);
      const double P11 = bcast_lane(dv, 4 * s + 1 + 16), P21 = bcast_lane(dv, 4 * s + 2 + 16), P31 = bcast_lane(dv, 4 * s + 3 + 16);
      const double P22 = bcast_lane(dv, 4 * s + 2 + 32), P32 = bcast_lane(dv, 4 * s + 3 + 32), P33 = bcast_lane(dv, 4 * s + 3 + 48);
      const double i0 = rsqrt_nr(P00);
      const double l10 = P10 * i0, l20 = P20 * i0, l30 = P30 * i0;
      const double d1 = P11 - l10 * l10;
      const double i1 = rsqrt_nr(d1);
      const double l21 = (P21 - l20 * l10) * i1, l31 = (P31 - l30 * l10) * i1;
      const double d2 = P22 - l20 * l20 - l21 * l21;
      const double i2 = rsqrt_nr(d2);
      const double l32 = (P32 - l30 * l20 - l31 * l21) * i2;
      const double d3 = P33 - l30 * l30 - l31 * l31 - l32 * l32;
      const double i3 = rsqrt_nr(d3);
      ok = ok && (P00 > 0.0) && (d1 > 0.0) && (d2 > 0.0) && (d3 > 0.0);
      const double m10 = -l10 * i0 * i1;
      const double m21 = -l21 * i1 * i2;
      const double m20 = -(l20 * i0 + l21 * m10) * i2;
      const double m32 = -l32 * i2 * i3;
      const double m31 = -(l31 * i1 + l32 * m21) * i3;
      const double m30 = -(l30 * i0 + l31 * m10 + l32 * m20) * i3;
      double mop = 0.0;
      mop = s00 ? i0 : mop;  mop = s11 ? i1 : mop;  mop = s10 ? m10 : mop;  mop = s22 ? i2 : mop;  mop = s21 ? m21 : mop;
      mop = s20 ? m20 : mop; mop = s33 ? i3 : mop;  mop = s32 ? m32 : mop;  mop = s31 ? m31 : mop;  mop = s30 ? m30 : mop;
      L.mop[n][lane] = mop;
      const double xd = mfma_f64(mop, Dt[s], zero)[0];
      Dt[s] = xd;
      const double xm = (lr > 4 * s + 3) ? xd : 0.0;
      L.xm[n][lane] = xm;
      lds_post(&L.it_done, 16 * PH + n + 1, lane);
      if (s < 3) Dt = mfma_f64(-xm, xm, Dt);
      if (s == 3 && JQ < 3) {
        lds_wait_p(&L.d_ready[JQ + 1], PH + 1);
        const double x3 = mfma_f64(mop, L.Dh3[JQ][lane], zero)[0];
#pragma unroll
        for (int r = 0; r < 4; ++r) Dnext[r] = L.Dh[JQ][r][lane];
        Dnext = mfma_f64(-x3, x3, Dnext);
      }
    }
  }
  if (!ok && lane == 0 && report) atomicOr(&status[1], 1);
}

// ---- waves 1..3 (a_worker_wave with the flags of column PH; NREAD: the sub / row waves that read column k's Lt / Wi) -----------------------
template <int W, int NPAN, int PH>
__device__ __forceinline__ void p_worker_wave(bool store, double* __restrict__ Ld, double* __restrict__ Winv, PLds& L, int lane,
                                              v4d (&R)[W + 1], int nread) {
  const int lr = lane & 15, lk = lane >> 4;
  const v4d zero = v4d{0.0, 0.0, 0.0, 0.0};
  if (NPAN > 0) {
    v4d e[W + 1];
#pragma unroll
    for (int J = 0; J <= W; ++J) e[J] = zero;
#pragma unroll
    for (int ks = 0; ks < 16 * NPAN; ks += 2) {
      const double w0 = L.Pk[4 * ks + lk][16 * W + lr], w1 = L.Pk[4 * ks + 4 + lk][16 * W + lr];
#pragma unroll
      for (int J = 0; J <= W; ++J) {
        R[J] = mfma_f64(-L.Pk[4 * ks + lk][16 * J + lr], w0, R[J]);
        e[J] = mfma_f64(-L.Pk[4 * ks + 4 + lk][16 * J + lr], w1, e[J]);
      }
    }
#pragma unroll
    for (int J = 0; J <= W; ++J) {
      R[J] += e[J];
      pin(R[J]);
    }
  }
  if (PH == 0) lds_add1(&L.pend_done, lane);
  v4d Wt = zero;
#pragma unroll
  for (int JQ = 0; JQ < W; ++JQ) {
    if (W == WT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Wt[r] = (lr == lk + 4 * r) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n = 4 * JQ + s;
      lds_wait_p(&L.it_done, 16 * PH + n + 1);
      const double mop = L.mop[n][lane], xm = L.xm[n][lane];
      const double x = mfma_f64(mop, R[JQ][s], zero)[0];
      double xw = 0.0;
      if (W == WT) xw = mfma_f64(mop, Wt[s], zero)[0];
      R[JQ][s] = x;
      if (W == WT) Wt[s] = xw;
      if (s < 3) {
        R[JQ] = mfma_f64(-xm, x, R[JQ]);
        if (W == WT) Wt = mfma_f64(-xm, xw, Wt);
      }
      R[W] = mfma_f64(-R[JQ][s], R[JQ][s], R[W]);
#pragma unroll
      for (int J = 0; J <= W; ++J) pin(R[J]);
      if (W == WT) pin(Wt);
      if (JQ == W - 1 && s == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) L.Dh[W - 1][r][lane] = R[W][r];
        L.Dh3[W - 1][lane] = R[JQ][3];
        lds_post(&L.d_ready[W], PH + 1, lane);
      }
    }
    // the second column overwrites Lt / Wi: not before every reader of the first column's is through (long since: they finish right
    // behind the first chain)
    if (PH == 1 && JQ == 0) lds_wait_p(&L.p1_done, nread);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      L.Lt[oidx(W, JQ)][r][lane] = R[JQ][r];
      if (store) Ld[(size_t)(16 * JQ + lk + 4 * r) * NB + 16 * W + lr] = R[JQ][r];
    }
    lds_post(&L.col_done[W], 4 * PH + JQ + 1, lane);
#pragma unroll
    for (int J = JQ + 1; J < W; ++J) {
      lds_wait_p(&L.col_done[J], 4 * PH + JQ + 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) R[J] = mfma_f64(-L.Lt[oidx(J, JQ)][ks][lane], R[JQ][ks], R[J]);
    }
    if (W == WT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = (lk + 4 * r >= lr) ? Wt[r] : 0.0;
        L.Wi[JQ][lr * 16 + lk + 4 * r] = v;
        if (store) Winv[(size_t)JQ * 256 + lr * 16 + lk + 4 * r] = v;
      }
      lds_post(&L.w_done, 4 * PH + JQ + 1, lane);
    }
  }
  static_assert(W >= 1 && W <= 3, "worker index");
  if (W == WT) {
#pragma unroll
    for (int JQ = WT; JQ < 4; ++JQ) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Wt[r] = (lr == lk + 4 * r) ? 1.0 : 0.0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int n = 4 * JQ + s;
        lds_wait_p(&L.it_done, 16 * PH + n + 1);
        const double mop = L.mop[n][lane], xm = L.xm[n][lane];
        const double xw = mfma_f64(mop, Wt[s], zero)[0];
        Wt[s] = xw;
        if (s < 3) Wt = mfma_f64(-xm, xw, Wt);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = (lk + 4 * r >= lr) ? Wt[r] : 0.0;
        L.Wi[JQ][lr * 16 + lk + 4 * r] = v;
        if (store) Winv[(size_t)JQ * 256 + lr * 16 + lk + 4 * r] = v;
      }
      lds_post(&L.w_done, 4 * PH + JQ + 1, lane);
    }
  }
}

#define PAIR_YIELD() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(1); __builtin_amdgcn_sched_barrier(0); } while (0)
// ---- sub wave Q: rows 16 Q .. of tile (k + 1, k) and sub-tile row Q of D(k+1, k+1) ----------------------------------------------------------
// Tq[b][r] = A[16 Q + lr][16 b + lk + 4 r] (tile (k+1, k)), tb[ks] = own rows of the pending panel tiles (k+1, k-2), (k+1, k-1),
// R2[J][r] = D(k+1,k+1)[16 Q + lr][16 J + lk + 4 r]
// L(k+1, k) is needed, as the ORIGINAL tile, by every workgroup of the system and written back by one of them: the one that draws the
// system's LAST ticket (`ticket`: a device counter; every workgroup draws once all its loads of the tile are back, so whoever draws
// n_wg - 1 knows that nobody reads the tile any more — workgroups of a later round of the launch included; nobody waits for anybody).
template <int NPAN, int Q>
__device__ __forceinline__ void p_sub_rows(double* __restrict__ S, int ld, int k, bool zs, int* ticket, int n_wg, int lane, PLds& L, v4d (&Tq)[4],
                                           const double (&tb)[NPAN > 0 ? 16 * NPAN : 1], v4d (&R2)[Q + 1], int nread_pk) {
  const int lr = lane & 15, lk = lane >> 4;
  const v4d zero = v4d{0.0, 0.0, 0.0, 0.0};
  const unsigned boff = (unsigned)(lk * ld + lr) * 8u;
  double* tcol = S + (size_t)(k * NB) * ld + (size_t)(k + 1) * NB + 16 * Q;      // (uniform) own rows of tile (k+1, k)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the own rows of the tile (and everything else this wave loads) are in registers
  lds_add1(&L.sub_loaded, lane);
  if (Q == 3) {
    lds_wait_p(&L.sub_loaded, 4);
    int flag = 1;
    if (lane == 0) {
      const int old = atomicAdd(ticket, 1);
      if (old == n_wg - 1) { flag = 2; atomicExch(ticket, 0); }      // (the last: everybody has drawn; the counter is ready for the next launch)
    }
    flag = __builtin_amdgcn_readfirstlane(flag);
    lds_post(&L.store_flag, flag, lane);
  }
  if (NPAN > 0) {
    // pending panels on the own rows of the tile: A = rows of (k, c) (LDS), B = own rows of (k+1, c)
#pragma unroll
    for (int ks = 0; ks < 16 * NPAN; ++ks) {
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) Tq[bb] = mfma_f64(-L.Pk[4 * ks + lk][16 * bb + lr], tb[ks], Tq[bb]);
    }
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) pin(Tq[bb]);
  }
  lds_add1(&L.pend_done, lane);
  if (NPAN > 0) {
    // ... and on the own sub-tile row of D(k+1, k+1): A = rows 16 J + lr of (k+1, c): the own rows are tb, the others come from memory
    // (the sibling sub waves have just loaded them: L1 / L2 hits), a few k-steps at a time
    const double* prow = S + (size_t)((k - NPAN) * NB) * ld + (size_t)(k + 1) * NB;
    constexpr int CH = Q <= 1 ? 8 : 4;
#pragma unroll
    for (int k0 = 0; k0 < 16 * NPAN; k0 += CH) {
      double a[Q > 0 ? Q : 1][CH];
#pragma unroll
      for (int J = 0; J < Q; ++J)
#pragma unroll
        for (int i = 0; i < CH; ++i) a[J][i] = ldu(prow + (size_t)(4 * (k0 + i)) * ld + 16 * J, boff);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
#pragma unroll
        for (int J = 0; J < Q; ++J) R2[J] = mfma_f64(-a[J][i], tb[k0 + i], R2[J]);
        R2[Q] = mfma_f64(-tb[k0 + i], tb[k0 + i], R2[Q]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int J = 0; J <= Q; ++J) pin(R2[J]);
  }
  // X = A L_kk^-T by blocked substitution with the first chain's phases (panel_rows); every finished block feeds Xs and D2's row
  v4d xt[4], t[4];
  lds_wait_p(&L.it_done, 4);
  t[0] = Tq[0];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    lds_wait_p(&L.w_done, b + 1);
    v4d x = zero;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) x = mfma_f64(L.Wi[b][(4 * s4 + lk) * 16 + lr], t[b][s4], x);
    xt[b] = x;
#pragma unroll
    for (int r = 0; r < 4; ++r) L.Xs[4 * b + r][Q][lane] = x[r];
    lds_post(&L.xs_done[Q], b + 1, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) R2[Q] = mfma_f64(-x[r], x[r], R2[Q]);
    if (Q == 0 && b == 3) {
      // sub-tile (0, 0) of the next diagonal block: the second chain starts from it
      lds_wait_p(&L.pend_done, nread_pk);
      D2Ptr D2 = pair_d2(L);
#pragma unroll
      for (int r = 0; r < 4; ++r) D2[didx(0, 0)][r][lane] = R2[0][r];
      lds_post(&L.d2_ready[0], 1, lane);
    }
#pragma unroll
    for (int J = 0; J < Q; ++J) {
      lds_wait_p(&L.xs_done[J], b + 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) R2[J] = mfma_f64(-L.Xs[4 * b + r][J][lane], x[r], R2[J]);
    }
    if (b < 3) {
      PAIR_YIELD();
      lds_wait_p(&L.col_done[b + 1], b + 1);
      v4d tn = Tq[b + 1];
#pragma unroll
      for (int c = 0; c <= b; ++c) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) tn = mfma_f64(-L.Lt[oidx(b + 1, c)][s4][lane], xt[c][s4], tn);
        if (b < 2 || c < b) PAIR_YIELD();
      }
      t[b + 1] = tn;
      pin(t[b + 1]);
    }
  }
  if (Q > 0) {
    lds_wait_p(&L.pend_done, nread_pk);
    D2Ptr D2 = pair_d2(L);
#pragma unroll
    for (int J = 0; J <= Q; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) D2[didx(Q, J)][r][lane] = R2[J][r];
    lds_post(&L.d2_ready[Q], 1, lane);
  }
  lds_add1(&L.p1_done, lane);
  // the tile goes back to memory from the workgroup that drew the last ticket
  lds_wait_p(&L.store_flag, 1);
  int sf;
  {
    const unsigned off = (unsigned)(size_t)&L.store_flag;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(sf) : "v"(off) : "memory");
  }
  if (__builtin_amdgcn_readfirstlane(sf) == 2 && !zs) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) stu(tcol + (size_t)(16 * b + 4 * r) * ld, boff, xt[b][r]);
  }
}

// ---- row wave: 16 rows (row set qq) of tile row `it`: column k, then column k + 1 -----------------------------------------------------------
template <int NPAN>
__device__ __forceinline__ void p_row_rows(double* __restrict__ S, int ld, int k, int it, int qq, int ncols, bool z0, int lane, PLds& L,
                                           v4d (&Tq)[4], const double (&tb)[NPAN > 0 ? 16 * NPAN : 1], v4d (&Tq2)[4]) {
  const int lr = lane & 15, lk = lane >> 4;
  const v4d zero = v4d{0.0, 0.0, 0.0, 0.0};
  const unsigned boff = (unsigned)(lk * ld + lr) * 8u;
  double* tcol = S + (size_t)(k * NB) * ld + (size_t)it * NB + 16 * qq;
  if (NPAN > 0 && !z0) {
#pragma unroll
    for (int ks = 0; ks < 16 * NPAN; ++ks) {
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) Tq[bb] = mfma_f64(-L.Pk[4 * ks + lk][16 * bb + lr], tb[ks], Tq[bb]);
    }
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) pin(Tq[bb]);
  }
  lds_add1(&L.pend_done, lane);
  if (NPAN > 0 && !z0 && ncols == 2) {
    // the same two panels on the own rows of tile (it, k + 1): A = rows of tile (k + 1, c), straight from memory
    const double* prow = S + (size_t)((k - NPAN) * NB) * ld + (size_t)(k + 1) * NB;
#pragma unroll
    for (int k0 = 0; k0 < 16 * NPAN; k0 += 2) {
      double a[4][2];
#pragma unroll
      for (int bb = 0; bb < 4; ++bb)
#pragma unroll
        for (int i = 0; i < 2; ++i) a[bb][i] = ldu(prow + (size_t)(4 * (k0 + i)) * ld + 16 * bb, boff);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) Tq2[bb] = mfma_f64(-a[bb][i], tb[k0 + i], Tq2[bb]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) pin(Tq2[bb]);
  }
  v4d xt[4], t[4];
  if (!z0) {
    lds_wait_p(&L.it_done, 4);
    t[0] = Tq[0];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      lds_wait_p(&L.w_done, b + 1);
      v4d x = zero;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) x = mfma_f64(L.Wi[b][(4 * s4 + lk) * 16 + lr], t[b][s4], x);
      xt[b] = x;
#pragma unroll
      for (int r = 0; r < 4; ++r) stu(tcol + (size_t)(16 * b + 4 * r) * ld, boff, x[r]);
      if (ncols == 2) {
        // the fresh panel on the own rows of tile (it, k + 1): A = rows of X = L(k+1, k) from Xs
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          lds_wait_p(&L.xs_done[bb], b + 1);
#pragma unroll
          for (int r = 0; r < 4; ++r) Tq2[bb] = mfma_f64(-L.Xs[4 * b + r][bb][lane], x[r], Tq2[bb]);
        }
      }
      if (b < 3) {
        PAIR_YIELD();
        lds_wait_p(&L.col_done[b + 1], b + 1);
        v4d tn = Tq[b + 1];
#pragma unroll
        for (int c = 0; c <= b; ++c) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) tn = mfma_f64(-L.Lt[oidx(b + 1, c)][s4][lane], xt[c][s4], tn);
          if (b < 2 || c < b) PAIR_YIELD();
        }
        t[b + 1] = tn;
        pin(t[b + 1]);
      }
    }
  }
  lds_add1(&L.p1_done, lane);
  if (ncols != 2) return;
  // column k + 1 with the second chain's phases
  double* tcol2 = S + (size_t)((k + 1) * NB) * ld + (size_t)it * NB + 16 * qq;
  t[0] = Tq2[0];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    lds_wait_p(&L.w_done, 4 + b + 1);
    v4d x = zero;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) x = mfma_f64(L.Wi[b][(4 * s4 + lk) * 16 + lr], t[b][s4], x);
    xt[b] = x;
#pragma unroll
    for (int r = 0; r < 4; ++r) stu(tcol2 + (size_t)(16 * b + 4 * r) * ld, boff, x[r]);
    if (b < 3) {
      PAIR_YIELD();
      lds_wait_p(&L.col_done[b + 1], 4 + b + 1);
      v4d tn = Tq2[b + 1];
#pragma unroll
      for (int c = 0; c <= b; ++c) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) tn = mfma_f64(-L.Lt[oidx(b + 1, c)][s4][lane], xt[c][s4], tn);
        if (b < 2 || c < b) PAIR_YIELD();
      }
      t[b + 1] = tn;
      pin(t[b + 1]);
    }
  }
}

// own rows (row set q of tile row `it`) of block column kc and of the NPAN panel tiles in front of column k
template <int NPAN>
__device__ __forceinline__ void pair_load_rows(const double* __restrict__ S, int ld, int k, int kc, int it, int q, int lr, int lk, bool zero_rows,
                                               v4d (&Tq)[4], double (&tb)[NPAN > 0 ? 16 * NPAN : 1], bool with_tb) {
  const unsigned boff = (unsigned)(lk * ld + lr) * 8u;
  const double* tcol = S + (size_t)(kc * NB) * ld + (size_t)it * NB + 16 * q;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) Tq[b][r] = zero_rows ? 0.0 : ldu(tcol + (size_t)(16 * b + 4 * r) * ld, boff);
  if (NPAN > 0 && with_tb) {
    const double* pi = S + (size_t)((k - NPAN) * NB) * ld + (size_t)it * NB + 16 * q;
#pragma unroll
    for (int ks = 0; ks < 16 * NPAN; ++ks) tb[ks] = zero_rows ? 0.0 : ldu(pi + (size_t)(4 * ks) * ld, boff);
  }
}

template <int NPAN, int Q>
__device__ __forceinline__ void pair_sub_wave(double* __restrict__ S, int ld, int k, bool zs, int* ticket, int n_wg, int lane, PLds& L, int nread_pk) {
  const int lr = lane & 15, lk = lane >> 4;
  v4d Tq[4];
  double tb[NPAN > 0 ? 16 * NPAN : 1];
  pair_load_rows<NPAN>(S, ld, k, k, k + 1, Q, lr, lk, false, Tq, tb, true);
  v4d R2[Q + 1];
  const unsigned boff = (unsigned)(lk * ld + lr) * 8u;
  const double* dcol = S + (size_t)((k + 1) * NB) * ld + (size_t)(k + 1) * NB + 16 * Q;
#pragma unroll
  for (int J = 0; J <= Q; ++J)
#pragma unroll
    for (int r = 0; r < 4; ++r) R2[J][r] = ldu(dcol + (size_t)(16 * J + 4 * r) * ld, boff);
  __syncthreads();
  p_sub_rows<NPAN, Q>(S, ld, k, zs, ticket, n_wg, lane, L, Tq, tb, R2, nread_pk);
}

// One type-A workgroup of the pair (k, k + 1): rows 32 half .. of tile row `it` (+ everything of the diagonal blocks).  ncols = 1: the
// system's last column alone (no sub waves, one chain).  z0: tile (it, k) is structurally zero (the row starts at column k + 1: nothing
// is loaded or stored there); zs: so is tile (k + 1, k) (not stored).  first: this workgroup publishes Ld / Winv / the status flag.
template <int NPAN>
__device__ __forceinline__ void pair_type_a(double* __restrict__ S, int ld, int k, int ncols, int it, int half, bool first, bool z0, bool zs,
                                            double* __restrict__ Ld, double* __restrict__ Winv, int* status, int* ticket, int n_wg, PLds& L) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..11
  const int lr = lane & 15, lk = lane >> 4;
  if (tid == 0) { L.it_done = 0; L.w_done = 0; L.pend_done = 0; L.p1_done = 0; L.sub_loaded = 0; L.store_flag = 0; }
  if (tid < 4) { L.col_done[tid] = 0; L.d_ready[tid] = 0; L.xs_done[tid] = 0; L.d2_ready[tid] = 0; }
  const int nread_pk = ncols == 2 ? 10 : 6;      // waves that read Pk: four factor waves, two row waves (+ four sub waves)
  const int nread_lt = 6;                        // sub + row waves that read the first column's Lt / Wi
  if (NPAN > 0 && wave >= 4) {
    // a sixteenth of the pending panel tiles (k, k-2), (k, k-1) per wave -> LDS
    const double* pq = S + (size_t)((k - NPAN) * NB + 16 * (wave - 4)) * ld + (size_t)k * NB;
    double stage[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) stage[c] = ldu(pq + (size_t)c * ld, (unsigned)lane * 8u);
#pragma unroll
    for (int c = 0; c < 16; ++c) L.Pk[16 * (wave - 4) + c][lane] = stage[c];
  }
  if (wave < 4) {
    const unsigned boff = (unsigned)(lk * ld + lr) * 8u;
    const double* dcol = S + (size_t)(k * NB) * ld + (size_t)k * NB + 16 * wave;
    v4d R[4];
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) R[J][r] = (J <= wave) ? ldu(dcol + (size_t)(16 * J + 4 * r) * ld, boff) : 0.0;
    __syncthreads();
    if (wave == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
    D2Ptr D2 = pair_d2(L);
    if (wave == 0) {
      p_chain_wave<NPAN, 0>(first, status, L, lane, R[0]);
      if (ncols == 2) {
        lds_wait_p(&L.d2_ready[0], 1);
        v4d D;
#pragma unroll
        for (int r = 0; r < 4; ++r) D[r] = D2[didx(0, 0)][r][lane];
        p_chain_wave<0, 1>(first, status, L, lane, D);
      }
    } else if (wave == 1) {
      v4d R1[2] = {R[0], R[1]};
      p_worker_wave<1, NPAN, 0>(first, Ld, Winv, L, lane, R1, nread_lt);
      if (ncols == 2) {
        lds_wait_p(&L.d2_ready[1], 1);
#pragma unroll
        for (int J = 0; J <= 1; ++J)
#pragma unroll
          for (int r = 0; r < 4; ++r) R1[J][r] = D2[didx(1, J)][r][lane];
        p_worker_wave<1, 0, 1>(first, Ld + NB * NB, Winv + 1024, L, lane, R1, nread_lt);
      }
    } else if (wave == 2) {
      v4d R2[3] = {R[0], R[1], R[2]};
      p_worker_wave<2, NPAN, 0>(first, Ld, Winv, L, lane, R2, nread_lt);
      if (ncols == 2) {
        lds_wait_p(&L.d2_ready[2], 1);
#pragma unroll
        for (int J = 0; J <= 2; ++J)
#pragma unroll
          for (int r = 0; r < 4; ++r) R2[J][r] = D2[didx(2, J)][r][lane];
        p_worker_wave<2, 0, 1>(first, Ld + NB * NB, Winv + 1024, L, lane, R2, nread_lt);
      }
    } else {
      p_worker_wave<3, NPAN, 0>(first, Ld, Winv, L, lane, R, nread_lt);
      if (ncols == 2) {
        lds_wait_p(&L.d2_ready[3], 1);
#pragma unroll
        for (int J = 0; J <= 3; ++J)
#pragma unroll
          for (int r = 0; r < 4; ++r) R[J][r] = D2[didx(3, J)][r][lane];
        p_worker_wave<3, 0, 1>(first, Ld + NB * NB, Winv + 1024, L, lane, R, nread_lt);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  } else if (wave == 10 || wave == 11) {
    const int qq = 2 * half + (wave - 10);
    v4d Tq[4], Tq2[4];
    double tb[NPAN > 0 ? 16 * NPAN : 1];
    pair_load_rows<NPAN>(S, ld, k, k, it, qq, lr, lk, z0, Tq, tb, true);
    if (ncols == 2) pair_load_rows<NPAN>(S, ld, k, k + 1, it, qq, lr, lk, false, Tq2, tb, false);
    __syncthreads();
    p_row_rows<NPAN>(S, ld, k, it, qq, ncols, z0, lane, L, Tq, tb, Tq2);
  } else if (ncols == 2 && wave == 5) {
    pair_sub_wave<NPAN, 0>(S, ld, k, zs, ticket, n_wg, lane, L, nread_pk);
  } else if (ncols == 2 && wave == 9) {
    pair_sub_wave<NPAN, 1>(S, ld, k, zs, ticket, n_wg, lane, L, nread_pk);
  } else if (ncols == 2 && wave == 6) {
    pair_sub_wave<NPAN, 2>(S, ld, k, zs, ticket, n_wg, lane, L, nread_pk);
  } else if (ncols == 2 && wave == 7) {
    pair_sub_wave<NPAN, 3>(S, ld, k, zs, ticket, n_wg, lane, L, nread_pk);
  } else {
    __syncthreads();      // waves 4 and 8 (and the sub waves of a single column): staging only
  }
}
#undef PAIR_YIELD

// Two block columns (k, k + 1) of up to 32 systems in ONE launch: the type-A workgroups of every system (two per tile row i >= k + ncols),
// then one work queue over the rank-128 items of all of them (panels k-2, k-1 onto the tiles (i, j >= k + 2): b_decode with pair base k + 1
// and the panel columns moved one to the left).
struct CholPairArgs {
  int n;
  double* S[CHOL_STEP_BATCH_MAX]; int ld[CHOL_STEP_BATCH_MAX];
  double* Ld[CHOL_STEP_BATCH_MAX]; double* Winv[CHOL_STEP_BATCH_MAX]; int* status[CHOL_STEP_BATCH_MAX];
  int ncols[CHOL_STEP_BATCH_MAX];           // block columns of this launch that exist in the system: 2, 1 (its last column) or 0
  int Tv0[CHOL_STEP_BATCH_MAX], Tv1[CHOL_STEP_BATCH_MAX];      // profile of column k / k + 1: rows below hold nothing
  int nb0[CHOL_STEP_BATCH_MAX], nb1[CHOL_STEP_BATCH_MAX];      // active border rows at column k / k + 1
  int TvB[CHOL_STEP_BATCH_MAX], nbB[CHOL_STEP_BATCH_MAX];      // the pass of panels k-2, k-1: profile of panel k-1, active border rows
  int nP[CHOL_STEP_BATCH_MAX];              // 2x2 tile groups per side of that pass
  int nbr[CHOL_STEP_BATCH_MAX]; int B0[CHOL_STEP_BATCH_MAX]; const int* ord[CHOL_STEP_BATCH_MAX];
  int a_base[CHOL_STEP_BATCH_MAX + 1];      // prefix sums of the type-A workgroup counts
  int b_base[CHOL_STEP_BATCH_MAX + 1];      // prefix sums of the item counts
};
// tickets: one int per system, zero between launches (p_sub_rows)
__global__ __launch_bounds__(PAIR_THREADS) void k_chol_pair_batched(CholPairArgs A, int k, int* __restrict__ ctr, int* tickets, int a_joins) {
  __shared__ PLds L;
  __shared__ int s_g;
  if (blockIdx.x == 0 && threadIdx.x == 0) ctr[k + 2] = 0;
  const int bid = (int)blockIdx.x;
  if (bid < A.a_base[A.n]) {
    int r = 0;
    while (bid >= A.a_base[r + 1]) ++r;
    const int local = bid - A.a_base[r], ia = local >> 1, half = local & 1;
    const int ncols = A.ncols[r], rbase = k + ncols;
    const int TvL = ncols == 2 ? A.Tv1[r] : A.Tv0[r], nbL = ncols == 2 ? A.nb1[r] : A.nb0[r];
    const int vi = rbase + ia - TvL;      // band rows rbase .. TvL - 1, then the active border rows, then the right-hand-side row
    const int it = vi < 0 ? rbase + ia : (vi < nbL ? A.B0[r] + (A.ord[r] ? A.ord[r][vi] : vi) : A.B0[r] + A.nbr[r]);
    const bool z0 = ncols == 2 && (vi < 0 ? rbase + ia >= A.Tv0[r] : (vi < nbL && vi >= A.nb0[r]));
    const bool zs = k + 1 >= A.Tv0[r];
    double* Ldk = A.Ld[r] + (size_t)k * NB * NB;
    double* Wik = A.Winv[r] + (size_t)k * 1024;
    const int n_wg = A.a_base[r + 1] - A.a_base[r];
    if (k >= 2) pair_type_a<2>(A.S[r], A.ld[r], k, ncols, it, half, ia == 0, z0, zs, Ldk, Wik, A.status[r], tickets + r, n_wg, L);
    else pair_type_a<0>(A.S[r], A.ld[r], k, ncols, it, half, ia == 0, z0, zs, Ldk, Wik, A.status[r], tickets + r, n_wg, L);
    if (!a_joins) return;
  }
  const int nItems = A.b_base[A.n];
  if (nItems <= 0) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_g = atomicAdd(&ctr[k], 1);
    __syncthreads();
    const int g = __builtin_amdgcn_readfirstlane(s_g);
    if (g >= nItems) break;
    if (wave >= 8) continue;
    int r = 0;
    while (g >= A.b_base[r + 1]) ++r;
    const int gl = g - A.b_base[r], nR = A.b_base[r + 1] - A.b_base[r];
    const long long nG = (long long)A.nP[r] * (A.nP[r] + 1) / 2;
    BItem it = b_decode(gl, nR, nR, 0, k, k + 1, A.B0[r], A.TvB[r], 0, A.nP[r], nG, wave, A.nbr[r], A.nbB[r], 0, A.ord[r]);
    if (!it.ok) continue;
    it.pcb = k - 2;
    b_quadrant<32>(A.S[r], A.ld[r], it, wave & 3);
  }
}


// ------------------------------------------------------------------------------------------------
// LEFT-LOOKING PERSISTENT factorisation: ALL block columns of up to 64 systems in ONE launch (round 4).
// The step kernels above pay a kernel boundary per block column: ~3 us of dispatch gap, ~1.5 us of loads and staging, ~1.8 us of
// closing on top of the ~8 us chain of a 64-column diagonal block — and an exact joint pass runs 58 such launches in a row.  Here the
// boundary is a FLAG.  The work is cut into tasks, one workgroup each, started in ticket order (an atomic counter: a task only ever
// waits for tasks with lower tickets, i.e. for workgroups that have started — no assumption about the dispatcher):
//   chain task (k):    the type-A workgroup of the step kernel for the first tile row of column k — pending panel (k, k-1) through LDS,
//                      the diagonal block's chain on wave 0, its sub-tile rows on waves 1-3, the tile's rows on the panel waves —
//                      preceded by the LEFT-LOOKING sum over the older panels c < k-1 (operands straight from L2: they were final
//                      long ago and are summed while the chain task of column k-1 still runs).  Publishes L_kk's pieces (Ld, Winv)
//                      and its tile.
//   tile task (k, i):  one or two further tile rows of column k, eight waves of sixteen rows: A(i, k) - sum_c L(i, c) L(k, c)^T over the
//                      columns c in which both are non-zero (ready except for the last one or two), then, once L_kk is published,
//                      X = A L_kk^-T by blocked substitution with the 16x16 inverses — the arithmetic of panel_rows.
// Every tile is written exactly once, by its task, and read by others only behind its flag: producers close with a release fence at
// agent scope (L2 write-back towards the memory side: the XCDs' L2s are not coherent with each other) before one lane raises the flag
// (relaxed agent-scope atomic), consumers poll the flags (agent-scope atomic loads, a lane per pending column) and pass an acquire
// fence before they read — the message-passing pattern of the AMDGPU memory model; tools/tile_hop_bench.hip measures the hand-over
// and counts stale values per protocol.  No trailing updates are written back at all (the right-looking flood of the step kernels
// re-reads and re-writes every trailing tile once per two columns).
// Exit condition every wave reaches: polls are bounded; a workgroup that gives up raises status bit 2 and the launch's abort word,
// which every other poll loop checks.
struct LLSys { double* S; double* Ld; double* Winv; int* status; int* flags; int ld, T, B0, nbr, frows; };
struct LLTask { int sys, k, kind, it0, it1, clo_d, clo0, clo1; };      // kind 0: chain task (tile row it0), 1: tile task (rows it0, it1 or -1), 2: follower chain task (row it0; factors D_k for itself)
constexpr int LL_SPIN_MAX = 1 << 19;

// Copies of a_worker_wave / panel_rows for the persistent kernel (the step kernels' own stay byte for byte what round 3 measured: their
// code is sensitive to the compiler's scheduling).  The only difference: results other workgroups read inside the SAME launch — the
// panel tile, Ld, Winv — are stored write-through at agent scope (global_store .. sc1), so that the release fence in front of the flag has
// nothing left to write back (tools/tile_hop_bench.hip: 1.16 us for a tile's stores + fence against 1.44 us with plain stores, and the
// stores of the early column blocks are long through by then).
__device__ __forceinline__ void st_wt(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int W, int NPAN>
__device__ __forceinline__ void ll_worker_wave(int ia, double* __restrict__ Ld, double* __restrict__ Winv, ALds& L, int lane,
                                              v4d (&R)[W + 1]) {
  const int lr = lane & 15, lk = lane >> 4;
  const v4d zero = v4d{0.0, 0.0, 0.0, 0.0};
  if (NPAN > 0) {
    // the whole sub-tile row before the first iteration, in the order of need: a worker answers an iteration of the
    // chain wave in about 500 cycles but needs twice that with a quarter of a sub-tile update on top, so it is better
    // late for the first iterations (it catches up well before its hand-off) than slow in all of them
    // (a chain of dependent MFMAs fed from LDS runs at ~150 cycles per link: 2 (W + 1) independent chains, k-steps outermost)
    v4d e[W + 1];
#pragma unroll
    for (int J = 0; J <= W; ++J) e[J] = zero;
#pragma unroll
    for (int ks = 0; ks < 16 * NPAN; ks += 2) {
      const double w0 = L.Pk[4 * ks + lk][16 * W + lr], w1 = L.Pk[4 * ks + 4 + lk][16 * W + lr];
#pragma unroll
      for (int J = 0; J <= W; ++J) {
        R[J] = mfma_f64(-L.Pk[4 * ks + lk][16 * J + lr], w0, R[J]);
        e[J] = mfma_f64(-L.Pk[4 * ks + 4 + lk][16 * J + lr], w1, e[J]);
      }
    }
#pragma unroll
    for (int J = 0; J <= W; ++J) {
      R[J] += e[J];
      pin(R[J]);
    }
  }
  v4d Wt = zero;                   // identity pseudo-tile (worker WT only)
#pragma unroll
  for (int JQ = 0; JQ < W; ++JQ) {
    if (W == WT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Wt[r] = (lr == lk + 4 * r) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int n = 4 * JQ + s;
      lds_wait(&L.it_done, n + 1);
      const double mop = L.mop[n][lane], xm = L.xm[n][lane];
      const double x = mfma_f64(mop, R[JQ][s], zero)[0];
      double xw = 0.0;
      if (W == WT) xw = mfma_f64(mop, Wt[s], zero)[0];
      R[JQ][s] = x;
      if (W == WT) Wt[s] = xw;
      if (s < 3) {
        R[JQ] = mfma_f64(-xm, x, R[JQ]);
        if (W == WT) Wt = mfma_f64(-xm, xw, Wt);
      }
      // the rank-16 update of the diagonal sub-tile one k-step at a time: register s of (W, JQ) is final from here on
      R[W] = mfma_f64(-R[JQ][s], R[JQ][s], R[W]);
#pragma unroll
      for (int J = 0; J <= W; ++J) pin(R[J]);
      if (W == WT) pin(Wt);
      if (JQ == W - 1 && s == 2) {
        // hand-off of the own diagonal sub-tile one iteration early: everything but the last iteration's rank-4 update is in it, and
        // register 3 of (W, JQ) as it stands goes along — the chain wave finishes both itself (a_chain_wave)
#pragma unroll
        for (int r = 0; r < 4; ++r) L.Dh[W - 1][r][lane] = R[W][r];
        L.Dh3[W - 1][lane] = R[JQ][3];
        lds_post(&L.d_ready[W], 1, lane);
      }
    }
    // column block JQ of row W is final
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      L.Lt[oidx(W, JQ)][r][lane] = R[JQ][r];
      if (ia == 0) st_wt(&Ld[(size_t)(16 * JQ + lk + 4 * r) * NB + 16 * W + lr], R[JQ][r]);
    }
    lds_post(&L.col_done[W], JQ + 1, lane);
    // rank-16 updates of the other sub-tiles right of this phase
#pragma unroll
    for (int J = JQ + 1; J < W; ++J) {
      lds_wait(&L.col_done[J], JQ + 1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) R[J] = mfma_f64(-L.Lt[oidx(J, JQ)][ks][lane], R[JQ][ks], R[J]);
    }
    if (W == WT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = (lk + 4 * r >= lr) ? Wt[r] : 0.0;      // Wt lane (lr, lk) reg r = (L_JQ,JQ^-1)[lk + 4r][lr]
        L.Wi[JQ][lr * 16 + lk + 4 * r] = v;
        if (ia == 0) st_wt(&Winv[(size_t)JQ * 256 + lr * 16 + lk + 4 * r], v);
      }
      lds_post(&L.w_done, JQ + 1, lane);
    }
  }
  static_assert(W >= 1 && W <= 3, "worker index");
  if (W == WT) {
    // identity pseudo-tiles of the phases after the own ones
#pragma unroll
    for (int JQ = WT; JQ < 4; ++JQ) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Wt[r] = (lr == lk + 4 * r) ? 1.0 : 0.0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int n = 4 * JQ + s;
        lds_wait(&L.it_done, n + 1);
        const double mop = L.mop[n][lane], xm = L.xm[n][lane];
        const double xw = mfma_f64(mop, Wt[s], zero)[0];
        Wt[s] = xw;
        if (s < 3) Wt = mfma_f64(-xm, xw, Wt);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = (lk + 4 * r >= lr) ? Wt[r] : 0.0;
        L.Wi[JQ][lr * 16 + lk + 4 * r] = v;
        if (ia == 0) st_wt(&Winv[(size_t)JQ * 256 + lr * 16 + lk + 4 * r], v);
      }
      lds_post(&L.w_done, JQ + 1, lane);
    }
  }
}

template <int NPAN, bool EARLY>
__device__ __forceinline__ void ll_panel_rows(double* __restrict__ S, int ld, int k, int it, int q, int gate, int lane, ALds& L, v4d (&Tq)[4],
                                           const double (&tb)[16], float* __restrict__ L32t) {
  const int lr = lane & 15, lk = lane >> 4;
  double* tcol = S + (size_t)(k * NB) * ld + (size_t)it * NB + 16 * q + lr;
  float* pcol = L32t ? L32t + lk * NB + 16 * q + lr : nullptr;      // packed f32 copy of the tile (element (row, col) at col * 64 + row)
  v4d xt[4], t[4];
  // order: | T(0) x(0) T(1) t(1) | x(1) T(2) t(2) | x(2) T(3) t(3) | x(3): after the last phase only the four MFMAs with the
  // last inverse are left.  t(b) = A_b - sum_{c<b} X_c L(b,c)^T needs phase b-1, x(b) = t(b) L_bb^-T the inverse of phase b.
  // This wave shares its SIMD with a worker that answers the chain wave with two or three MFMAs per iteration: the
  // throughput work here goes in bursts of four MFMAs with a pause after each, so the matrix pipe is free half the time.
#define PANEL_YIELD() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(1); __builtin_amdgcn_sched_barrier(0); } while (0)
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (b == 0) {
      if (NPAN > 0 && EARLY) {
        if (gate == 1) lds_wait(&L.d_ready[1], 1);   // wave 5 / 6 share their SIMD with worker 1 / 2: not before that one's hand-off
        if (gate == 2) lds_wait(&L.d_ready[2], 1);   // (worker 3 has two phases of slack: wave 7 starts at once)
        // the whole pending update at once, while the factor waves are in their own: the phases that follow then see
        // only the short substitution bursts of this wave on their SIMD
#pragma unroll
        for (int ks = 0; ks < 16 * NPAN; ++ks) {
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) Tq[bb] = mfma_f64(-L.Pk[4 * ks + lk][16 * bb + lr], tb[ks], Tq[bb]);
        }
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) pin(Tq[bb]);
      }
      lds_wait(&L.it_done, 4);
      if (NPAN > 0 && !EARLY) {
#pragma unroll
        for (int ks = 0; ks < 16 * NPAN; ++ks) {
          Tq[0] = mfma_f64(-L.Pk[4 * ks + lk][lr], tb[ks], Tq[0]);
          if ((ks & 3) == 3) PANEL_YIELD();
        }
      }
      t[0] = Tq[0];
    }
    lds_wait(&L.w_done, b + 1);
    v4d x = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) x = mfma_f64(L.Wi[b][(4 * s4 + lk) * 16 + lr], t[b][s4], x);   // (L_bb^-1)[lr][4 s4 + lk]
    xt[b] = x;
#pragma unroll
    for (int r = 0; r < 4; ++r) st_wt(&tcol[(size_t)(16 * b + lk + 4 * r) * ld], x[r]);
    if (pcol) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pcol[(16 * b + 4 * r) * NB] = (float)x[r];
    }
    if (b < 3) {
      PANEL_YIELD();
      if (NPAN > 0 && !EARLY) {
#pragma unroll
        for (int ks = 0; ks < 16 * NPAN; ++ks) {
          Tq[b + 1] = mfma_f64(-L.Pk[4 * ks + lk][16 * (b + 1) + lr], tb[ks], Tq[b + 1]);
          if ((ks & 3) == 3) PANEL_YIELD();
        }
      }
      lds_wait(&L.col_done[b + 1], b + 1);
      v4d tn = Tq[b + 1];
#pragma unroll
      for (int c = 0; c <= b; ++c) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) tn = mfma_f64(-L.Lt[oidx(b + 1, c)][s4][lane], xt[c][s4], tn);   // -L[16(b+1) + lr][16c + lk + 4 s4]
        if (b < 2 || c < b) PANEL_YIELD();        // (not before the closing x(3))
      }
      t[b + 1] = tn;
      pin(t[b + 1]);
    }
  }
#undef PANEL_YIELD
}

// diagnostic (tools/ll_trace.py): host-pinned progress words per task, written at system scope so that the host can read them while
// the launch is still running: [0] stage reached by wave 0 (1 started, 2 older panels summed, 3 staged, 4 factored, 5 published),
// [1] kind, [2] k, [3] first tile row, [4] / [5] bit w: wave w is behind the first / second barrier, [6] wait calls that gave up
__device__ __forceinline__ void ll_mark(int* tr, int word, int v) {
  if (tr && (threadIdx.x & 63) == 0) __hip_atomic_store(tr + word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void ll_time(int* tr, int word) {      // [8 + ..]: wall clock (100 MHz, low 32 bits) at a stage / event
  if (tr && (threadIdx.x & 63) == 0) __hip_atomic_store(tr + word, (int)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void ll_stage(int* tr, int v) { ll_mark(tr, 0, v); ll_time(tr, 8 + v); }
__device__ __forceinline__ void ll_mark_or(int* tr, int word, int v) {
  if (tr && (threadIdx.x & 63) == 0) __hip_atomic_fetch_or(tr + word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// flags f0[c * stride] (and f1[c * stride] unless null), c0 <= c < c1, all raised?  One lane per column, the whole wave spins.
__device__ __forceinline__ bool ll_wait_cols(const int* f0, const int* f1, int stride, int c0, int c1, int* ctl, int* status) {
  const int lane = threadIdx.x & 63;
  bool ok = true;
  for (int base = c0; base < c1 && ok; base += 64) {
    const int c = base + lane;
    const bool in = c < c1;
    int spins = 0;
    for (;;) {
      int a = 1, b = 1;
      if (in) {
        a = __hip_atomic_load(f0 + (size_t)c * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (f1) b = __hip_atomic_load(f1 + (size_t)c * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (__all(a != 0 && b != 0)) break;
      ++spins;
      if (spins > LL_SPIN_MAX || ((spins & 63) == 0 && __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        if (lane == 0) {
          atomicOr(&status[1], 2);
          __hip_atomic_store(ctl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  // No acquire-side invalidation (buffer_inv sc1) here: every tile has ONE writer and no workgroup reads it before its flag, so no cache
  // of the reading CU / XCD can hold a line of it from before the write (tools/tile_hop_bench.hip: stale values only with a read ahead
  // of the flag) — and the invalidation empties the XCD's L2 under every other workgroup's operands.
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  return ok;
}
// sixteen rows (16 q ..) of tile row `it` of column k in the T-layout of panel_rows: Tq[b][r] = A[16 q + lr][16 b + lk + 4 r]
__device__ __forceinline__ void ll_rows_load(const double* __restrict__ S, int ld, int k, int it, int q, int lr, int lk, v4d (&Tq)[4]) {
  const double* tcol = S + (size_t)(k * NB) * ld + (size_t)it * NB + 16 * q + lr;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) Tq[b][r] = tcol[(size_t)(16 * b + lk + 4 * r) * ld];
}
// Tq -= L(it, c)[rows 16 q ..] L(k, c)^T for one finished block column c: operands straight from memory (A: the own rows of tile
// (it, c), B: tile (k, c), sixteen-lane runs down its columns)
__device__ __forceinline__ void ll_rows_term(const double* __restrict__ S, int ld, int k, int it, int q, int c, int lr, int lk, v4d (&Tq)[4]) {
  const double* pa = S + (size_t)(c * NB + lk) * ld + (size_t)it * NB + 16 * q + lr;
  const double* pb = S + (size_t)(c * NB + lk) * ld + (size_t)k * NB + lr;
#pragma unroll 8
  for (int ks = 0; ks < 16; ++ks) {
    const size_t o = (size_t)(4 * ks) * ld;
    const double a = pa[o];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) Tq[bb] = mfma_f64(-pb[o + 16 * bb], a, Tq[bb]);
  }
}
// sub-tile row w of the diagonal block: R[J] -= L(k, c)[rows 16 w ..] L(k, c)[rows 16 J ..]^T, J <= w
__device__ __forceinline__ void ll_diag_term(const double* __restrict__ S, int ld, int k, int w, int c, int lr, int lk, v4d (&R)[4]) {
  const double* pc = S + (size_t)(c * NB + lk) * ld + (size_t)k * NB + lr;
#pragma unroll 8
  for (int ks = 0; ks < 16; ++ks) {
    const size_t o = (size_t)(4 * ks) * ld;
    const double a = pc[o + 16 * w];
#pragma unroll
    for (int J = 0; J < 4; ++J)
      if (J <= w) R[J] = mfma_f64(-pc[o + 16 * J], a, R[J]);
  }
}

// The older pending panels of a chain task (c < k-1), one at a time through LDS by the WHOLE workgroup: every wave loads eight columns of
// tile (k, c) in full-line runs and the sixteen values of its own rows of tile (it, c), the tile is staged in L.Pk (what the pending
// panel k-1 uses afterwards), the factor waves update their sub-tile row of D with both operands from LDS, the row waves their rows of
// tile (it, k).  (Operands straight from memory — one 128-byte run per MFMA operand — cost ~50 us per term under the flood of
// acquire-side invalidations of a persistent launch: tools/ll_trace.py.)  All waves of the workgroup call this together (barriers).
__device__ __forceinline__ void ll_chain_presum(const double* __restrict__ S, int ld, int k, int it, int c_lo_d, int c_lo_row, int c_end, ALds& L,
                                                int wave, int lane, bool factor, int w, bool rows, int q, v4d (&R)[4], v4d (&Tq)[4]) {
  const int lr = lane & 15, lk = lane >> 4;
  for (int c = c_lo_d; c < c_end; ++c) {
    constexpr int NC = 8;
    const double* pq = S + (size_t)(c * NB + NC * wave) * ld + (size_t)k * NB + lane;
    double stage[NC];
#pragma unroll
    for (int e = 0; e < NC; ++e) stage[e] = pq[(size_t)e * ld];
    const bool row_on = rows && c >= c_lo_row;
    double ta[16];
    if (row_on) {
      const double* pi = S + (size_t)(c * NB) * ld + (size_t)it * NB + 16 * q + lr;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) ta[ks] = pi[(size_t)(4 * ks + lk) * ld];
    }
    __syncthreads();                      // (everybody is through with the previous panel in L.Pk)
#pragma unroll
    for (int e = 0; e < NC; ++e) L.Pk[NC * wave + e][lane] = stage[e];
    __syncthreads();
    if (factor) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const double w0 = L.Pk[4 * ks + lk][16 * w + lr];
#pragma unroll
        for (int J = 0; J < 4; ++J)
          if (J <= w) R[J] = mfma_f64(-L.Pk[4 * ks + lk][16 * J + lr], w0, R[J]);
      }
    }
    if (row_on) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) Tq[bb] = mfma_f64(-L.Pk[4 * ks + lk][16 * bb + lr], ta[ks], Tq[bb]);
    }
  }
  __syncthreads();                        // (L.Pk is free for the pending panel k-1)
}

// ---- tile task: tile rows it0 (waves 0..3) and it1 (waves 4..7; -1: none) of column k ----
__device__ __forceinline__ void ll_tile_task(const LLSys& Y, const LLTask& tk, ALds& L, int* ctl, int* tr) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int slot = wave >> 2, q = wave & 3;
  const int it = slot ? tk.it1 : tk.it0;
  const int clo = slot ? tk.clo1 : tk.clo0;
  const int k = tk.k, ld = Y.ld, fr = Y.frows;
  double* S = Y.S;
  const int* fk = Y.flags + k;
  v4d Tq[4];
  if (it >= 0) ll_rows_load(S, ld, k, it, q, lr, lk, Tq);
  // sums over the finished columns, the tile (k, c) both rows share staged through LDS by the whole workgroup (ll_chain_presum); wave 0
  // polls the flags for everybody: first all columns but the last (through long ago), then the last one
  const int cmin = tk.it1 >= 0 ? min(tk.clo0, tk.clo1) : tk.clo0;
  for (int half = 0; half < 2; ++half) {
    const int c0 = half ? max(cmin, k - 1) : cmin, c1 = half ? k : k - 1;
    if (c0 >= c1) continue;
    if (wave == 0) {
      ll_wait_cols(fk, nullptr, fr, c0, c1, ctl, Y.status);
      if (max(tk.clo0, c0) < c1) ll_wait_cols(Y.flags + tk.it0, nullptr, fr, max(tk.clo0, c0), c1, ctl, Y.status);
      if (tk.it1 >= 0 && max(tk.clo1, c0) < c1) ll_wait_cols(Y.flags + tk.it1, nullptr, fr, max(tk.clo1, c0), c1, ctl, Y.status);
    }
    __syncthreads();
    v4d Rdummy[4];
    ll_chain_presum(S, ld, k, it, c0, clo, c1, L, wave, lane, false, 0, it >= 0, q, Rdummy, Tq);
  }
  if (wave == 0) {
    ll_time(tr, 10);                         // (tile task: the sums over the finished columns are through)
    ll_wait_cols(fk, nullptr, fr, k, k + 1, ctl, Y.status);
    ll_time(tr, 15);
  }
  __syncthreads();
  if (it >= 0) {
    // L_kk: the off-diagonal 16x16 sub-tiles (Ld) and the 16x16 inverses (Winv) the chain task of this column published
    const double* Ldk = Y.Ld + (size_t)k * NB * NB;
    const double* Wk = Y.Winv + (size_t)k * 1024;
    double wi[4][4], lt[6][4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) wi[b][s4] = Wk[(size_t)b * 256 + (4 * s4 + lk) * 16 + lr];                 // (L_bb^-1)[lr][4 s4 + lk]
#pragma unroll
    for (int I = 1; I < 4; ++I)
#pragma unroll
      for (int J = 0; J < I; ++J)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) lt[oidx(I, J)][s4] = Ldk[(size_t)(16 * J + lk + 4 * s4) * NB + 16 * I + lr];   // L[16 I + lr][16 J + lk + 4 s4]
    double* tcol = S + (size_t)(k * NB) * ld + (size_t)it * NB + 16 * q + lr;
    v4d xt[4];
    v4d t = Tq[0];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      v4d x = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) x = mfma_f64(wi[b][s4], t[s4], x);
      xt[b] = x;
#pragma unroll
      for (int r = 0; r < 4; ++r) st_wt(&tcol[(size_t)(16 * b + lk + 4 * r) * ld], x[r]);
      if (b < 3) {
        v4d tn = Tq[b + 1];
#pragma unroll
        for (int c = 0; c <= b; ++c)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) tn = mfma_f64(-lt[oidx(b + 1, c)][s4], xt[c][s4], tn);
        t = tn;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  }
  ll_mark_or(tr, 4, 1 << wave);
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(Y.flags + (size_t)k * fr + tk.it0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tk.it1 >= 0) __hip_atomic_store(Y.flags + (size_t)k * fr + tk.it1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (wave == 0) ll_stage(tr, 5);
}

// ---- chain task: the diagonal block of column k and tile row it0 (step_type_a_impl behind a flag instead of a kernel boundary) ----
template <int NPAN>
__device__ __forceinline__ void ll_chain_task(const LLSys& Y, const LLTask& tk, ALds& L, int& row_ok, int* ctl, int* tr) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..7
  const int lr = lane & 15, lk = lane >> 4;
  const int k = tk.k, it = tk.it0, ld = Y.ld, fr = Y.frows;
  double* S = Y.S;
  double* Ld = Y.Ld + (size_t)k * NB * NB;
  double* Winv = Y.Winv + (size_t)k * 1024;
  const int* fk = Y.flags + k;
  const int* fi = Y.flags + it;
  const bool row_prev = tk.clo0 <= k - 1;        // tile (it, k-1) exists (else its contribution is zero)
  const int ia = tk.kind == 0 ? 0 : 1;           // 0: the column's LEAD chain task publishes L_kk's pieces; followers (kind 2) factor the block for themselves only
  if (tid == 0) L.it_done = 0;
  if (tid < 4) { L.col_done[tid] = 0; L.d_ready[tid] = 0; }
  if (tid == 4) L.w_done = 0;
  if (tid == 5) row_ok = 0;
  const int kold = k - 1;                        // panels c < kold are summed left-looking before the pending panel k-1 arrives
  if (wave < 4) {
    // ---------------- factor waves: own sub-tile row of D ----------------
    const double* dcol = S + (size_t)(k * NB) * ld + (size_t)k * NB + 16 * wave + lr;
    v4d R[4];
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
      for (int r = 0; r < 4; ++r) R[J][r] = (J <= wave) ? dcol[(size_t)(16 * J + lk + 4 * r) * ld] : 0.0;
    v4d Tq[4];
    double tb[16];
    if (wave == 1) ll_rows_load(S, ld, k, it, 0, lr, lk, Tq);
    // ONE wave of the workgroup polls (wave 0 the panels' flags, wave 4 — otherwise idle — the own rows' last flag), the others wait at
    // a barrier: eight polling waves per workgroup and ~170 workgroups per block column load the memory-side path that the flags and the
    // tiles themselves travel on
    if (tk.clo_d < kold) {
      if (wave == 0) {
        ll_wait_cols(fk, nullptr, fr, tk.clo_d, kold, ctl, Y.status);
        if (tk.clo0 < kold) ll_wait_cols(fi, nullptr, fr, tk.clo0, kold, ctl, Y.status);      // (tile (it, c) exists from column clo0 on only)
      }
      __syncthreads();
      ll_chain_presum(S, ld, k, it, tk.clo_d, tk.clo0, kold, L, wave, lane, true, wave, wave == 1, 0, R, Tq);
    }
    if (wave == 0) ll_stage(tr, 2);
    if (NPAN > 0) {
      // the pending panel: tile (k, k-1), the previous chain task's own tile — the one flag the chain waits for
      if (wave == 0) {
        ll_wait_cols(fk, nullptr, fr, k - 1, k, ctl, Y.status);
        ll_time(tr, 14);
      }
      __syncthreads();
      constexpr int NC = 8;
      const double* pq = S + (size_t)((k - 1) * NB + NC * wave) * ld + (size_t)k * NB + lane;
      double stage[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) stage[c] = pq[(size_t)c * ld];
#pragma unroll
      for (int c = 0; c < NC; ++c) L.Pk[NC * wave + c][lane] = stage[c];
    }
    ll_mark_or(tr, 4, 1 << wave);
    __syncthreads();
    if (wave == 0) ll_stage(tr, 3);
    if (wave == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);   // ahead of the panel wave sharing the SIMD
    if (wave == 0) {
      a_chain_wave<NPAN>(ia, Y.status, L, lane, R[0]);
    } else if (wave == 1) {
      v4d R1[2] = {R[0], R[1]};
      ll_worker_wave<1, NPAN>(ia, Ld, Winv, L, lane, R1);
      if (NPAN > 0) {
        // the own rows of tile (it, k-1) — a TILE task of column k-1, through some microseconds after that column's chain task: waited
        // for here, not in front of the chain
        if (row_prev) lds_wait(&row_ok, 1);
        const double* pi = S + (size_t)((k - 1) * NB) * ld + (size_t)it * NB + lr;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) tb[ks] = row_prev ? pi[(size_t)(4 * ks + lk) * ld] : 0.0;
      }
      ll_panel_rows<NPAN, false>(S, ld, k, it, 0, 0, lane, L, Tq, tb, nullptr);
    } else if (wave == 2) {
      v4d R2[3] = {R[0], R[1], R[2]};
      ll_worker_wave<2, NPAN>(ia, Ld, Winv, L, lane, R2);
    } else {
      ll_worker_wave<3, NPAN>(ia, Ld, Winv, L, lane, R);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    ll_mark_or(tr, 5, 1 << wave);
    __syncthreads();
    if (wave == 0) ll_stage(tr, 4);
  } else {
    // ---------------- panel waves 5..7: rows 16..63 of tile (it, k) (rows 0..15: worker 1 after its hand-off; wave 4 idles so that
    // the chain wave has its SIMD to itself) ----------------
    const int q = wave - 4;
    const bool rows = q > 0;
    v4d Tq[4];
    double tb[16];
    if (rows) ll_rows_load(S, ld, k, it, q, lr, lk, Tq);
    if (tk.clo_d < kold) {
      v4d Rdummy[4];
      __syncthreads();                      // (wave 0 has seen the older panels' flags)
      ll_chain_presum(S, ld, k, it, tk.clo_d, tk.clo0, kold, L, wave, lane, false, 0, rows, q, Rdummy, Tq);
    }
    if (NPAN > 0) {
      __syncthreads();                      // (wave 0 has seen the flag of tile (k, k-1))
      constexpr int NC = 8;
      const double* pq = S + (size_t)((k - 1) * NB + NC * wave) * ld + (size_t)k * NB + lane;
      double stage[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) stage[c] = pq[(size_t)c * ld];
#pragma unroll
      for (int c = 0; c < NC; ++c) L.Pk[NC * wave + c][lane] = stage[c];
    }
    ll_mark_or(tr, 4, 1 << wave);
    __syncthreads();
    if (NPAN > 0 && row_prev && wave == 4) {      // the poller of the own rows' last flag
      ll_wait_cols(fi, nullptr, fr, k - 1, k, ctl, Y.status);
      lds_post(&row_ok, 1, lane);
    }
    if (NPAN > 0 && rows) {      // (the own rows of tile (it, k-1): see worker 1)
      if (row_prev) lds_wait(&row_ok, 1);
      const double* pi = S + (size_t)((k - 1) * NB) * ld + (size_t)it * NB + 16 * q + lr;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) tb[ks] = row_prev ? pi[(size_t)(4 * ks + lk) * ld] : 0.0;
    }
    if (rows) ll_panel_rows<NPAN, true>(S, ld, k, it, q, wave - 4, lane, L, Tq, tb, nullptr);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    ll_mark_or(tr, 5, 1 << wave);
    __syncthreads();
  }
  if (tid == 0) {
    if (ia == 0) __hip_atomic_store(Y.flags + (size_t)k * fr + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(Y.flags + (size_t)k * fr + it, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (wave == 0) ll_stage(tr, 5);
}

// ctl[0]: ticket counter, ctl[1]: abort word (both cleared with the flags before the launch); trace: null, or 16 host-pinned ints per task
__global__ __launch_bounds__(512) void k_chol_ll(const LLSys* __restrict__ sysv, const LLTask* __restrict__ tasks, int n_tasks, int* ctl, int* trace) {
  __shared__ ALds L;
  __shared__ int s_t, s_row_ok;
  if (threadIdx.x == 0) s_t = atomicAdd(&ctl[0], 1);
  __syncthreads();
  const int t = __builtin_amdgcn_readfirstlane(s_t);
  if (t >= n_tasks) return;
  const LLTask tk = tasks[t];
  const LLSys Y = sysv[tk.sys];
  int* tr = trace ? trace + 16 * (size_t)t : nullptr;
  if (tr && threadIdx.x == 0) { ll_mark(tr, 1, tk.kind); ll_mark(tr, 2, tk.k); ll_mark(tr, 3, tk.it0); ll_stage(tr, 1); }
  if (tk.kind != 1) {
    if (tk.clo_d <= tk.k - 1) ll_chain_task<1>(Y, tk, L, s_row_ok, ctl, tr);
    else ll_chain_task<0>(Y, tk, L, s_row_ok, ctl, tr);
  } else {
    ll_tile_task(Y, tk, L, ctl, tr);
  }
}

// (diagnostic, not part of include/slide_gpu.h) flag waits of the pair kernel that gave up since the library was loaded: 0 unless a
// launch was mis-planned
extern "C" int slide_debug_pair_timeouts() {
  int v = -1;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_pair_timeouts), sizeof(int)) != hipSuccess) return -1;
  return v;
}
#ifdef SLIDE_STAMPS
extern "C" void slide_debug_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)); }
extern "C" void slide_debug_chain_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_stamps), sizeof(g_chain_stamps)); }
#endif

// A quiet-NaN payload no solution value can equal bit for bit: dp[] is filled with it before the backward substitution,
// whose workgroups poll the entries of the blocks they depend on ("flag in data": one round trip per link of the chain).
constexpr unsigned long long BWD_SENT = CHAIN_SENTINEL;      // (kernels.hpp: k_pcg_update pre-fills the forward chain's output with it)

__global__ void k_chol_extract_y(const double* __restrict__ S, int ld, int T, int Tr, double* __restrict__ yv, double* __restrict__ dp, int* status,
                                 double* __restrict__ prev) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) status[4] = 0;            // ticket counter of the chained backward substitution that follows
  if (c < T * NB) {
    yv[c] = S[(size_t)c * ld + (size_t)Tr * NB];      // (Tr: tile row of the right-hand side, T + border rows)
    if (prev) prev[c] = dp[c];                        // (bounded back-substitution: the last solve's solution, see bwd_chain_body<.., true>)
    dp[c] = __longlong_as_double((long long)BWD_SENT);
  }
}

// the same for up to 8 systems in one launch (blockIdx.y = system): eight 5 us launches in a row at the end of the bands' steps otherwise
struct ExtractArgs { int n; const double* S[CHOL_STEP_BATCH_MAX]; int ld[CHOL_STEP_BATCH_MAX]; int T[CHOL_STEP_BATCH_MAX]; int Tr[CHOL_STEP_BATCH_MAX]; double* yv[CHOL_STEP_BATCH_MAX]; double* dp[CHOL_STEP_BATCH_MAX]; int* status[CHOL_STEP_BATCH_MAX]; };
__global__ void k_chol_extract_y_b(ExtractArgs A) {
  const int r = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0) A.status[r][4] = 0;
  if (c < A.T[r] * NB) {
    A.yv[r][c] = A.S[r][(size_t)c * A.ld[r] + (size_t)A.Tr[r] * NB];
    A.dp[r][c] = __longlong_as_double((long long)BWD_SENT);
  }
}

// Backward substitution L^T x = y as ONE launch: workgroup b owns block c = T-1-b.  It accumulates
//   y_c - sum_{j > c} L(j, c)^T x_j
// in descending j as the x_j appear in dp, then x_c = (L_cc^-1)^T (.) and publishes it.  Block ids are TICKETS drawn from an
// atomic counter (status[4], cleared by k_chol_extract_y) when a workgroup starts, not blockIdx: a block only ever waits for blocks
// with lower tickets, i.e. for workgroups that have started already (resident or finished) — the order is enforced, not assumed
// from the dispatcher.  The tiles L(j, c)
// do not depend on x and are prefetched three ahead; the explicit 64x64 inverse of L_cc is assembled from the 16x16 inverses
// and sub-tiles of the factorisation while the workgroup would otherwise wait.  Polling loads and publishing stores are
// relaxed device-scope atomics (they bypass the non-coherent cache levels between XCDs); the data is its own flag.
// F32 = false: tiles from S (f64) — the solve of the factorisation's own right-hand side.  F32 = true: tiles from L32, the packed f32
// copy the type-A workgroups of the factorisation write next to every panel tile (step_type_a) — the preconditioner of the joint
// solve (pcg_kernels.hip) streams half the bytes, in 16 KB runs; any SPD M~ = L~ L~^T is a valid preconditioner, the operator S0
// and all vectors stay f64.  L32: tile (j, c), j > c, at 4096 * (c (T-1) - c (c-1) / 2 + j - c - 1), element (row, col) of it at
// col * 64 + row.
// Workgroups of the chained substitutions: four waves own the tiles, a fifth does nothing but poll for the next x_j.  Vector-memory
// loads return in order, so a poll issued by a wave that has tile prefetches in flight is not seen before those have come back from
// HBM (measured with tools/chain_stamps.py: 0.9 us mean per hop against 0.5 us for the bare publish -> poll round trip,
// tools/hop_bench.hip); the polling wave has no other loads outstanding.
constexpr int CHAIN_THREADS = 320;
struct alignas(16) ChainLds {     // one per workgroup, shared by every instantiation of the chain bodies (16-byte aligned: ds_read_b128)
  alignas(16) double xs[2][NB];
  alignas(16) double ys[NB];
  alignas(16) double red[4][NB];
  alignas(16) double Lo[6][256];  // off-diagonal 16x16 blocks (b > a) of L_cc: Lo[b (b-1)/2 + a][col * 16 + row]
  alignas(16) double Ws[4][256];  // 16x16 inverses: Ws[b][col * 16 + row]
  alignas(16) double tmp[3][256];
  alignas(16) double Ms[NB][NB + 1];   // M = L_cc^-1 (lower triangle), Ms[row][col]
};
// BOUNDED variant (WFIRE; iSAM2's wildfire threshold, ISAM2GaussNewtonParams::wildfireThreshold = 1e-3 in GTSAM 4.0.3, which the
// reference's ISAM2 runs with: graph.cpp:15-18, 260-272).  iSAM2 does not re-solve a clique whose parents' delta changed by less than
// the threshold; on the block chain of the banded reduced system: block c keeps the solution of the LAST solve (wf_prev) when every
// block it depends on (c+1 .. prof[c]) changed by less than wf_thr in the infinity norm — and, the profile being monotone, so does
// every block below it: the first such block (walking down) raises the stop word (status[7]) and all workgroups below leave with their
// previous values at once instead of passing the chain on hop by hop.  wf_Tprev: blocks >= it have no previous value (new key frames);
// wf_lim (<= wf_Tprev): the first DIRTY block column — only blocks below it may keep their value; the re-factored blocks above it are
// always solved, and their change against the last solve counts like any other dependency's (round 5: they used to count as changed
// whatever they did, which kept the first blocks below the dirty column from ever being quiet; ISAM2's rule compares the re-eliminated
// cliques' new delta with the old one too).
// status[3] counts the blocks that were kept.  Off (the default): the chain always runs in full — the linear system is solved exactly.
template <bool F32, bool WFIRE = false>
__device__ __forceinline__ void bwd_chain_body(ChainLds& W, const double* __restrict__ S, int ld, int T, const double* __restrict__ Ld,
                                               const double* __restrict__ Winv, const double* __restrict__ yv, double* dp, int* status, int bidx,
                                               const float* __restrict__ L32, const int* __restrict__ prof,
                                               const double* __restrict__ wf_prev = nullptr, double wf_thr = 0.0, int wf_Tprev = 0,
                                               int* wf_lds = nullptr, int wf_lim = 0) {
  int dummy_state = 0;
  int& wf_state = WFIRE ? *wf_lds : dummy_state;      // (LDS word of the bounded kernel) 0: compute, 1: a block above raised the stop word, 2: all inputs quiet (this block raises it)
  auto& Ms = W.Ms; auto& Lo = W.Lo; auto& Ws = W.Ws; auto& tmp = W.tmp; auto& xs = W.xs; auto& ys = W.ys;
  const int tid = threadIdx.x;
  const bool worker = tid < 256;                       // waves 0..3: the tiles; wave 4 only polls (see CHAIN_THREADS)
  CSTAMP(0);
  const int c = T - 1 - bidx;
  if (WFIRE) {
    if (tid == 0) wf_state = 0;
    __syncthreads();
  }
  const int col = tid >> 2, part = tid & 3;            // tile work: column col, rows 16 part .. 16 part + 15
  const int jtop = prof ? prof[c] : T - 1;             // last tile row of column c inside the factor's profile (dense: T - 1)
  const int nj = jtop - c;                             // tiles (j, c), j = jtop-q, q = 0 .. nj-1
  constexpr int RB = F32 ? 6 : 3;
  typedef typename std::conditional<F32, float, double>::type tile_t;
  tile_t tr[RB][16];
  const double* tcol = S + (size_t)(c * NB + col) * ld + 16 * part;
  // packed tiles of column c: (c+1, c) first; tile (jtop-q, c) is number nj-1-q of the run
  const float* pcol = F32 ? L32 + ((size_t)c * (T - 1) - (size_t)c * (c - 1) / 2) * (NB * NB) + col * NB + 16 * part : nullptr;
  auto tile_load = [&](tile_t (&dst)[16], int q) {
    if (F32) {
      const float4* tp = reinterpret_cast<const float4*>(pcol + (size_t)(nj - 1 - q) * (NB * NB));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4 v = tp[r];
        dst[4 * r] = v.x; dst[4 * r + 1] = v.y; dst[4 * r + 2] = v.z; dst[4 * r + 3] = v.w;
      }
    } else {
      const double* tp = tcol + (size_t)(jtop - q) * NB;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r] = tp[r];
    }
  };
  double y0 = 0.0;
  if (worker) {
#pragma unroll
    for (int q = 0; q < RB; ++q)
      if (q < nj) tile_load(tr[q], q);
    y0 = yv[c * NB + col];
  }
  if (worker) {
    const double* Ldk = Ld + (size_t)c * NB * NB;
    const double* Wk = Winv + (size_t)c * 1024;
#pragma unroll
    for (int b = 1; b < 4; ++b)
#pragma unroll
      for (int a = 0; a < b; ++a) Lo[b * (b - 1) / 2 + a][tid] = Ldk[(size_t)(16 * a + (tid >> 4)) * NB + 16 * b + (tid & 15)];
#pragma unroll
    for (int b = 0; b < 4; ++b) Ws[b][tid] = Wk[(size_t)b * 256 + tid];
  }
  __syncthreads();
  {
    // M_bb = W_b ; M_ba = -W_b sum_{m = a}^{b-1} L_bm M_ma  (b > a), by distance from the diagonal; thread = element (r, cc)
    const int r = tid & 15, cc = (tid >> 4) & 15;
    if (worker) {
#pragma unroll
      for (int b = 0; b < 4; ++b) Ms[16 * b + r][16 * b + cc] = Ws[b][cc * 16 + r];
    }
    __syncthreads();
#pragma unroll
    for (int d = 1; d < 4; ++d) {
#pragma unroll
      for (int a = 0; a + d < 4; ++a) {
        const int b = a + d;
        double t = 0.0;
#pragma unroll
        for (int m = a; m < b; ++m)
#pragma unroll
          for (int n = 0; n < 16; ++n) t += Lo[b * (b - 1) / 2 + m][n * 16 + r] * Ms[16 * m + n][16 * a + cc];
        if (worker) tmp[a][cc * 16 + r] = t;
      }
      __syncthreads();
#pragma unroll
      for (int a = 0; a + d < 4; ++a) {
        const int b = a + d;
        double v = 0.0;
#pragma unroll
        for (int n = 0; n < 16; ++n) v += Ws[b][n * 16 + r] * tmp[a][cc * 16 + n];
        if (worker) Ms[16 * b + r][16 * a + cc] = -v;
      }
      __syncthreads();
    }
  }
  double mreg[16];                       // own column of M^T: rows 16 part .. of column col (zero above the diagonal)
#pragma unroll
  for (int r = 0; r < 16; ++r) mreg[r] = (worker && 16 * part + r >= col) ? Ms[16 * part + r][col] : 0.0;
  double acc = 0.0;
  bool wf_quiet = WFIRE && nj > 0 && c < wf_lim;      // (only a block below the first dirty column may keep its value: wf_lim <= wf_Tprev)      // (polling wave: every block this one depends on changed by less than the threshold, so far)
  bool wf_leave = false;
  CSTAMP(1);
  for (int q0 = 0; q0 < nj && !wf_leave; q0 += RB) {
#pragma unroll
    for (int qq = 0; qq < RB; ++qq) {
      const int q = q0 + qq;
      if (q < nj && !wf_leave) {
        const int j = jtop - q;
        if (q == nj - 1) CSTAMPP(2);
        if (!worker) {                          // the polling wave has no other vector-memory traffic: its loads are not queued behind tile prefetches
          const int lane = tid - 256;
          double v;
          int spins = 0;
          bool stopped = false;
          for (;;) {
            v = __hip_atomic_load(dp + (size_t)j * NB + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned long long)__double_as_longlong(v) != BWD_SENT) break;
            if (WFIRE && (spins & 3) == 3 && __hip_atomic_load(status + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > c) { stopped = true; break; }
            if (++spins > (1 << 21)) {          // exit condition every wave reaches: give up (seconds), flag the solve as failed
              v = __builtin_nan("");
              atomicOr(&status[1], 2);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
          if (WFIRE) {
            stopped = __any(stopped);
            if (stopped) {
              if (lane == 0) wf_state = 1;
            } else {
              double dv = fabs(v - ((j < wf_Tprev) ? wf_prev[(size_t)j * NB + lane] : 1e300));
#pragma unroll
              for (int m = 32; m >= 1; m >>= 1) dv = fmax(dv, __shfl_xor(dv, m));
              wf_quiet = wf_quiet && (dv < wf_thr);
              if (q == nj - 1 && wf_quiet && lane == 0) wf_state = 2;
            }
          }
          xs[q & 1][lane] = v;
          if (q == nj - 1) CSTAMPP(3);
          if (q == nj - 2) CSTAMPP(7);
        }
        __syncthreads();
        if (WFIRE && wf_state == 1) { wf_leave = true; continue; }
        if (worker) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc += (double)tr[qq][r] * xs[q & 1][16 * part + r];
          if (q + RB < nj) tile_load(tr[qq], q + RB);
        }
      }
    }
  }
  CSTAMP(4);
  if (WFIRE && wf_state != 0) {
    // this block keeps the last solve's values: a block above raised the stop word (1), or all of this block's inputs are quiet (2) —
    // then it raises the stop word for everything below.  (A block without a previous value never gets here: wf_quiet needs c < Tprev,
    // and the stop word is only raised at a block with one; the blocks below it are older still.)
    if (tid < NB) __hip_atomic_store(dp + (size_t)c * NB + tid, wf_prev[(size_t)c * NB + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
      if (wf_state == 2) atomicMax(status + 7, c);
      atomicAdd(status + 3, 1);
    }
    return;
  }
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  if (worker && part == 0) ys[col] = y0 - acc;
  __syncthreads();
  CSTAMP(5);
  double x = 0.0;
#pragma unroll
  for (int r = 0; r < 16; ++r) x += mreg[r] * ys[16 * part + r];
  x += __shfl_xor(x, 1);
  x += __shfl_xor(x, 2);
  if (worker && part == 0) __hip_atomic_store(dp + (size_t)c * NB + col, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  CSTAMP(6);
}
__device__ __forceinline__ int bwd_ticket(int* ctr) {
  __shared__ int s_t;
  if (threadIdx.x == 0) s_t = atomicAdd(ctr, 1);
  __syncthreads();
  const int t = s_t;
  __syncthreads();
  return t;
}
__global__ __launch_bounds__(CHAIN_THREADS) void k_chol_bwd_chain(const double* __restrict__ S, int ld, int T, const double* __restrict__ Ld,
                                                        const double* __restrict__ Winv, const double* __restrict__ yv, double* dp, int* status,
                                                        const int* __restrict__ prof) {
  __shared__ ChainLds W;
  const int t = bwd_ticket(&status[4]);
  if (t >= T) return;
  bwd_chain_body<false>(W, S, ld, T, Ld, Winv, yv, dp, status, t, nullptr, prof);
}
// the bounded variant (streaming updates with a wildfire threshold; status[7] = stop word, status[3] = blocks kept, both 0 before)
__global__ __launch_bounds__(CHAIN_THREADS) void k_chol_bwd_chain_wf(const double* __restrict__ S, int ld, int T, const double* __restrict__ Ld,
                                                           const double* __restrict__ Winv, const double* __restrict__ yv, double* dp, int* status,
                                                           const int* __restrict__ prof, const double* __restrict__ prev, double thr, int Tprev, int lim) {
  __shared__ ChainLds W;
  __shared__ int s_wf;
  const int t = bwd_ticket(&status[4]);
  if (t >= T) return;
  bwd_chain_body<false, true>(W, S, ld, T, Ld, Winv, yv, dp, status, t, nullptr, prof, prev, thr, Tprev, &s_wf, lim);
}
constexpr int BWD_BATCH_MAX = 32;     // systems per batched backward substitution (the segments of eight robots' bands: up to 32)
struct BwdBatchArgs {
  int n;
  const double* S[BWD_BATCH_MAX]; int ld[BWD_BATCH_MAX]; int T[BWD_BATCH_MAX];
  const double* Ld[BWD_BATCH_MAX]; const double* Winv[BWD_BATCH_MAX]; const double* yv[BWD_BATCH_MAX]; double* dp[BWD_BATCH_MAX];
  int* status[BWD_BATCH_MAX];
  const int* prof[BWD_BATCH_MAX];      // device: profile of the factor (see plan_step) or null
  int base[BWD_BATCH_MAX + 1];         // prefix sums of T (grid size = base[n])
  int Tmax;
};
// ticket t -> system t % n, block t / n of it: the chains of all systems advance side by side; a ticket beyond a shorter system's
// end is void and the workgroup draws again (there are exactly as many valid tickets as workgroups)
__global__ __launch_bounds__(CHAIN_THREADS) void k_chol_bwd_chain_batched(BwdBatchArgs A) {
  __shared__ ChainLds W;
  for (;;) {
    const int t = bwd_ticket(&A.status[0][4]);
    if (t >= A.n * A.Tmax) return;
    const int r = t % A.n, b = t / A.n;
    if (b >= A.T[r]) continue;
    bwd_chain_body<false>(W, A.S[r], A.ld[r], A.T[r], A.Ld[r], A.Winv[r], A.yv[r], A.dp[r], A.status[r], b, nullptr, A.prof[r]);
    return;
  }
}


// ---- tables for the chained substitutions of the joint solve ---------------------------------------------------------------------------
// The preconditioner of the joint solve runs both substitutions eight times per factorisation.  Once per factorisation, one
// workgroup per block c writes, in the thread order (i = tid >> 2, part = tid & 3: sixteen consecutive entries of row / column i):
//   TAB_MF  M[i][16 part + k]                      M = L_cc^-1 (explicit, from the 16x16 inverses and sub-tiles of the factorisation)
//   TAB_MB  M[16 part + r][i]
//   TAB_PF  (M T(c, c-1))[i][16 part + k]          the sub-diagonal tile folded into the inverse: the block that arrives LAST in the
//   TAB_PB  (T(c+1, c) M)[16 part + r][i]          forward (x_{c-1}) / backward (x_{c+1}) chain then costs ONE 64x64 product on the
// critical path, x_c = z_c - P x_last, where z_c (everything else, times M) is finished while the chain is still one block away.
// ctab: [4][T][4096] doubles per system.  Without the tables a block builds M in LDS (8 us of prologue) and pays two dependent
// products per hop (tools/chain_stamps.py: 0.72 of 1.4 us per block).
enum { TAB_MF = 0, TAB_MB = 1, TAB_PF = 2, TAB_PB = 3 };
struct ChainTabArgs {
  int n; int T[CHOL_BATCH_MAX]; const double* S[CHOL_BATCH_MAX]; int ld[CHOL_BATCH_MAX];
  const double* Ld[CHOL_BATCH_MAX]; const double* Winv[CHOL_BATCH_MAX]; const int* prof[CHOL_BATCH_MAX]; double* ctab[CHOL_BATCH_MAX];
};
__global__ __launch_bounds__(256) void k_chain_tables(ChainTabArgs A) {
  __shared__ ChainLds W;
  __shared__ double Ts[NB][NB];       // the sub-diagonal tile as it lies in memory: Ts[column][row]
  auto& Ms = W.Ms; auto& Lo = W.Lo; auto& Ws = W.Ws; auto& tmp = W.tmp;
  const int sys = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
  const int T = A.T[sys];
  if (c >= T) return;
  {
    const double* Ldk = A.Ld[sys] + (size_t)c * NB * NB;
    const double* Wk = A.Winv[sys] + (size_t)c * 1024;
#pragma unroll
    for (int b = 1; b < 4; ++b)
#pragma unroll
      for (int a = 0; a < b; ++a) Lo[b * (b - 1) / 2 + a][tid] = Ldk[(size_t)(16 * a + (tid >> 4)) * NB + 16 * b + (tid & 15)];
#pragma unroll
    for (int b = 0; b < 4; ++b) Ws[b][tid] = Wk[(size_t)b * 256 + tid];
  }
  for (int e = tid; e < NB * (NB + 1); e += 256) (&Ms[0][0])[e] = 0.0;      // (zeros above the diagonal blocks)
  __syncthreads();
  {
    // M_bb = W_b ; M_ba = -W_b sum_{m = a}^{b-1} L_bm M_ma  (b > a), by distance from the diagonal (as in bwd_chain_body)
    const int r = tid & 15, cc = tid >> 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) Ms[16 * b + r][16 * b + cc] = Ws[b][cc * 16 + r];
    __syncthreads();
#pragma unroll
    for (int d = 1; d < 4; ++d) {
#pragma unroll
      for (int a = 0; a + d < 4; ++a) {
        const int b = a + d;
        double t = 0.0;
#pragma unroll
        for (int m = a; m < b; ++m)
#pragma unroll
          for (int n = 0; n < 16; ++n) t += Lo[b * (b - 1) / 2 + m][n * 16 + r] * Ms[16 * m + n][16 * a + cc];
        tmp[a][cc * 16 + r] = t;
      }
      __syncthreads();
#pragma unroll
      for (int a = 0; a + d < 4; ++a) {
        const int b = a + d;
        double v = 0.0;
#pragma unroll
        for (int n = 0; n < 16; ++n) v += Ws[b][n * 16 + r] * tmp[a][cc * 16 + n];
        Ms[16 * b + r][16 * a + cc] = -v;
      }
      __syncthreads();
    }
  }
  const int i = tid >> 2, part = tid & 3;
  double* tab = A.ctab[sys] + (size_t)c * (NB * NB) + (size_t)tid * 16;
  const size_t tstride = (size_t)T * (NB * NB);
#pragma unroll
  for (int k = 0; k < 16; ++k) tab[TAB_MF * tstride + k] = Ms[i][16 * part + k];      // (zero above the diagonal already)
#pragma unroll
  for (int r = 0; r < 16; ++r) tab[TAB_MB * tstride + r] = Ms[16 * part + r][i];
  const int ld = A.ld[sys];
  const int* prof = A.prof[sys];
  // P_f = M T(c, c-1): rows of block c, columns of block c-1
  {
    double pf[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) pf[k] = 0.0;
    const bool have = c >= 1 && (prof ? prof[c - 1] >= c : true);
    if (have) {
      const double* tp = A.S[sys] + (size_t)((c - 1) * NB) * ld + (size_t)c * NB;
      for (int e = tid; e < NB * NB; e += 256) Ts[e >> 6][e & 63] = tp[(size_t)(e >> 6) * ld + (e & 63)];
    }
    __syncthreads();
    if (have) {
      for (int m = 0; m <= i; ++m) {
        const double a = Ms[i][m];
#pragma unroll
        for (int k = 0; k < 16; ++k) pf[k] += a * Ts[16 * part + k][m];
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) tab[TAB_PF * tstride + k] = pf[k];
    __syncthreads();
  }
  // P_b = T(c+1, c) M: rows of block c+1, columns of block c
  {
    double pb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pb[r] = 0.0;
    const bool have = c + 1 < T && (prof ? prof[c] >= c + 1 : true);
    if (have) {
      const double* tp = A.S[sys] + (size_t)(c * NB) * ld + (size_t)(c + 1) * NB;
      for (int e = tid; e < NB * NB; e += 256) Ts[e >> 6][e & 63] = tp[(size_t)(e >> 6) * ld + (e & 63)];
    }
    __syncthreads();
    if (have) {
      for (int m = i; m < NB; ++m) {
        const double a = Ms[m][i];
#pragma unroll
        for (int r = 0; r < 16; ++r) pb[r] += Ts[m][16 * part + r] * a;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) tab[TAB_PB * tstride + r] = pb[r];
  }
}

// One block of a chained substitution with the tables (see above; the f64 variant solves the factorisation's own right-hand side in a
// joint-solve pass, the f32 variant is the preconditioner).  FWD: block row c, tiles (c, j), j0 <= j < c, x_j polled in ascending
// order; the last, x_{c-1}, meets P_f.  BWD: block column c, tiles (j, c), c < j <= prof[c], polled in descending order; the last,
// x_{c+1}, meets P_b.  Waves 0..3 own the tiles, wave 4 polls (CHAIN_THREADS).
struct alignas(16) TabLds {
  alignas(16) double xs[2][NB];
  alignas(16) double ys[NB];
  alignas(16) double red[4][NB];
};
template <bool FWD, bool F32>
__device__ __forceinline__ void tab_chain_body(TabLds& W, const double* __restrict__ S, int ld, int T, const float* __restrict__ L32,
                                               const double* __restrict__ ctab, const double* __restrict__ rin, double* xout, int* status, int c,
                                               const int* __restrict__ prof, const int* __restrict__ first, double* __restrict__ next_out) {
  auto& xs = W.xs; auto& ys = W.ys; auto& red = W.red;
  const int tid = threadIdx.x;
  const bool worker = tid < 256;
  // the substitution that follows this one (the backward chain after the forward one) finds its output pre-filled with the sentinel
  if (next_out && !worker) next_out[(size_t)c * NB + (tid - 256)] = __longlong_as_double((long long)BWD_SENT);
  const int i = (tid >> 2) & 63, part = tid & 3;       // table products: row / column i, entries 16 part ..
  const int row = tid & 63, cp = (tid >> 6) & 3;       // forward tile products: row `row`, columns 16 cp ..
  const int jlo = FWD ? (first ? first[c] : 0) : c + 1;
  const int jhi = FWD ? c - 1 : (prof ? prof[c] : T - 1);      // tiles jlo .. jhi beside the diagonal
  const int nj = jhi - jlo + 1;
  const bool has_p = nj >= 1;                          // the neighbour tile: folded into P
  const int nloop = has_p ? nj - 1 : 0;                // the others, in the order their x arrive: FWD j = jlo + q, BWD j = jhi - q
  constexpr int RB = F32 ? 6 : 3;
  typedef typename std::conditional<F32, float, double>::type tile_t;
  tile_t tr[RB][16];
  auto tile_load = [&](tile_t (&dst)[16], int q) {
    const int j = FWD ? jlo + q : jhi - q;
    if (FWD) {
      if (F32) {
        const float* tp = L32 + ((size_t)j * (T - 1) - (size_t)j * (j - 1) / 2 + (c - j - 1)) * (NB * NB) + 16 * cp * NB + row;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r] = tp[r * NB];
      } else {
        const double* tp = S + (size_t)(j * NB + 16 * cp) * ld + (size_t)c * NB + row;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r] = tp[(size_t)r * ld];
      }
    } else {
      if (F32) {
        const float4* tp = reinterpret_cast<const float4*>(L32 + ((size_t)c * (T - 1) - (size_t)c * (c - 1) / 2 + (j - c - 1)) * (NB * NB) + i * NB + 16 * part);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float4 v = tp[r];
          dst[4 * r] = v.x; dst[4 * r + 1] = v.y; dst[4 * r + 2] = v.z; dst[4 * r + 3] = v.w;
        }
      } else {
        const double* tp = S + (size_t)(c * NB + i) * ld + (size_t)j * NB + 16 * part;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r] = tp[r];
      }
    }
  };
  double mt[16], pt[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) mt[k] = pt[k] = 0.0;
  double r0 = 0.0;
  if (worker) {
#pragma unroll
    for (int q = 0; q < RB; ++q)
      if (q < nloop) tile_load(tr[q], q);
    const size_t tstride = (size_t)T * (NB * NB);
    const double* tab = ctab + (size_t)c * (NB * NB) + (size_t)tid * 16;
#pragma unroll
    for (int k = 0; k < 16; ++k) mt[k] = tab[(FWD ? TAB_MF : TAB_MB) * tstride + k];
#pragma unroll
    for (int k = 0; k < 16; ++k) pt[k] = has_p ? tab[(FWD ? TAB_PF : TAB_PB) * tstride + k] : 0.0;
    r0 = rin[c * NB + (FWD ? row : i)];
  }
  auto poll = [&](int j, int slot) {                   // the polling wave: block x_j -> xs[slot]
    const int lane = tid - 256;
    double v;
    int spins = 0;
    for (;;) {
      v = __hip_atomic_load(xout + (size_t)j * NB + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned long long)__double_as_longlong(v) != BWD_SENT) break;
      if (++spins > (1 << 21)) {                       // exit condition every wave reaches: give up (seconds), flag the solve as failed
        v = __builtin_nan("");
        atomicOr(&status[1], 2);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    xs[slot][lane] = v;
  };
  double acc = 0.0;
  for (int q0 = 0; q0 < nloop; q0 += RB) {
#pragma unroll
    for (int qq = 0; qq < RB; ++qq) {
      const int q = q0 + qq;
      if (q < nloop) {
        if (!worker) poll(FWD ? jlo + q : jhi - q, q & 1);
        __syncthreads();
        if (worker) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc += (double)tr[qq][r] * xs[q & 1][16 * (FWD ? cp : part) + r];
          if (q + RB < nloop) tile_load(tr[qq], q + RB);
        }
      }
    }
  }
  // right-hand side of the block without its neighbour's term, in ys
  if (FWD) {
    if (worker) red[cp][row] = acc;
    __syncthreads();
    if (tid < NB) ys[row] = r0 - ((red[0][row] + red[1][row]) + (red[2][row] + red[3][row]));
  } else {
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (worker && part == 0) ys[i] = r0 - acc;
  }
  __syncthreads();
  if (!worker && has_p) poll(FWD ? c - 1 : c + 1, nloop & 1);      // (while the others finish z)
  double z = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) z += mt[k] * ys[16 * part + k];
  z += __shfl_xor(z, 1);
  z += __shfl_xor(z, 2);
  if (has_p) {
    __syncthreads();
    double pv = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) pv += pt[k] * xs[nloop & 1][16 * part + k];
    pv += __shfl_xor(pv, 1);
    pv += __shfl_xor(pv, 2);
    z -= pv;
  }
  if (worker && part == 0) __hip_atomic_store(xout + (size_t)c * NB + i, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Both triangular solves for up to CHOL_BATCH_MAX systems with the tables: fwd (ticket counter status[0][5]) or bwd (status[0][4]);
// in / out: per system, T * NB doubles; out pre-filled with the sentinel, the ticket counter cleared (k_chain_prepare).
struct ChainBatchArgs {
  int n;
  const double* S[CHOL_BATCH_MAX]; int ld[CHOL_BATCH_MAX]; int T[CHOL_BATCH_MAX];
  const double* ctab[CHOL_BATCH_MAX]; const double* in[CHOL_BATCH_MAX]; double* out[CHOL_BATCH_MAX];
  int* status[CHOL_BATCH_MAX];
  const float* L32[CHOL_BATCH_MAX];    // packed f32 copy of the factor (written by the type-A workgroups of the factorisation)
  const int* prof[CHOL_BATCH_MAX];     // device: profile of the factor and, per block row, the first block column reaching it; or null
  const int* first[CHOL_BATCH_MAX];
  double* next_out[CHOL_BATCH_MAX];    // output of the chain launched after this one: sentinel-filled here, its ticket counter cleared (or null)
  int* next_ticket;
  int Tmax;
};
template <bool FWD, bool F32>
__global__ __launch_bounds__(CHAIN_THREADS) void k_chain_batched(ChainBatchArgs A) {
  __shared__ TabLds W;
  for (;;) {
    const int t = bwd_ticket(&A.status[0][FWD ? 5 : 4]);
    if (t >= A.n * A.Tmax) return;
    const int r = t % A.n, b = t / A.n;
    if (t == 0 && A.next_ticket && threadIdx.x == 0) *A.next_ticket = 0;
    if (b >= A.T[r]) continue;
    tab_chain_body<FWD, F32>(W, A.S[r], A.ld[r], A.T[r], A.L32[r], A.ctab[r], A.in[r], A.out[r], A.status[r], FWD ? b : A.T[r] - 1 - b,
                             A.prof[r], A.first[r], A.next_out[r]);
    return;
  }
}
struct ChainPrepArgs { int n; double* out[CHOL_BATCH_MAX]; int len[CHOL_BATCH_MAX]; int* ticket; };
__global__ void k_chain_prepare(ChainPrepArgs A) {
  const int r = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0 && i == 0) *A.ticket = 0;
  if (i < A.len[r]) A.out[r][i] = __longlong_as_double((long long)BWD_SENT);
}
// the tables of every system of the batch: once per factorisation in a joint-solve pass (systems with L32 and ctab)
static bool chain_has_tables(const CholSystem* d, int n) {
  for (int i = 0; i < n; ++i)
    if (!d[i].L32 || !d[i].ctab) return false;
  return n > 0;
}
void launch_chain_tables(const CholSystem* d, int n, hipStream_t s) {
  if (!chain_has_tables(d, n)) return;
  ChainTabArgs A{};
  A.n = n;
  int Tmax = 0;
  for (int i = 0; i < n; ++i) {
    A.T[i] = d[i].T; A.S[i] = d[i].S; A.ld[i] = d[i].ld; A.Ld[i] = d[i].Ld; A.Winv[i] = d[i].Winv; A.prof[i] = d[i].prof; A.ctab[i] = d[i].ctab;
    Tmax = d[i].T > Tmax ? d[i].T : Tmax;
  }
  if (Tmax > 0) hipLaunchKernelGGL(k_chain_tables, dim3(Tmax, n), dim3(256), 0, s, A);
}
// x = L^-1 in (fwd) or x = L^-T in (bwd) for every system of the batch (tables required); f32: tiles from the packed f32 copy (the
// preconditioner), else from S.  out[i] must be sentinel-filled and the ticket counter clear: `prepared` = the caller's kernels did
// that (k_chol_extract_y, k_pcg_update, the chain before — next_out), else one extra launch does.  next_out (or null): the outputs of
// the chain that will be launched after this one, prepared here.
void launch_chain_batch(const CholSystem* d, int n, const double* const* in, double* const* out, bool fwd, bool f32, bool prepared,
                        double* const* next_out, hipStream_t s) {
  ChainBatchArgs A{};
  ChainPrepArgs Pr{};
  A.n = n;
  Pr.n = n;
  int Tmax = 0, total = 0;
  for (int i = 0; i < n; ++i) {
    A.S[i] = d[i].S; A.ld[i] = d[i].ld; A.T[i] = d[i].T; A.ctab[i] = d[i].ctab; A.in[i] = in[i]; A.out[i] = out[i];
    A.status[i] = d[i].status;
    A.L32[i] = d[i].L32;
    A.prof[i] = d[i].prof; A.first[i] = d[i].first;
    Tmax = d[i].T > Tmax ? d[i].T : Tmax;
    total += d[i].T;
    Pr.out[i] = out[i];
    Pr.len[i] = d[i].T * NB;
    A.next_out[i] = next_out ? next_out[i] : nullptr;
  }
  A.next_ticket = next_out ? d[0].status + (fwd ? 4 : 5) : nullptr;
  A.Tmax = Tmax;
  if (total <= 0) return;
  if (!chain_has_tables(d, n)) { fprintf(stderr, "slide_slam_amd: launch_chain_batch without the tables / the packed f32 factor\n"); abort(); }
  Pr.ticket = d[0].status + (fwd ? 5 : 4);
  if (!prepared) hipLaunchKernelGGL(k_chain_prepare, dim3((Tmax * NB + 255) / 256, n), dim3(256), 0, s, Pr);
  if (fwd) {
    if (f32) hipLaunchKernelGGL((k_chain_batched<true, true>), dim3(total), dim3(CHAIN_THREADS), 0, s, A);
    else hipLaunchKernelGGL((k_chain_batched<true, false>), dim3(total), dim3(CHAIN_THREADS), 0, s, A);
  } else {
    if (f32) hipLaunchKernelGGL((k_chain_batched<false, true>), dim3(total), dim3(CHAIN_THREADS), 0, s, A);
    else hipLaunchKernelGGL((k_chain_batched<false, false>), dim3(total), dim3(CHAIN_THREADS), 0, s, A);
  }
}

// ---- marginal covariance of one pose (getPoseCovariance graph.cpp:314-323) ---------------------------------------------------
// Cov = E^T S^-1 E = Y^T Y with L Y = E (E = the six unit columns of the pose): a forward substitution with six right-hand
// sides on the finished factor (panel tiles in S, diagonal blocks as their off-diagonal 16x16 sub-tiles in Ld plus the 16x16
// inverses in Winv), one launch per block row k from the pose's block on: every workgroup redoes the 64 x 6 solve of block
// k, workgroup 0 stores it, workgroup b > 0 applies tile (k + b, k) to its own rows of Y.   Y: 6 columns of nT = T * 64 rows.
__global__ __launch_bounds__(256) void k_cov_fwd(const double* __restrict__ S, int ld, int k, const double* __restrict__ Ldk,
                                                 const double* __restrict__ Wk, double* __restrict__ Y, int nT) {
  __shared__ double yk[6][NB];
  __shared__ double xk[6][NB];
  const int tid = threadIdx.x;
  for (int e = tid; e < 6 * NB; e += 256) yk[e / NB][e % NB] = Y[(size_t)(e / NB) * nT + (size_t)k * NB + e % NB];
  __syncthreads();
  const int c = tid >> 4, r = tid & 15;            // column c < 6 of the right-hand sides, row r of a 16-block
#pragma unroll 1
  for (int b = 0; b < 4; ++b) {
    if (c < 6) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) s += Wk[(size_t)b * 256 + j * 16 + r] * yk[c][16 * b + j];     // (L_bb^-1)[r][j]
      xk[c][16 * b + r] = s;
    }
    __syncthreads();
    for (int e = tid; e < 6 * 16 * (3 - b); e += 256) {      // rows below block b: y[m] -= sum_n L[m][16b + n] x_b[n]
      const int cc = e / (16 * (3 - b)), m = 16 * (b + 1) + e % (16 * (3 - b));
      double s = 0.0;
#pragma unroll
      for (int n = 0; n < 16; ++n) s += Ldk[(size_t)(16 * b + n) * NB + m] * xk[cc][16 * b + n];
      yk[cc][m] -= s;
    }
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    for (int e = tid; e < 6 * NB; e += 256) Y[(size_t)(e / NB) * nT + (size_t)k * NB + e % NB] = xk[e / NB][e % NB];
    return;
  }
  const int i = k + (int)blockIdx.x;               // row tile below
  const double* tile = S + (size_t)(k * NB) * ld + (size_t)i * NB;
  for (int e = tid; e < 6 * NB; e += 256) {
    const int cc = e / NB, row = e % NB;
    double s = 0.0;
#pragma unroll 8
    for (int q = 0; q < NB; ++q) s += tile[(size_t)q * ld + row] * xk[cc][q];
    Y[(size_t)cc * nT + (size_t)i * NB + row] -= s;
  }
}
__global__ __launch_bounds__(256) void k_cov_gram(const double* __restrict__ Y, int nT, int row0, double* __restrict__ cov36) {
  __shared__ double part[36][257];
  const int tid = threadIdx.x;
  double acc[36];
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = 0.0;
  for (int row = row0 + tid; row < nT; row += 256) {
    double y[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) y[c] = Y[(size_t)c * nT + row];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) acc[6 * a + b] += y[a] * y[b];
  }
#pragma unroll
  for (int e = 0; e < 36; ++e) part[e][tid] = acc[e];
  __syncthreads();
  if (tid < 36) {
    double s = 0.0;
    for (int q = 0; q < 256; ++q) s += part[tid][q];
    cov36[tid] = s;
  }
}
// Y (6 * T * NB doubles) must hold the six unit columns of rows row0 .. row0 + 5 (zero elsewhere); cov36 device
void launch_pose_covariance(const double* S, int ld, int T, const double* Ld, const double* Winv, double* Y, int row0, double* cov36,
                            hipStream_t s) {
  const int nT = T * NB;
  for (int k = row0 / NB; k < T; ++k)
    hipLaunchKernelGGL(k_cov_fwd, dim3((unsigned)(T - k)), dim3(256), 0, s, S, ld, k, Ld + (size_t)k * NB * NB, Winv + (size_t)k * 1024, Y, nT);
  hipLaunchKernelGGL(k_cov_gram, dim3(1), dim3(256), 0, s, Y, nT, row0 / NB * NB, cov36);
}

// ---- the border of a bordered system (exact joint step of several robots) --------------------------------------------------------------
// After the steps over the band's T block columns the nbr border row tiles hold W^T = B^T L^-T (B: coupling of the poses to the
// separator variables) and the right-hand-side row y^T = (L^-1 b)^T.  The Schur complement of the band onto the separator,
//     bord(i, j) -= sum_c W^T(i, c) W^T(j, c)^T          i >= j border tile rows, i = nbr: the right-hand-side row (b_s - W^T y),
// is ONE product over the T column blocks — a real GEMM (K = 64 T), every output tile independent: one workgroup per 64x64 tile, four
// waves with a 32x32 quadrant each (the quadrant of b_quadrant: four 16x16 accumulators, operands as 16-byte loads three k-steps
// ahead), C read and written once.  bfirst[i]: first column block in which border tile row i can be non-zero (a landmark first seen
// late in the trajectory has an all-zero W^T row up to there): the sum starts at max(bfirst[i], bfirst[j]).
// Split K (ks > 1, one system with a handful of border tiles — the separator's own lambda border): blockIdx.z is the chunk of the column
// blocks; chunk 0 works on the tile itself, chunk q > 0 leaves its partial product in scratch tile (tile, q - 1) and k_border_syrk_reduce
// adds the partials in chunk order (deterministic, no atomics).
struct SyrkArgs {
  int n;
  const double* S[CHOL_BATCH_MAX]; int ld[CHOL_BATCH_MAX]; int T[CHOL_BATCH_MAX]; int nbr[CHOL_BATCH_MAX];
  int b0[CHOL_BATCH_MAX];              // tile row of the first border row (T, or CholSystem::b0 for a view that covers a range of the columns only)
  double* bord[CHOL_BATCH_MAX]; int ldb[CHOL_BATCH_MAX]; const int* bfirst[CHOL_BATCH_MAX];
  const double* bsrc[CHOL_BATCH_MAX];  // or null: where the tiles' values before the product are read (CholSystem::bord_src); null: bord itself
  int ks; double* scratch;
  size_t scratch_stride;     // split K with two systems in one launch: system r's partial tiles start at scratch + r * scratch_stride
  int ks_sys[2];             // ... and each may have its own number of chunks (0: ks) — a leaf's product is cut the way a rank owning that leaf cuts it
  const int* jobs;       // or null: (system << 20 | ib << 10 | jb) per workgroup, longest sums first (launch_border_syrk_jobs)
  const int* segtab[CHOL_BATCH_MAX];      // or null: CholSystem::segtab — the sum runs over the segments in which both tile rows are non-zero
};
template <int RD>
__device__ __forceinline__ void border_syrk_body(const SyrkArgs& A) {
  int r = A.ks > 1 ? 0 : blockIdx.z, ib = blockIdx.x, jb = blockIdx.y;
  const int q = A.ks > 1 ? blockIdx.z : 0;
  if (A.jobs) {
    const int j = A.jobs[blockIdx.x];
    r = j >> 20; ib = (j >> 10) & 1023; jb = j & 1023;
  }
  const int nbr = A.nbr[r];
  if (ib > nbr || jb >= nbr || ib < jb) return;
  const int T = A.T[r], ld = A.ld[r];
  int ldb = A.ldb[r];
  int c0 = 0, c1 = T;                             // column blocks [c0, c1) of the band
  const int* st = A.segtab[r];                    // a segmented band: one range per segment in which both tile rows are non-zero
  const int nseg = st ? st[0] : 1;
  if (!st && A.bfirst[r]) c0 = max(A.bfirst[r][ib], A.bfirst[r][jb]);
  int ksr = A.ks;                                 // chunks of this system's column blocks
  if (A.ks > 1) {
    if (A.scratch_stride && r < 2 && A.ks_sys[r] > 0) ksr = A.ks_sys[r];
    if (q >= ksr) return;
    const int len = (T - c0 + ksr - 1) / ksr;
    c0 += q * len;
    c1 = min(T, c0 + len);
  }
  const double* bsrc = A.bsrc[r];
  if (!st && c0 >= c1 && !(bsrc && q == 0)) return;      // (nothing to sum; with a separate source the tile is still copied)
  const double* S = A.S[r];
  const int wq = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int ch = (wq >> 1) & 1, rh = wq & 1;      // column half, row half of the tile
  if (A.jobs && ib == nbr && rh == 1 && !bsrc) return;     // right-hand-side row tile: only its first row is in use (the others are zero and stay zero)
  const int B0 = A.b0[r];
  const double* pj0 = S + (size_t)lk * ld + (size_t)(B0 + jb) * NB + 32 * ch + 2 * lr;
  const double* pi0 = S + (size_t)lk * ld + (size_t)(B0 + ib) * NB + 32 * rh + 2 * lr;
  double* cbh = A.bord[r] + (size_t)(jb * NB + 32 * ch + 2 * lk) * ldb + (size_t)ib * NB + 32 * rh + 2 * lr;
  const double* cin = bsrc ? bsrc + (size_t)(jb * NB + 32 * ch + 2 * lk) * ldb + (size_t)ib * NB + 32 * rh + 2 * lr : cbh;
  const int ldin = ldb;
  if (q > 0) {       // partial of a later chunk: a 64 x 64 scratch tile (leading dimension NB)
    ldb = NB;
    cbh = A.scratch + (size_t)r * A.scratch_stride + ((size_t)(jb * (nbr + 1) + ib) * (ksr - 1) + (q - 1)) * (NB * NB) + (size_t)(32 * ch + 2 * lk) * NB + 32 * rh + 2 * lr;
  }
  // The k-steps of ALL segments as ONE sequence (round 4): the operand loads run RD - 1 steps ahead of the matrix pipe ACROSS segment
  // boundaries — a cut band's job has three short sums (one to three block columns each), and a prefetch pipeline that drains and
  // refills per segment shows the full load latency three times per job (the kernel is bound by that latency: RD 4 -> 16 alone took the
  // robots' launch from 0.196 to 0.148 ms).  Same order of the sum (segments ascending, columns ascending): same bits.
  int total = 0;                                  // k-steps of four columns
  if (st) {
    for (int sg = 0; sg < nseg; ++sg) {
      const int* sf = st + 1 + nseg + sg * (nbr + 1);
      const int a0 = max(sf[ib], sf[jb]), a1 = st[1 + sg];
      if (a0 < a1) total += (a1 - a0) * 16;       // (1 << 30: zero in this segment)
    }
  } else {
    total = c1 > c0 ? (c1 - c0) * 16 : 0;
  }
  int sg_next = 0, rem = 0, issued = 0;           // load side: next segment to open, k-steps left in the open one, steps issued
  const double *lpj = pj0, *lpi = pi0;
  v2d pa[RD], pb[RD];
  auto issue = [&](v2d& ra, v2d& rb) {            // (wave-uniform control flow)
    while (rem == 0 && sg_next < nseg) {
      int a0 = c0, a1 = c1;
      if (st) {
        const int* sf = st + 1 + nseg + sg_next * (nbr + 1);
        a0 = max(sf[ib], sf[jb]);
        a1 = st[1 + sg_next];
      }
      ++sg_next;
      if (a0 < a1) {
        rem = (a1 - a0) * 16;
        lpj = pj0 + (size_t)a0 * NB * ld;
        lpi = pi0 + (size_t)a0 * NB * ld;
      }
    }
    ra = *(const v2d*)lpj;
    rb = *(const v2d*)lpi;
    lpj += (size_t)4 * ld;
    lpi += (size_t)4 * ld;
    --rem;
    ++issued;
  };
#pragma unroll
  for (int pre = 0; pre < RD - 1; ++pre)
    if (issued < total) issue(pa[pre], pb[pre]);
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v2d c2;
      c2[0] = 0.0; c2[1] = 0.0;
      if (q == 0) c2 = *(const v2d*)(cin + (size_t)(8 * e + a) * ldin);
      acc[a][0][e] = c2[0]; acc[a][1][e] = c2[1];
    }
  for (int ks0 = 0; ks0 < total; ks0 += RD) {     // (every segment's length is a multiple of 16 k-steps: RD divides the total)
#pragma unroll
    for (int u = 0; u < RD; ++u) {
      if (issued < total) issue(pa[(u + RD - 1) % RD], pb[(u + RD - 1) % RD]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const double na = -pa[u][a];
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma_f64(na, pb[u][b], acc[a][b]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v2d c2;
      c2[0] = acc[a][0][e]; c2[1] = acc[a][1][e];
      *(v2d*)(cbh + (size_t)(8 * e + a) * ldb) = c2;
    }
}
// operand loads fifteen k-steps ahead of the matrix pipe (142 VGPRs, two waves per SIMD): the kernel is bound by the latency of its
// operand loads, not by occupancy — C4's robots' launch: 0.175 ms with loads three steps ahead (round 3's depth), 0.159 with seven,
// 0.138 with fifteen (0.56 of the FP64 MFMA peak).  SLIDE_SYRK_RD=4 / 8: the shallower variants
__global__ __launch_bounds__(256) void k_border_syrk(SyrkArgs A) { border_syrk_body<16>(A); }
__global__ __launch_bounds__(256) void k_border_syrk_rd8(SyrkArgs A) { border_syrk_body<8>(A); }
__global__ __launch_bounds__(256) void k_border_syrk_rd4(SyrkArgs A) { border_syrk_body<4>(A); }
static void launch_syrk_kernel(dim3 grid, size_t lds, hipStream_t s, const SyrkArgs& A) {
  static const int rd = getenv("SLIDE_SYRK_RD") ? atoi(getenv("SLIDE_SYRK_RD")) : 16;
  if (rd == 4) hipLaunchKernelGGL(k_border_syrk_rd4, grid, dim3(256), lds, s, A);
  else if (rd == 8) hipLaunchKernelGGL(k_border_syrk_rd8, grid, dim3(256), lds, s, A);
  else hipLaunchKernelGGL(k_border_syrk, grid, dim3(256), lds, s, A);
}
__global__ __launch_bounds__(256) void k_border_syrk_reduce(double* __restrict__ bord, int ldb, int nbr, int ks, const double* __restrict__ scratch,
                                                            const int* __restrict__ bfirst, int T) {
  // one element per thread: blockIdx.x = 16 ib + the tile's sixteenth (a workgroup per tile looping over its 4096 elements took 54 us
  // for ~180 tiles, all of it load latency); gridDim.y may stop short of nbr: only the tile columns the product was run for
  const int ib = blockIdx.x >> 4, jb = blockIdx.y;
  if (ib > nbr || jb >= nbr || ib < jb) return;
  int c0 = 0;
  if (bfirst) c0 = max(bfirst[ib], bfirst[jb]);
  const int len = (T - c0 + ks - 1) / ks;
  const int e = (blockIdx.x & 15) * 256 + threadIdx.x;
  const int col = e / NB, row = e - col * NB;
  double* dst = bord + (size_t)(jb * NB + col) * ldb + (size_t)ib * NB + row;
  double v = *dst;
  for (int q = 1; q < ks; ++q)
    if (c0 + q * len < T) v += scratch[((size_t)(jb * (nbr + 1) + ib) * (ks - 1) + (q - 1)) * (NB * NB) + (size_t)col * NB + row];
  *dst = v;
}
// Two systems with the same border whose results are ADDED (a canonical whole pass: each half's partial top block minus its leaf's Schur
// complement): A <- (A + its partials, in chunk order) + (B + its partials, in chunk order) — what two ranks owning a leaf each compute
// and then exchange, bit for bit, in one launch instead of two reductions and a sum
__global__ __launch_bounds__(256) void k_border_syrk_reduce2(double* __restrict__ bordA, int ldbA, const double* __restrict__ bordB, int ldbB, int nbr,
                                                             int ksA, int ksB, const double* __restrict__ scratchA, const double* __restrict__ scratchB,
                                                             int TA, int TB) {
  const int ib = blockIdx.x >> 4, jb = blockIdx.y;
  if (ib > nbr || jb >= nbr || ib < jb) return;
  const int e = (blockIdx.x & 15) * 256 + threadIdx.x;
  const int col = e / NB, row = e - col * NB;
  double* dst = bordA + (size_t)(jb * NB + col) * ldbA + (size_t)ib * NB + row;
  double va = *dst, vb = bordB[(size_t)(jb * NB + col) * ldbB + (size_t)ib * NB + row];
  const int lenA = (TA + ksA - 1) / ksA, lenB = (TB + ksB - 1) / ksB;
  for (int q = 1; q < ksA; ++q)
    if (q * lenA < TA) va += scratchA[((size_t)(jb * (nbr + 1) + ib) * (ksA - 1) + (q - 1)) * (NB * NB) + (size_t)col * NB + row];
  for (int q = 1; q < ksB; ++q)
    if (q * lenB < TB) vb += scratchB[((size_t)(jb * (nbr + 1) + ib) * (ksB - 1) + (q - 1)) * (NB * NB) + (size_t)col * NB + row];
  *dst = va + vb;
}
// scratch (or null): (nbr + 1) * nbr * (ks - 1) tiles of NB * NB doubles — with it, ONE system's product is split over ks chunks of its
// column blocks (a system with a handful of border tiles would otherwise occupy a handful of CUs for the whole K)
void launch_border_syrk(const CholSystem* d, int n, hipStream_t s, double* scratch, int ks) {
  SyrkArgs A{};
  A.n = n;
  int nb = 0;
  for (int i = 0; i < n; ++i) {
    A.S[i] = d[i].S; A.ld[i] = d[i].ld; A.T[i] = d[i].T; A.nbr[i] = d[i].nbr; A.bord[i] = d[i].bord; A.ldb[i] = d[i].ldb; A.bfirst[i] = d[i].bfirst; A.bsrc[i] = d[i].bord_src;
    A.segtab[i] = d[i].segtab; A.b0[i] = d[i].b0 > 0 ? d[i].b0 : d[i].T;
    nb = d[i].nbr > nb ? d[i].nbr : nb;
  }
  if (nb <= 0) return;
  if (scratch && n == 1 && ks > 1) {
    A.ks = ks; A.scratch = scratch;
    launch_syrk_kernel(dim3(nb + 1, nb, ks), 0, s, A);
    hipLaunchKernelGGL(k_border_syrk_reduce, dim3(16 * (nb + 1), nb), dim3(256), 0, s, d[0].bord, d[0].ldb, d[0].nbr, ks, scratch, d[0].bfirst, d[0].T);
    return;
  }
  A.ks = 1; A.scratch = nullptr;
  launch_syrk_kernel(dim3(nb + 1, nb, n), 0, s, A);
}
// The same product with the workgroups in the order of a job table (all systems' lower tiles + right-hand-side rows, longest sum first)
// and lds_pad bytes of idle dynamic LDS per workgroup to bound the workgroups resident on a CU: the hardware dispatcher then hands the
// next job to whichever slot frees first — longest-processing-time list scheduling.  With the plain (ib, jb, system) grid all ~1800
// workgroups of eight robots become resident at once, six or seven per CU, and a CU's finishing time is the sum of whatever it was
// dealt (sums of 0 .. T column blocks: the slowest CU carries ~1.6 x the mean).
// scratch + ks > 1 (one system): split K as in launch_border_syrk; jb_end >= 0: the table holds the tile columns jb < jb_end only (the
// rest of the border block is somebody else's: the lambda block of the separator system)
void launch_border_syrk_jobs(const CholSystem* d, int n, const int* jobs, int njobs, int lds_pad, hipStream_t s, double* scratch, int ks, int jb_end,
                             const int* ks_sys, bool fuse_add) {
  SyrkArgs A{};
  A.n = n;
  for (int i = 0; i < n; ++i) {
    A.S[i] = d[i].S; A.ld[i] = d[i].ld; A.T[i] = d[i].T; A.nbr[i] = d[i].nbr; A.bord[i] = d[i].bord; A.ldb[i] = d[i].ldb; A.bfirst[i] = d[i].bfirst; A.bsrc[i] = d[i].bord_src;
    A.segtab[i] = d[i].segtab; A.b0[i] = d[i].b0 > 0 ? d[i].b0 : d[i].T;
  }
  if (njobs <= 0) return;
  A.ks = 1; A.scratch = nullptr; A.jobs = jobs;
  if (scratch && n == 2 && fuse_add && d[0].nbr == d[1].nbr && !d[0].bfirst && !d[1].bfirst) {
    // two systems with the same border, split K (each with its own number of chunks: ks_sys, or ks), ONE launch — the two leaves' Schur
    // complements onto the two halves' partial top blocks of a canonical whole pass: the job table names the system, each system's
    // partial tiles have their own half of the scratch — and ONE reduction that also adds the second result onto the first
    const int nb = d[0].nbr;
    const int ka = std::max(1, ks_sys ? ks_sys[0] : ks), kb = std::max(1, ks_sys ? ks_sys[1] : ks), km = std::max(ka, kb);
    A.ks = km; A.scratch = scratch; A.scratch_stride = std::max<size_t>(1, (size_t)(nb + 1) * nb * (km - 1) * NB * NB);
    A.ks_sys[0] = ka; A.ks_sys[1] = kb;
    launch_syrk_kernel(dim3(njobs, 1, km), lds_pad, s, A);
    hipLaunchKernelGGL(k_border_syrk_reduce2, dim3(16 * (nb + 1), jb_end >= 0 ? jb_end : nb), dim3(256), 0, s, d[0].bord, d[0].ldb, d[1].bord, d[1].ldb, nb,
                       ka, kb, scratch, scratch + A.scratch_stride, d[0].T, d[1].T);
    return;
  }
  if (scratch && n == 1 && ks > 1) {
    A.ks = ks; A.scratch = scratch;
    launch_syrk_kernel(dim3(njobs, 1, ks), lds_pad, s, A);
    const int nb = d[0].nbr;
    hipLaunchKernelGGL(k_border_syrk_reduce, dim3(16 * (nb + 1), jb_end >= 0 ? jb_end : nb), dim3(256), 0, s, d[0].bord, d[0].ldb, nb, ks, scratch, d[0].bfirst, d[0].T);
    return;
  }
  launch_syrk_kernel(dim3(njobs), lds_pad, s, A);
}
// y -= W x_loc before the backward substitution of the band (x_loc: the separator's solution in the system's own border order, zeros
// in the padding): one wave per column of the band, lanes over the border rows (contiguous down a column of S)
struct BorderApplyArgs {
  int n;
  const double* S[CHOL_BATCH_MAX]; int ld[CHOL_BATCH_MAX]; int T[CHOL_BATCH_MAX]; int nbr[CHOL_BATCH_MAX]; int b0[CHOL_BATCH_MAX];
  double* yv[CHOL_BATCH_MAX]; const double* x[CHOL_BATCH_MAX];
  const int* segtab[CHOL_BATCH_MAX];      // or null: CholSystem::segtab — a cut band's border rows are non-zero in some block columns only
};
__global__ __launch_bounds__(256) void k_border_apply(BorderApplyArgs A) {
  const int r = blockIdx.z;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (col >= A.T[r] * NB) return;
  const int nrow = A.nbr[r] * NB;
  const double* w = A.S[r] + (size_t)col * A.ld[r] + (size_t)A.b0[r] * NB;      // (b0: first border tile row — T, or further down for a view)
  const double* x = A.x[r];
  double acc = 0.0;
  const int* st = A.segtab[r];
  const int* msk = st ? st + 1 + st[0] + st[0] * (A.nbr[r] + 1) : nullptr;      // (behind the segments' table: T, then a 64-bit mask per block column)
  if (msk && msk[0] > 0) {
    const int c = col / NB;
    unsigned long long m = (unsigned long long)(unsigned)msk[1 + 2 * c] | ((unsigned long long)(unsigned)msk[2 + 2 * c] << 32);
    while (m) {      // the tile rows some segment works on in this block column: the others are zero
      const int t = __ffsll((long long)m) - 1;
      m &= m - 1;
      acc += w[t * NB + lane] * x[t * NB + lane];
    }
  } else {
    for (int q = lane; q < nrow; q += 64) acc += w[q] * x[q];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if (lane == 0) A.yv[r][col] -= acc;
}
void launch_border_apply(const CholSystem* d, int n, const double* const* xloc, hipStream_t s) {
  BorderApplyArgs A{};
  A.n = n;
  int Tmax = 0;
  for (int i = 0; i < n; ++i) {
    A.S[i] = d[i].S; A.ld[i] = d[i].ld; A.T[i] = d[i].T; A.nbr[i] = d[i].nbr; A.yv[i] = d[i].yv; A.x[i] = xloc[i];
    A.b0[i] = d[i].b0 > 0 ? d[i].b0 : d[i].T;
    A.segtab[i] = d[i].segtab;
    Tmax = d[i].T > Tmax ? d[i].T : Tmax;
  }
  if (Tmax > 0) hipLaunchKernelGGL(k_border_apply, dim3(Tmax * NB / 4, 1, n), dim3(256), 0, s, A);
}

// ------------------------------------------------------------------------------------------------
// Schedule (see step_type_b): launch 0 factors column 0, launch 1 column 1 with panel 0 (and brings column 2 up to panel 0);
// from then on every launch k takes the one pending panel k-1 in its column, even launches start the rank-128 pass of panels
// k-2, k-1 over the trailing matrix, odd launches finish the pass their predecessor started and bring column k+1 up to panel k-1.
void launch_chol_bwd_all(const double* S, int ld, int T, const double* Ld, const double* Winv, double* yv, double* dp,
                         int* status, const int* prof, hipStream_t s, const double* wf_prev, double wf_thr, int wf_Tprev, int wf_lim);
void launch_chol_extract_y(const double* S, int ld, int T, double* yv, double* dp, int* status, hipStream_t s, int nbr, int b0, double* prev) {
  hipLaunchKernelGGL(k_chol_extract_y, dim3((T * NB + 255) / 256), dim3(256), 0, s, S, ld, T, (b0 > 0 ? b0 : T) + nbr, yv, dp, status, prev);
}
struct StepPlan { int kb, nP, g0, g1, nX, TvA, TvB, TvX, nbA, nbB, nbX; long long nA, nB; };
static int chol_n_cu() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n_cu = pr.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  return n_cu;
}
// prof (host, T ints, or null = dense): prof[c] = last tile row of block column c inside the monotone profile of the factor (>= c).
// Column k's tiles reach row prof[k]; the rank-128 pass of pair base kb (panels kb-2, kb-1) touches rows / columns <= prof[kb-1]; the
// column items of an odd launch (panel k-1 onto column k+1) rows <= prof[k-1].  Everything else is structurally zero and skipped.
// bfirst (host, nbr ints, non-decreasing, or null): first block column in which border tile row i is non-zero; a row whose first column lies
// beyond the panels a launch handles is all-zero there and skipped
static StepPlan plan_step(int k, int T, const int* prof, int nbr = 0, const int* bfirst = nullptr, int kofs = 0) {
  static const double frac = getenv("SLIDE_CHOL_FRAC") ? atof(getenv("SLIDE_CHOL_FRAC")) : 0.5;   // diagnostic: share of a pass done by its first launch
  StepPlan p{};
  p.kb = k & ~1;                                                // base of the pair
  p.TvA = p.TvB = p.TvX = T;
  if (prof && k < T) {
    p.TvA = prof[k] + 1;
    if (p.kb >= 1) p.TvB = prof[p.kb - 1] + 1;
    if (k >= 1) p.TvX = prof[k - 1] + 1;
  }
  auto active = [&](int upto) { int c = 0; while (c < nbr && (!bfirst || bfirst[c] <= upto + kofs)) ++c; return c; };
  p.nbA = active(k); p.nbB = active(p.kb - 1); p.nbX = active(k - 1);
  p.nA = k < T ? p.TvA - k + p.nbA : 0;                         // column-k tiles below the diagonal (+ active border tiles + RHS tile)
  if (k >= 2 && k < T) {
    p.nP = p.TvB + p.nbB > p.kb ? (p.TvB + p.nbB - p.kb + 1) / 2 : 0;     // 2x2 tile groups per side of the trailing matrix of the pair
    long long nG = (long long)p.nP * (p.nP + 1) / 2;
    if (nbr > 0) {
      // the border has rows only: the group columns right of the profile's last column hold no item — the enumeration (column-major
      // over the groups, b_decode) is cut behind the last group column with a real tile column
      const int ncol = p.TvB - 1 >= p.kb + 1 ? (p.TvB - 2 - p.kb) / 2 + 1 : 0;
      long long nv = 0;
      for (int bj = 0; bj < ncol && bj < p.nP; ++bj) nv += p.nP - bj;
      nG = nv;
    }
    long long first = (long long)(frac * (double)nG + 0.5);
    if (first < p.nP) first = p.nP;                             // the group column with tile columns kb+1, kb+2 entirely
    if (first > nG) first = nG;
    if (k & 1) { p.g0 = (int)(2 * first); p.g1 = (int)(2 * nG); } else { p.g0 = 0; p.g1 = (int)(2 * first); }   // items = half groups
  }
  p.nX = (k & 1) && k + 1 < p.TvX ? (p.TvX + p.nbX - k + 1) / 2 : 0;    // column items: tile rows k+1 .. TvX (+ active border) of column k+1, two per item
  p.nB = p.g1 - p.g0 + p.nX;
  return p;
}
void launch_chol_step(double* S, int ld, int k, int T, double* Ld, double* Winv, int* status, int* ctr, float* L32, const int* h_prof,
                      hipStream_t s, int nbr) {
  const int n_cu = chol_n_cu();
  const StepPlan p = plan_step(k, T, h_prof, nbr, nullptr);
  const long long nA = p.nA, nB = p.nB;
  static const int split_pct = getenv("SLIDE_CHOL_ASPLIT") ? atoi(getenv("SLIDE_CHOL_ASPLIT")) : 100;   // diagnostic: 0 = never
  // two workgroups per type-A tile once the launch is bound by the chain, not by the flood
  const int a_split = (k > 0 && (2 * nA + nB) * 100 <= (long long)n_cu * split_pct) ? 1 : 0;
  const long long nAw = nA << a_split;
  const long long extra = nB < n_cu ? nB : n_cu;                // queue workers beside the type-A workgroups (one 512-thread workgroup per CU)
  // type-A workgroups join the queue only when the flood needs more than a round of the other CUs (~ the chain's length)
  const long long free_cu = n_cu - nAw > 8 ? n_cu - nAw : 8;
  static const int join_mul = getenv("SLIDE_CHOL_JOIN") ? atoi(getenv("SLIDE_CHOL_JOIN")) : 1;
  const int a_joins = nB > join_mul * free_cu ? 1 : 0;
  hipLaunchKernelGGL(k_chol_step, dim3((unsigned)(nAw + extra)), dim3(512), 0, s, S, ld, k, T, Ld, Winv, status, ctr, p.kb, p.nP,
                     p.g0, p.g1, p.nX, a_joins, a_split, L32, p.TvA, p.TvB, p.TvX, nbr, p.nbA, p.nbB, p.nbX);
}
// The factorisations + solves of several systems as one launch sequence (max T step launches, the extractions, one chained backward
// substitution): d[i] describes system i; ctr is the work counter array of the batch (max T + 2 ints, zeroed once).
// Pair launches (k_chol_pair_batched): plan of block columns k, k + 1 of one system.  Rows of the launch: the band's tile rows k + ncols ..
// prof[k + ncols - 1], the border rows active at the pair's last column, the right-hand-side row — two workgroups each.  Items: the
// rank-128 pass of panels k-2, k-1 over the tiles (i, j >= k + 2) inside the profile of panel k-1 (+ the border rows active there).
struct PairPlan { int ncols, Tv0, Tv1, nb0, nb1, TvB, nbB, nP; long long nA, nB; };
static PairPlan plan_pair(int k, int T, const int* prof, int nbr, const int* bfirst, int kofs) {
  PairPlan p{};
  p.ncols = T - k >= 2 ? 2 : (T - k == 1 ? 1 : 0);
  if (p.ncols == 0) return p;
  auto active = [&](int upto) { int c = 0; while (c < nbr && (!bfirst || bfirst[c] <= upto + kofs)) ++c; return c; };
  p.Tv0 = prof ? prof[k] + 1 : T;
  p.Tv1 = p.ncols == 2 ? (prof ? prof[k + 1] + 1 : T) : p.Tv0;
  p.nb0 = active(k);
  p.nb1 = p.ncols == 2 ? active(k + 1) : p.nb0;
  const int TvL = p.ncols == 2 ? p.Tv1 : p.Tv0, nbL = p.ncols == 2 ? p.nb1 : p.nb0;
  p.nA = (long long)(TvL - (k + p.ncols)) + nbL + 1;
  if (k >= 2) {
    const int kbE = k + 1;
    p.TvB = prof ? prof[k - 1] + 1 : T;
    p.nbB = active(k - 1);
    p.nP = p.TvB + p.nbB > kbE ? (p.TvB + p.nbB - kbE + 1) / 2 : 0;
    long long nG = (long long)p.nP * (p.nP + 1) / 2;
    if (nbr > 0) {      // (plan_step: the group columns right of the profile's last column hold no item)
      const int ncol = p.TvB - 1 >= kbE + 1 ? (p.TvB - 2 - kbE) / 2 + 1 : 0;
      long long nv = 0;
      for (int bj = 0; bj < ncol && bj < p.nP; ++bj) nv += p.nP - bj;
      nG = nv;
    }
    p.nB = 2 * nG;
  }
  return p;
}
static void launch_chol_pair_steps(const CholSystem* d, int n, int* ctr, int* tickets, hipStream_t s) {
  const int n_cu = chol_n_cu();
  int Tmax = 0;
  for (int i = 0; i < n; ++i) Tmax = d[i].T > Tmax ? d[i].T : Tmax;
  for (int k = 0; k < Tmax; k += 2) {
    CholPairArgs A{};
    A.n = n;
    A.a_base[0] = A.b_base[0] = 0;
    for (int i = 0; i < n; ++i) {
      const PairPlan p = plan_pair(k, d[i].T, d[i].h_prof, d[i].nbr, d[i].h_bfirst, d[i].kofs);
      A.S[i] = d[i].S; A.ld[i] = d[i].ld; A.Ld[i] = d[i].Ld; A.Winv[i] = d[i].Winv; A.status[i] = d[i].status;
      A.ncols[i] = p.ncols; A.Tv0[i] = p.Tv0; A.Tv1[i] = p.Tv1; A.nb0[i] = p.nb0; A.nb1[i] = p.nb1;
      A.TvB[i] = p.TvB; A.nbB[i] = p.nbB; A.nP[i] = p.nP;
      A.nbr[i] = d[i].nbr; A.B0[i] = d[i].b0 > 0 ? d[i].b0 : d[i].T; A.ord[i] = d[i].ord;
      A.a_base[i + 1] = A.a_base[i] + (int)(2 * p.nA);
      A.b_base[i + 1] = A.b_base[i] + (int)p.nB;
    }
    const long long nAw = A.a_base[n], nB = A.b_base[n];
    if (nAw <= 0) continue;
    const long long extra = nB < n_cu ? nB : n_cu;
    const long long free_cu = n_cu - nAw > 8 ? n_cu - nAw : 8;
    const int a_joins = nB > free_cu ? 1 : 0;
    hipLaunchKernelGGL(k_chol_pair_batched, dim3((unsigned)(nAw + extra)), dim3(PAIR_THREADS), 0, s, A, k, ctr, tickets, a_joins);
  }
}
bool chol_pair_supported(const CholSystem* d, int n) {
  if (n < 1 || n > CHOL_STEP_BATCH_MAX) return false;
  for (int i = 0; i < n; ++i)
    if (d[i].L32) return false;      // (the f32 factor copy of the joint solve's preconditioner is written by the step kernels only)
  return true;
}
void launch_chol_batch(const CholSystem* d, int n, int* ctr, hipStream_t s, hipEvent_t after_steps, bool solve, int cu_share, int* pair_tickets) {
  const bool pair = pair_tickets != nullptr;
  const int n_cu = chol_n_cu();
  int Tmax = 0;
  for (int i = 0; i < n; ++i) Tmax = d[i].T > Tmax ? d[i].T : Tmax;
  static const bool verify = getenv("SLIDE_PAIR_VERIFY") && getenv("SLIDE_PAIR_VERIFY")[0] == '1';
  if (pair && chol_pair_supported(d, n) && verify) {
    // diagnostic (needs SLIDE_PASS_DIRECT=1: synchronises the stream): the same systems through the step kernels on copies, the pair
    // kernel on the originals, tile by tile on the host — the first tiles that differ are printed
    (void)hipStreamSynchronize(s);
    std::vector<CholSystem> cp(d, d + n);
    std::vector<size_t> len(n);
    std::vector<void*> tmp;
    auto dmal = [&](size_t bytes) { void* p = nullptr; (void)hipMalloc(&p, bytes); tmp.push_back(p); return p; };
    for (int i = 0; i < n; ++i) {
      const int B0 = d[i].b0 > 0 ? d[i].b0 : d[i].T;
      len[i] = (size_t)d[i].ld * ((size_t)d[i].T * NB - 1) + (size_t)(B0 + d[i].nbr + 1) * NB;
      cp[i].S = (double*)dmal(len[i] * sizeof(double));
      cp[i].Ld = (double*)dmal((size_t)d[i].T * NB * NB * sizeof(double));
      cp[i].Winv = (double*)dmal((size_t)d[i].T * 1024 * sizeof(double));
      cp[i].status = (int*)dmal(8 * sizeof(int));
      cp[i].yv = (double*)dmal((size_t)d[i].T * NB * sizeof(double));
      cp[i].dp = (double*)dmal((size_t)d[i].T * NB * sizeof(double));
      (void)hipMemset(cp[i].status, 0, 8 * sizeof(int));
      (void)hipMemcpy(cp[i].S, d[i].S, len[i] * sizeof(double), hipMemcpyDeviceToDevice);
    }
    int* vctr = (int*)dmal(((size_t)Tmax + 3) * sizeof(int));
    (void)hipMemset(vctr, 0, ((size_t)Tmax + 3) * sizeof(int));
    launch_chol_batch(cp.data(), n, vctr, s, nullptr, false, cu_share, nullptr);
    launch_chol_pair_steps(d, n, ctr, pair_tickets, s);
    (void)hipStreamSynchronize(s);
    int printed = 0;
    for (int i = 0; i < n; ++i) {
      std::vector<double> a(len[i]), b(len[i]);
      (void)hipMemcpy(a.data(), cp[i].S, len[i] * sizeof(double), hipMemcpyDeviceToHost);
      (void)hipMemcpy(b.data(), d[i].S, len[i] * sizeof(double), hipMemcpyDeviceToHost);
      const int ld = d[i].ld, T = d[i].T, B0 = d[i].b0 > 0 ? d[i].b0 : T, rows = B0 + d[i].nbr + 1;
      for (int c = 0; c < T; ++c)
        for (int r = c; r < rows; ++r) {
          if (c == T - 1 && r > B0 + d[i].nbr) continue;
          double worst = 0.0, ref = 0.0;
          for (int x = 0; x < NB; ++x)
            for (int y = 0; y < NB; ++y) {
              if (r == c) continue;      // (the diagonal tiles are not written back)
              const size_t o = (size_t)(c * NB + x) * ld + (size_t)r * NB + y;
              if (o >= len[i]) continue;
              const double dd = std::fabs(a[o] - b[o]);
              if (!(dd <= worst)) worst = dd;
              ref = std::max(ref, std::fabs(a[o]));
            }
          if (worst > 1e-9 * std::max(ref, 1.0) && printed < 40) {
            ++printed;
            fprintf(stderr, "[pair verify] system %d (T %d, nbr %d, b0 %d, kofs %d): tile (%d, %d) differs by %.3e (step kernels' max |value| %.3e)\n", i, T, d[i].nbr, d[i].b0,
                    d[i].kofs, r, c, worst, ref);
          }
        }
    }
    if (!printed) fprintf(stderr, "[pair verify] %d systems (Tmax %d): identical within 1e-9\n", n, Tmax);
    for (void* p : tmp) (void)hipFree(p);
    Tmax = Tmax > 0 ? -Tmax : 0;
  } else if (pair && chol_pair_supported(d, n)) { launch_chol_pair_steps(d, n, ctr, pair_tickets, s); Tmax = Tmax > 0 ? -Tmax : 0; }
  for (int k = 0; k < Tmax; ++k) {
    CholBatchArgs A{};
    A.n = n;
    long long nA2 = 0, nBt = 0;
    StepPlan pl[CHOL_STEP_BATCH_MAX];
    for (int i = 0; i < n; ++i) { pl[i] = plan_step(k, d[i].T, d[i].h_prof, d[i].nbr, d[i].h_bfirst, d[i].kofs); nA2 += 2 * pl[i].nA; nBt += pl[i].nB; }
    // two workgroups per type-A tile only while the launch's share of the chip holds them: `cu_share` percent of the CUs (the launch
    // sequences that run side by side divide the chip: three sequences of eight cut bands at 100 % each 0.63 ms, at 33 - 50 % 0.54)
    static const int env_share = getenv("SLIDE_CHOL_BSPLIT") ? atoi(getenv("SLIDE_CHOL_BSPLIT")) : -1;
    const int share = env_share >= 0 ? env_share : cu_share;
    const int a_split = (k > 0 && (nA2 + nBt) * 100 <= (long long)n_cu * share) ? 1 : 0;
    A.a_base[0] = A.b_base[0] = 0;
    for (int i = 0; i < n; ++i) {
      A.S[i] = d[i].S; A.ld[i] = d[i].ld; A.T[i] = d[i].T; A.Ld[i] = d[i].Ld; A.Winv[i] = d[i].Winv; A.status[i] = d[i].status;
      A.L32[i] = d[i].L32;
      A.TvA[i] = pl[i].TvA; A.TvB[i] = pl[i].TvB; A.TvX[i] = pl[i].TvX;
      A.nP[i] = pl[i].nP; A.g0[i] = pl[i].g0; A.g1[i] = pl[i].g1; A.nX[i] = pl[i].nX; A.a_split[i] = a_split;
      A.nbr[i] = d[i].nbr; A.nbA[i] = pl[i].nbA; A.nbB[i] = pl[i].nbB; A.nbX[i] = pl[i].nbX;
      A.B0[i] = d[i].b0 > 0 ? d[i].b0 : d[i].T;      // (a view with its border further down carries no f32 factor copy: L32's tile index uses T)
      A.ord[i] = d[i].ord;
      A.a_base[i + 1] = A.a_base[i] + (int)(pl[i].nA << a_split);
      A.b_base[i + 1] = A.b_base[i] + (int)pl[i].nB;
    }
    const long long nAw = A.a_base[n], nB = A.b_base[n];
    const long long extra = nB < n_cu ? nB : n_cu;
    const long long free_cu = n_cu - nAw > 8 ? n_cu - nAw : 8;
    const int a_joins = nB > free_cu ? 1 : 0;
    static const int xcd_map = getenv("SLIDE_CHOL_XCD") ? atoi(getenv("SLIDE_CHOL_XCD")) : 0;      // measured r5: 0.529 vs 0.526 ms of band factorisations (noise) - opt-in
    hipLaunchKernelGGL(k_chol_step_batched, dim3((unsigned)(nAw + extra)), dim3(512), 0, s, A, k, k & ~1, ctr, a_joins, (xcd_map && n > 1) ? 1 : 0);
  }
  if (Tmax < 0) Tmax = -Tmax;
  if (after_steps) (void)hipEventRecord(after_steps, s);
  if (!solve) {       // the caller continues with the border (k_border_syrk, the separator system) and runs launch_chol_bwd_batch itself
    ExtractArgs E{};
    E.n = n;
    for (int i = 0; i < n; ++i) {
      E.S[i] = d[i].S; E.ld[i] = d[i].ld; E.T[i] = d[i].T; E.Tr[i] = (d[i].b0 > 0 ? d[i].b0 : d[i].T) + d[i].nbr;
      E.yv[i] = d[i].yv; E.dp[i] = d[i].dp; E.status[i] = d[i].status;
    }
    if (Tmax > 0) hipLaunchKernelGGL(k_chol_extract_y_b, dim3((Tmax * NB + 255) / 256, n), dim3(256), 0, s, E);
    return;
  }
  launch_chain_tables(d, n, s);
  const double* yin[CHOL_BATCH_MAX];
  double* xout[CHOL_BATCH_MAX];
  for (int i = 0; i < n; ++i) {
    launch_chol_extract_y(d[i].S, d[i].ld, d[i].T, d[i].yv, d[i].dp, d[i].status, s, d[i].nbr);
    yin[i] = d[i].yv; xout[i] = d[i].dp;
  }
  if (chain_has_tables(d, n)) {                 // joint-solve pass: the tables are there for the preconditioner anyway
    launch_chain_batch(d, n, yin, xout, false, false, true, nullptr, s);      // (k_chol_extract_y prepared dp)
    return;
  }
  launch_chol_bwd_batch(d, n, s);
}
// the chained backward substitutions yv -> dp of up to 8 factored systems in one launch (after launch_chol_extract_y)
void launch_chol_bwd_batch(const CholSystem* d, int n, hipStream_t s) {
  if (n > BWD_BATCH_MAX) {      // (more systems than one launch takes: in chunks)
    for (int lo = 0; lo < n; lo += BWD_BATCH_MAX) launch_chol_bwd_batch(d + lo, n - lo < BWD_BATCH_MAX ? n - lo : BWD_BATCH_MAX, s);
    return;
  }
  BwdBatchArgs B{};
  B.n = n;
  B.base[0] = 0;
  int Tmax = 0;
  for (int i = 0; i < n; ++i) {
    B.S[i] = d[i].S; B.ld[i] = d[i].ld; B.T[i] = d[i].T; B.Ld[i] = d[i].Ld; B.Winv[i] = d[i].Winv; B.yv[i] = d[i].yv; B.dp[i] = d[i].dp;
    B.status[i] = d[i].status;
    B.prof[i] = d[i].prof;
    B.base[i + 1] = B.base[i] + d[i].T;
    Tmax = d[i].T > Tmax ? d[i].T : Tmax;
  }
  B.Tmax = Tmax;
  if (B.base[n] <= 0) return;
  hipLaunchKernelGGL(k_chol_bwd_chain_batched, dim3(B.base[n]), dim3(CHAIN_THREADS), 0, s, B);
}
// the solve of the factorisation's own right-hand side for one system (after launch_chol_extract_y): yv -> dp
void launch_chol_solve_bwd(const CholSystem& cs, hipStream_t s) {
  if (chain_has_tables(&cs, 1)) {
    launch_chain_tables(&cs, 1, s);
    const double* yin = cs.yv;
    double* xout = cs.dp;
    launch_chain_batch(&cs, 1, &yin, &xout, false, false, true, nullptr, s);
    return;
  }
  launch_chol_bwd_all(cs.S, cs.ld, cs.T, cs.Ld, cs.Winv, cs.yv, cs.dp, cs.status, cs.prof, s);
}
void launch_chol_bwd_all(const double* S, int ld, int T, const double* Ld, const double* Winv, double* yv, double* dp,
                         int* status, const int* prof, hipStream_t s, const double* wf_prev, double wf_thr, int wf_Tprev, int wf_lim) {
  if (wf_lim < 0 || wf_lim > wf_Tprev) wf_lim = wf_Tprev;
  if (wf_prev && wf_thr > 0.0 && wf_Tprev > 0 && wf_lim > 0)
    hipLaunchKernelGGL(k_chol_bwd_chain_wf, dim3(T), dim3(CHAIN_THREADS), 0, s, S, ld, T, Ld, Winv, yv, dp, status, prof, wf_prev, wf_thr, wf_Tprev, wf_lim);
  else
    hipLaunchKernelGGL(k_chol_bwd_chain, dim3(T), dim3(CHAIN_THREADS), 0, s, S, ld, T, Ld, Winv, yv, dp, status, prof);
}

// ---- plan and launch of the left-looking persistent factorisation (k_chol_ll) ------------------------------------------------------
// Task table of n systems, in ticket order: block column by block column over all systems (every system's chain task of column k, then
// every system's tile tasks of column k) — a topological order of the dependence graph that lets all systems advance side by side.
// Rows of column k of a system (plan_step / step_type_a): the band's tile rows k+1 .. prof[k], the border rows active at k (first
// column <= k; the j-th active row of a view is tile row B0 + ord[j]), the right-hand-side row B0 + nbr.  Pending columns of a tile
// (i, k): those c < k in which both L(i, c) and L(k, c) are structurally non-zero — from first[k] (the first column whose profile
// reaches row k) resp. first[i] / the border row's first column on.
struct CholLLPlan {
  LLSys* d_sys = nullptr; LLTask* d_tasks = nullptr; int* d_ints = nullptr;
  int n_sys = 0, n_tasks = 0, n_ints = 0, n_chain = 0, max_cols = 0;
};
CholLLPlan* chol_ll_plan_create(const CholSystem* d, int n, const int* const* h_ord, hipStream_t up) {
  // SLIDE_LL_MODE: how the rows of a column beyond the chain task's own are factored.  0: tile tasks (wait for the published L_kk, blocked
  // substitution; two tile rows per workgroup); 1: FOLLOWER chain tasks — every tile row its own workgroup that factors the diagonal
  // block redundantly beside the lead (the type-A workgroups of the step kernels): no row lags the chain by a hand-over + substitution,
  // at the price of a CU per row for the length of the chain; 2: followers for systems with a profile (bands) and for short dense ones
  // (T <= 8), tile tasks for long dense ones
  // 3: followers for the band's own tile rows, tile tasks for border rows.  Measured (profiles/r04_ll_experiments.txt): 0 is the fastest
  // on the cut bands of C4 (0.775 ms against 0.98 / 0.84 for 1 / 3) — and none beats the step kernels (0.52 ms)
  static const int ll_mode = getenv("SLIDE_LL_MODE") ? atoi(getenv("SLIDE_LL_MODE")) : 0;
  std::vector<LLSys> sys(n);
  std::vector<std::vector<int>> first(n), bf(n), bphys(n), profv(n);
  size_t n_ints = 16;                                   // [0] ticket counter, [1] abort word; flags from 16 on
  int Tmax = 0;
  for (int i = 0; i < n; ++i) {
    const CholSystem& c = d[i];
    LLSys& y = sys[i];
    y.S = c.S; y.Ld = c.Ld; y.Winv = c.Winv; y.status = c.status; y.ld = c.ld; y.T = c.T; y.B0 = c.b0 > 0 ? c.b0 : c.T; y.nbr = c.nbr;
    y.frows = y.B0 + y.nbr + 1;
    y.flags = reinterpret_cast<int*>(n_ints);           // offset for now: the pool is allocated below
    n_ints += (size_t)y.T * y.frows;
    Tmax = c.T > Tmax ? c.T : Tmax;
    profv[i].resize(c.T);
    for (int k = 0; k < c.T; ++k) profv[i][k] = c.h_prof ? std::min(c.h_prof[k], c.T - 1) : c.T - 1;
    first[i].assign(c.T, 0);
    for (int r = 0, col = 0; r < c.T; ++r) {            // first[r] = min { col : prof[col] >= r } (prof monotone, prof[col] >= col)
      while (col < r && profv[i][col] < r) ++col;
      first[i][r] = col;
    }
    // border rows in the order in which they become active: (first column in the view's numbering, physical tile row)
    for (int j = 0; j < c.nbr; ++j) {
      long long f = c.h_bfirst ? (long long)c.h_bfirst[j] - c.kofs : 0;
      if (c.h_bfirst && j > 0 && c.h_bfirst[j] < c.h_bfirst[j - 1]) f = (long long)1 << 40;      // (not sorted: plan_step's count stops here too)
      if (f >= c.T) break;                              // this row and all later ones are all-zero in this system
      bf[i].push_back(f < 0 ? 0 : (int)f);
      bphys[i].push_back(y.B0 + ((h_ord && h_ord[i]) ? h_ord[i][j] : j));
    }
  }
  std::vector<LLTask> tasks;
  int n_chain = 0;
  for (int k = 0; k < Tmax; ++k) {
    for (int pass = 0; pass < 2; ++pass)
      for (int i = 0; i < n; ++i) {
        const LLSys& y = sys[i];
        if (k >= y.T) continue;
        const int fk = first[i][k];
        std::vector<std::pair<int, int>> rows;            // (physical tile row, first pending column)
        for (int r = k + 1; r <= profv[i][k]; ++r) rows.emplace_back(r, std::max(first[i][r], fk));
        for (size_t j = 0; j < bf[i].size() && bf[i][j] <= k; ++j) rows.emplace_back(bphys[i][j], std::max(bf[i][j], fk));
        rows.emplace_back(y.B0 + y.nbr, fk);
        if (pass == 0) {
          tasks.push_back(LLTask{i, k, 0, rows[0].first, -1, fk, rows[0].second, k});
          ++n_chain;
        } else if (ll_mode == 1 || (ll_mode == 2 && (d[i].h_prof != nullptr || d[i].T <= 8))) {      // (bands, and short dense systems: few older panels per row)
          for (size_t r = 1; r < rows.size(); ++r) tasks.push_back(LLTask{i, k, 2, rows[r].first, -1, fk, rows[r].second, k});
        } else if (ll_mode == 3) {
          // followers for the BAND's tile rows (the next chain tasks' own rows come from them: no row of the diagonal chain then lags a
          // column's chain by a hand-over + substitution), tile tasks for the border rows and the right-hand side (nobody on the chain
          // waits for those)
          const size_t nband = (size_t)std::max(0, profv[i][k] - k);
          size_t r = 1;
          for (; r < rows.size() && r <= nband; ++r) tasks.push_back(LLTask{i, k, 2, rows[r].first, -1, fk, rows[r].second, k});
          for (; r < rows.size(); r += 2) {
            const bool two = r + 1 < rows.size();
            tasks.push_back(LLTask{i, k, 1, rows[r].first, two ? rows[r + 1].first : -1, fk, rows[r].second, two ? rows[r + 1].second : k});
          }
        } else {
          for (size_t r = 1; r < rows.size(); r += 2) {
            const bool two = r + 1 < rows.size();
            tasks.push_back(LLTask{i, k, 1, rows[r].first, two ? rows[r + 1].first : -1, fk, rows[r].second, two ? rows[r + 1].second : k});
          }
        }
      }
  }
  {
    // self-check of the table (host, once per plan): replay the tasks in ticket order against a host copy of the flags — every flag a
    // task waits for must have been raised by an EARLIER task (else the launch could only end by its time-out), every tile of every
    // column must be produced exactly once
    std::vector<std::vector<char>> fl(n);
    for (int i = 0; i < n; ++i) fl[i].assign((size_t)sys[i].T * sys[i].frows, 0);
    auto need = [&](const LLTask& t, int c, int row) {
      const LLSys& y = sys[t.sys];
      if (c < 0 || c >= y.T || row < 0 || row >= y.frows || !fl[t.sys][(size_t)c * y.frows + row]) {
        fprintf(stderr, "slide_slam_amd: left-looking plan: task (system %d, column %d, kind %d, rows %d %d) waits for tile (%d, %d) that no earlier task produces\n",
                t.sys, t.k, t.kind, t.it0, t.it1, row, c);
        return false;
      }
      return true;
    };
    auto raise = [&](const LLTask& t, int row) {
      const LLSys& y = sys[t.sys];
      char& f = fl[t.sys][(size_t)t.k * y.frows + row];
      if (f) { fprintf(stderr, "slide_slam_amd: left-looking plan: tile (%d, %d) of system %d is produced twice\n", row, t.k, t.sys); return false; }
      f = 1;
      return true;
    };
    bool good = true;
    for (const LLTask& t : tasks) {
      if (t.kind != 1) {
        for (int c = t.clo_d; c < t.k && good; ++c) good = need(t, c, t.k);
        for (int c = t.clo0; c < t.k && good; ++c) good = need(t, c, t.it0) && need(t, c, t.k);
        good = good && t.clo_d >= 0 && t.clo0 >= t.clo_d && (t.kind != 0 || raise(t, t.k)) && raise(t, t.it0);
      } else {
        for (int sl = 0; sl < 2 && good; ++sl) {
          const int it = sl ? t.it1 : t.it0, clo = sl ? t.clo1 : t.clo0;
          if (it < 0) continue;
          for (int c = clo; c < t.k && good; ++c) good = need(t, c, it) && need(t, c, t.k);
          good = good && clo >= t.clo_d && need(t, t.k, t.k);
        }
        good = good && raise(t, t.it0) && (t.it1 < 0 || raise(t, t.it1));
      }
      if (!good) break;
    }
    if (!good) return nullptr;
  }
  CholLLPlan* p = new CholLLPlan();
  p->n_sys = n; p->n_tasks = (int)tasks.size(); p->n_ints = (int)n_ints; p->n_chain = n_chain; p->max_cols = Tmax;
  bool ok = hipMalloc(reinterpret_cast<void**>(&p->d_ints), n_ints * sizeof(int)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&p->d_sys), std::max<size_t>(1, sys.size()) * sizeof(LLSys)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&p->d_tasks), std::max<size_t>(1, tasks.size()) * sizeof(LLTask)) == hipSuccess;
  if (ok) {
    for (LLSys& y : sys) y.flags = p->d_ints + reinterpret_cast<size_t>(y.flags);
    // (on the caller's stream, not the legacy stream: other host threads of the process may be capturing)
    ok = hipMemcpyAsync(p->d_sys, sys.data(), sys.size() * sizeof(LLSys), hipMemcpyHostToDevice, up) == hipSuccess &&
         hipMemcpyAsync(p->d_tasks, tasks.data(), tasks.size() * sizeof(LLTask), hipMemcpyHostToDevice, up) == hipSuccess &&
         hipMemsetAsync(p->d_ints, 0, n_ints * sizeof(int), up) == hipSuccess && hipStreamSynchronize(up) == hipSuccess;
  }
  if (!ok) { chol_ll_plan_destroy(p); return nullptr; }
  return p;
}
void chol_ll_plan_destroy(CholLLPlan* p) {
  if (!p) return;
  if (p->d_ints) (void)hipFree(p->d_ints);
  if (p->d_sys) (void)hipFree(p->d_sys);
  if (p->d_tasks) (void)hipFree(p->d_tasks);
  delete p;
}
int chol_ll_plan_tasks(const CholLLPlan* p) { return p ? p->n_tasks : 0; }
int chol_ll_plan_columns(const CholLLPlan* p) { return p ? p->max_cols : 0; }
__global__ void k_ll_clear(int* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}
// the factorisations of the plan's systems (the steps of launch_chol_batch(.., solve = false) in ONE launch) + the extraction of y
void launch_chol_ll(const CholLLPlan* p, const CholSystem* d, int n, hipStream_t s, int* trace) {
  if (!p || p->n_tasks <= 0) return;
  hipLaunchKernelGGL(k_ll_clear, dim3((p->n_ints + 255) / 256), dim3(256), 0, s, p->d_ints, p->n_ints);
  hipLaunchKernelGGL(k_chol_ll, dim3((unsigned)p->n_tasks), dim3(512), 0, s, p->d_sys, p->d_tasks, p->n_tasks, p->d_ints, trace);
  int Tmax = 0;
  for (int lo = 0; lo < n; lo += CHOL_STEP_BATCH_MAX) {
    ExtractArgs E{};
    E.n = std::min(n - lo, CHOL_STEP_BATCH_MAX);
    Tmax = 0;
    for (int i = 0; i < E.n; ++i) {
      const CholSystem& c = d[lo + i];
      E.S[i] = c.S; E.ld[i] = c.ld; E.T[i] = c.T; E.Tr[i] = (c.b0 > 0 ? c.b0 : c.T) + c.nbr;
      E.yv[i] = c.yv; E.dp[i] = c.dp; E.status[i] = c.status;
      Tmax = c.T > Tmax ? c.T : Tmax;
    }
    if (Tmax > 0) hipLaunchKernelGGL(k_chol_extract_y_b, dim3((Tmax * NB + 255) / 256, E.n), dim3(256), 0, s, E);
  }
}

int chol_factor_solve(double* S, int ld, int T, double* Ld, double* Winv, double* yv, double* dp, int* status, int* ctr, hipStream_t s) {
  for (int k = 0; k < T; ++k) launch_chol_step(S, ld, k, T, Ld + (size_t)k * NB * NB, Winv + (size_t)k * 1024, status, ctr, nullptr, nullptr, s);
  launch_chol_extract_y(S, ld, T, yv, dp, status, s);
  launch_chol_bwd_all(S, ld, T, Ld, Winv, yv, dp, status, nullptr, s);
  return 0;
}

}  // namespace sl
