// Blocked right-looking FP64 Cholesky of the dense reduced pose system on gfx950 matrix cores.
//
// This is the "dense block-diagonal Schur-complement solve" of BASELINE.json's north_star: the
// reference hands the same linear algebra to GTSAM's multifrontal Cholesky (ISAM2Params::CHOLESKY,
// backend/sloam/src/factorgraph/graph.cpp:15).  Layout: S column-major, leading dimension
// ld = (T+1)*64, lower triangle of the (T*64)^2 system plus ONE extra row tile whose first row is
// the right-hand side, so the forward substitution L y = b falls out of the panel/update steps.
// Per step k:  diag (POTRF 64x64 + explicit inverse W_k, LDS, one workgroup)
//              panel (X = A W_k^T on v_mfma_f64_16x16x4_f64, operands straight from L2)
//              update (C_ij -= L_ik L_jk^T on v_mfma_f64_16x16x4_f64, 64x64 tile per workgroup)
// MFMA operand orientation is chosen so that every global access is a 128-byte run down a column.
// f64 MFMA lane maps (cdna_hip_programming.md §3): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// D[row = (lane>>4) + 4*reg][col = lane&15].
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "graph_dev.hpp"
#include "kernels.hpp"

namespace sl {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ inline v4d mfma_f64(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// ------------------------------------------------------------------------------------------------
// Factorisation of the 64x64 diagonal tile ON THE MATRIX CORES, one complete copy per wave (no barrier, no LDS
// inside the loop).  The tile lives in registers as its ten lower 16x16 sub-tiles in the MFMA accumulator layout
// of the TRANSPOSE ("T-layout"): lane (lr = lane&15, lk = lane>>4), register r of sub-tile (I, J) holds
// A[16I + lr][16J + lk + 4r].  In that layout
//   * register s of a sub-tile IS the B operand "rows lr, four columns 4s..4s+3", so the rank-4 panel step
//     X = A[:, j..j+3] L_p^-T is ONE v_mfma_f64_16x16x4_f64 per 16 rows (A operand = the 4x4 inverse of the pivot
//     Cholesky factor, zero-padded to 16x4) and its result register 0 is again "rows lr, four columns";
//   * the rank-4 trailing update of a sub-tile is ONE MFMA with those X registers as A and B operands.
// Only the 4x4 pivot Cholesky + inverse (a chain of four rsqrt) runs on the vector ALU, on wave-uniform values
// fetched with v_readlane.  Left-looking across the four 16-column phases: sub-tiles right of the current phase
// receive one rank-16 update (four MFMAs) at the end of the phase.  A pseudo row tile that starts as the identity
// rides along in each phase and ends as L_JJ^-T (the 16x16 inverses the triangular solves use).
__device__ inline double rsqrt_nr(double d) {
  // 1/sqrt(d): hardware estimate + two Newton steps (full double precision)
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - 0.5 * d * y * y);
  y = y * (1.5 - 0.5 * d * y * y);
  return y;
}

__device__ __forceinline__ double bcast_lane(double v, int src) {   // src: compile-time lane
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ constexpr int tidx(int I, int J) { return I * (I + 1) / 2 + J; }

// Lt: ten lower sub-tiles (T-layout).  The inverse of diagonal sub-tile JQ leaves for LDS (Wi[JQ][c * 16 + r] =
// (L_JQ,JQ^-1)[r][c], the A-operand image of the triangular solves) and, when Winv != nullptr, for global memory at
// the end of its phase.  Returns true when a pivot was not positive (fmin also propagates the NaNs it breeds).
__device__ __forceinline__ bool factor64_mfma(v4d (&Lt)[10], double (*Wi)[16 * 16], double* __restrict__ Winv, int lr, int lk) {
  double dmin = 1.0;
  const v4d zero = v4d{0.0, 0.0, 0.0, 0.0};
  // lane predicates of the 4x4 inverse inside the 16x4 A operand: row lr < 4, column lk <= lr
  const bool s00 = lr == 0 && lk == 0, s10 = lr == 1 && lk == 0, s11 = lr == 1 && lk == 1, s20 = lr == 2 && lk == 0,
             s21 = lr == 2 && lk == 1, s22 = lr == 2 && lk == 2, s30 = lr == 3 && lk == 0, s31 = lr == 3 && lk == 1,
             s32 = lr == 3 && lk == 2, s33 = lr == 3 && lk == 3;
#pragma unroll
  for (int JQ = 0; JQ < 4; ++JQ) {
    v4d W;
#pragma unroll
    for (int r = 0; r < 4; ++r) W[r] = (lr == lk + 4 * r) ? 1.0 : 0.0;
    const int dq = tidx(JQ, JQ);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double dv = Lt[dq][s];   // columns 4s..4s+3 of the diagonal sub-tile; element (4s+a, 4s+b) sits in lane 4s+a+16b
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
      dmin = fmin(fmin(dmin, fmin(P00, d1)), fmin(d2, d3));
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
      // panel: X_I = A_I[:, 4s..4s+3] L_p^-T  (register 0 of the product = X_I[row lr][column lk])
      double x[4];
#pragma unroll
      for (int I = JQ; I < 4; ++I) x[I] = mfma_f64(mop, Lt[tidx(I, JQ)][s], zero)[0];
      const double xw = mfma_f64(mop, W[s], zero)[0];
#pragma unroll
      for (int I = JQ; I < 4; ++I) Lt[tidx(I, JQ)][s] = x[I];
      W[s] = xw;
      if (s < 3) {
        // rank-4 update of the columns right of the pivot block inside this 16-column phase
        const double xm = (lr > 4 * s + 3) ? x[JQ] : 0.0;
        Lt[dq] = mfma_f64(-xm, xm, Lt[dq]);
#pragma unroll
        for (int I = JQ + 1; I < 4; ++I) Lt[tidx(I, JQ)] = mfma_f64(-xm, x[I], Lt[tidx(I, JQ)]);
        W = mfma_f64(-xm, xw, W);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double v = (lk + 4 * r >= lr) ? W[r] : 0.0;          // W lane (lr, lk) reg r = (L_JQ,JQ^-1)[lk + 4r][lr]
      Wi[JQ][lr * 16 + lk + 4 * r] = v;
      if (Winv) Winv[(size_t)JQ * 256 + lr * 16 + lk + 4 * r] = v;
    }
    // rank-16 update of the sub-tiles right of this phase (diagonal ones first: they head the next chain)
#pragma unroll
    for (int J = JQ + 1; J < 4; ++J)
#pragma unroll
      for (int I = J; I < 4; ++I)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          Lt[tidx(I, J)] = mfma_f64(-Lt[tidx(J, JQ)][ks], Lt[tidx(I, JQ)][ks], Lt[tidx(I, J)]);
  }
  return !(dmin > 0.0);
}

#ifdef SLIDE_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i)
#endif

// ------------------------------------------------------------------------------------------------
// ONE kernel per block column k ("step kernel").  Two kinds of workgroups share the launch:
//  type A (blockIdx < nA = T - k, dispatched first): tile (i, k), i = k+1..T (T = the RHS tile).  It applies
//     the trailing update of panel k-1 to its own tile AND (redundantly) to the diagonal tile (k, k), factors
//     the diagonal tile in registers, inverts its four 16x16 diagonal sub-blocks and solves its own 64 rows
//     X = A L^-T by blocked substitution on v_mfma_f64_16x16x4_f64 — i.e. the whole "look-ahead" chain of
//     column k without any inter-workgroup dependency.  Redundant factoring costs no latency (the blocks run
//     concurrently) and keeps L_kk / its inverses out of L2 round trips.
//  type B: tile (i, j), j >= k+1: plain trailing update with panel k-1 (C_ij -= L_i,k-1 L_j,k-1^T).
// The long type-A blocks therefore run BESIDE the memory-bound type-B flood inside one launch; the next step
// needs only the kernel boundary.  MFMA accumulator layout of X_c^T (row (lane>>4) + 4r, column lane&15) is
// exactly the B-operand layout of k-step r, so chained products need no lane movement.
// ---------------- type B: trailing update with panel k-1, one 2x2 group of 64x64 tiles per workgroup ----------------
// Wave (wm, wn) owns tile (i, j) = (i0 + wm, j0 + wn) entirely: sixteen 16x16 accumulators initialised with C itself,
// sixteen k-steps of eight operand loads feeding sixteen MFMAs (the loads run two k-steps ahead of their use), so the
// wave is paced by the matrix pipe even at one or two waves per SIMD — the occupancy the register-heavy type A
// leaves to the launch.  No LDS, no barrier: waves whose tile lies outside the lower triangle leave at once.
__device__ __forceinline__ long long tri_row(long long t) {
  long long ii = (long long)floor((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (ii * (ii + 1) / 2 > t) --ii;
  while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
  return ii;
}
__device__ __forceinline__ void step_type_b(double* __restrict__ S, int ld, int k, int T, long long t) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const long long bi = tri_row(t);
  const int bj = (int)(t - bi * (bi + 1) / 2);
  const int i = k + 1 + 2 * (int)bi + (wave >> 1), j = k + 1 + 2 * bj + (wave & 1);
  if (i > T || j > T - 1 || i < j) return;
  const double* pj = S + (size_t)((k - 1) * NB) * ld + (size_t)j * NB + lr;   // + 16a : rows 16a + lr of panel tile (j, k-1)
  const double* pi = S + (size_t)((k - 1) * NB) * ld + (size_t)i * NB + lr;   // + 16b : rows 16b + lr of panel tile (i, k-1)
  double* cb = S + (size_t)(j * NB + lk) * ld + (size_t)i * NB + lr;          // + (16a + 4r) ld + 16b
  // two passes of 32 columns (a = 2h, 2h+1) x 64 rows: eight accumulators live at a time
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    const double* pjh = pj + 32 * h;
    double* cbh = cb + (size_t)(32 * h) * ld;
    v4d acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] = cbh[(size_t)(16 * a + 4 * r) * ld + 16 * b];
    double pa[3][2], pb[3][4];
#pragma unroll
    for (int pre = 0; pre < 2; ++pre) {
      const size_t off = (size_t)(4 * pre + lk) * ld;
      pa[pre][0] = pjh[off]; pa[pre][1] = pjh[off + 16];
#pragma unroll
      for (int b = 0; b < 4; ++b) pb[pre][b] = pi[off + 16 * b];
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 2 < 16) {
        const size_t off = (size_t)(4 * (ks + 2) + lk) * ld;
        pa[(ks + 2) % 3][0] = pjh[off]; pa[(ks + 2) % 3][1] = pjh[off + 16];
#pragma unroll
        for (int b = 0; b < 4; ++b) pb[(ks + 2) % 3][b] = pi[off + 16 * b];
      }
      __builtin_amdgcn_sched_barrier(0);      // keep the two-k-step prefetch distance: no further hoisting of loads
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const double na = -pa[ks % 3][a];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma_f64(na, pb[ks % 3][b], acc[a][b]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) cbh[(size_t)(16 * a + 4 * r) * ld + 16 * b] = acc[a][b][r];
  }
}

// ---------------- type A: look-ahead chain of block column k for row tile k + 1 + ia ----------------
__device__ __forceinline__ void step_type_a(double* __restrict__ S, int ld, int k, int ia, double* __restrict__ Ld,
                                            double* __restrict__ Winv, int* status, double (*Tx)[4][64], double (*Wi)[16 * 16]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  STAMP(0);
  const int it = k + 1 + ia;
  const int ti = tid & 15, tj = tid >> 4;
  // own rows (16 per wave) of tile (it, k) and of the diagonal tile (k, k), both in the transposed MFMA layout
  // t[b][r] = A[row 16*wave + lr][col 16b + lk + 4r]
  const double* col = S + (size_t)(k * NB) * ld + (size_t)it * NB + 16 * wave + lr;
  const double* dcol = S + (size_t)(k * NB) * ld + (size_t)k * NB + 16 * wave + lr;
  v4d tt[4], dd[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      tt[b][r] = col[(size_t)(16 * b + lk + 4 * r) * ld];
      dd[b][r] = dcol[(size_t)(16 * b + lk + 4 * r) * ld];
    }
  STAMP(1);
  if (k > 0) {
    // pending update from panel k-1:  A[m][n] -= sum_kk L[m][kk] * L_k[n][kk]   (L_k = rows of tile row k)
    const double* pk = S + (size_t)((k - 1) * NB) * ld + (size_t)k * NB + lr;             // + 16 b : L_k[16b + lr][.]
    const double* pi = S + (size_t)((k - 1) * NB) * ld + (size_t)it * NB + 16 * wave + lr;   // own rows
    const double* pd = S + (size_t)((k - 1) * NB) * ld + (size_t)k * NB + 16 * wave + lr;    // diagonal tile's rows
    double pa[3][4], pb[3][2];               // operands run two k-steps ahead of their MFMAs
#pragma unroll
    for (int pre = 0; pre < 2; ++pre) {
      const size_t off = (size_t)(4 * pre + lk) * ld;
#pragma unroll
      for (int b = 0; b < 4; ++b) pa[pre][b] = pk[off + 16 * b];
      pb[pre][0] = pi[off]; pb[pre][1] = pd[off];
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 2 < 16) {
        const size_t off = (size_t)(4 * (ks + 2) + lk) * ld;
#pragma unroll
        for (int b = 0; b < 4; ++b) pa[(ks + 2) % 3][b] = pk[off + 16 * b];
        pb[(ks + 2) % 3][0] = pi[off]; pb[(ks + 2) % 3][1] = pd[off];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const double aop = -pa[ks % 3][b];
        tt[b] = mfma_f64(aop, pb[ks % 3][0], tt[b]);
        dd[b] = mfma_f64(aop, pb[ks % 3][1], dd[b]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  STAMP(2);
  // every wave takes a complete copy of the updated diagonal tile: wave w holds sub-tile row w (dd[b] = (w, b))
#pragma unroll
  for (int b = 0; b < 4; ++b)
    if (b <= wave) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Tx[wave * (wave + 1) / 2 + b][r][lane] = dd[b][r];
    }
  __syncthreads();
  v4d Lt[10];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) Lt[t][r] = Tx[t][r][lane];
#pragma unroll
  for (int J = 0; J < 4; ++J)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (lr < lk + 4 * r) Lt[tidx(J, J)][r] = 0.0;          // upper triangle: never read for a result, keep it finite
  STAMP(3);
  const bool bad = factor64_mfma(Lt, Wi + 4 * wave, (ia == 0 && wave == 0) ? Winv : nullptr, lr, lk);   // per-wave copies of the inverses
  if (bad && tid == 0 && ia == 0) atomicOr(&status[1], 1);
  STAMP(4);
  if (ia == 0) {
    // L_kk for the backward substitution: wave w writes sub-tile row w (every wave holds the whole factor)
#pragma unroll
    for (int I = 0; I < 4; ++I)
      if (I == wave) {
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * I + lr, c = 16 * J + lk + 4 * r;
            double v = 0.0;
            if (J <= I) v = (i >= c) ? Lt[tidx(I, J <= I ? J : 0)][r] : 0.0;
            Ld[(size_t)c * NB + i] = v;
          }
      }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  STAMP(5);
  // blocked triangular solve of this workgroup's 64 rows (16 per wave); L operands straight from registers.
  // The store addresses are rebuilt from a laundered copy of ld: sixteen 64-bit addresses kept alive since the loads
  // at the top would otherwise sit in registers (or spill) through the whole factorisation.
  int ld2 = ld;
  asm volatile("" : "+s"(ld2));
  double* col2 = S + (size_t)(k * NB) * ld2 + (size_t)it * NB + 16 * wave + lr;
  v4d xt[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    v4d t = tt[b];
#pragma unroll
    for (int c = 0; c < b; ++c)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) t = mfma_f64(-Lt[tidx(b, c)][s4], xt[c][s4], t);   // -L[16b + lr][16c + lk + 4 s4]
    v4d x = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const double aop = Wi[4 * wave + b][(4 * s4 + lk) * 16 + lr];                       // (L_bb^-1)[lr][4 s4 + lk]
      x = mfma_f64(aop, t[s4], x);
    }
    xt[b] = x;
#pragma unroll
    for (int r = 0; r < 4; ++r) col2[(size_t)(16 * b + lk + 4 * r) * ld2] = x[r];
  }
  STAMP(6);
}

// Work queue of the type-B groups: every workgroup of the launch (the type-A ones after their chain, too) draws group
// indices from ctr[k] until they run out, so the ~T-k long type-A chains and the flood balance by themselves at the one
// workgroup per CU the register-heavy kernel gets.  ctr[k+1] is cleared here for the next launch (ctr[0..1] start at 0).
__global__ __launch_bounds__(256) void k_chol_step(double* __restrict__ S, int ld, int k, int T, double* __restrict__ Ld,
                                                   double* __restrict__ Winv, int* status, int* __restrict__ ctr, int nGroups) {
  __shared__ double Tx[10][4][64];        // exchange of the updated diagonal tile: sub-tile, register, lane
  __shared__ double Wi[16][16 * 16];      // per wave w: Wi[4w + b][c * 16 + r] = (L_bb^-1)[r][c]
  __shared__ int s_g;
  const int nA = T - k;
  if (blockIdx.x == 0 && threadIdx.x == 0) ctr[k + 1] = 0;
  if ((int)blockIdx.x < nA) step_type_a(S, ld, k, (int)blockIdx.x, Ld, Winv, status, Tx, Wi);
  if (nGroups <= 0) return;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_g = atomicAdd(&ctr[k], 1);
    __syncthreads();
    const int g = s_g;
    if (g >= nGroups) break;
    step_type_b(S, ld, k, T, (long long)g);
  }
}
// diagnostic split (SLIDE_CHOL_SPLIT=1): the two kinds of workgroups as separate launches
__global__ __launch_bounds__(256) void k_chol_a(double* __restrict__ S, int ld, int k, int T, double* __restrict__ Ld,
                                                double* __restrict__ Winv, int* status) {
  __shared__ double Tx[10][4][64];
  __shared__ double Wi[16][16 * 16];
  step_type_a(S, ld, k, (int)blockIdx.x, Ld, Winv, status, Tx, Wi);
}
__global__ __launch_bounds__(256) void k_chol_b(double* __restrict__ S, int ld, int k, int T) { step_type_b(S, ld, k, T, (long long)blockIdx.x); }

#ifdef SLIDE_STAMPS
extern "C" void slide_debug_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)); }
#endif

__global__ void k_chol_extract_y(const double* __restrict__ S, int ld, int T, double* __restrict__ yv) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < T * NB) yv[c] = S[(size_t)c * ld + (size_t)T * NB];
}

// Backward substitution, BWD_GROUP block steps per launch (blocks kTop, kTop-1, ... in descending order).
// Every workgroup redundantly solves the small dense chunk of the group (d_kb = L_kb,kb^-T (y_kb - couplings inside
// the group), with the 16x16 inverses of the factorisation) and then applies the group's L tiles to its own 64
// columns:  y_c -= L[kb-block, c]^T d_kb.  EVERYTHING the chain needs (coupling tiles, own tiles, the off-diagonal
// 16x16 blocks of each L_kk and the 16x16 inverses) is fetched up front into registers in one wave of loads, so
// the dependent phases run from registers / LDS only; tiles are reduced over rows through an LDS transpose.
constexpr int BWD_GROUP = 3;

__device__ inline double bwd_tile_dot(double (*tile)[NB + 1], const double* d, int tid) {
  const int c = tid >> 2, part = tid & 3;
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += tile[c][16 * part + r] * d[16 * part + r];
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  return s;   // valid where part == 0
}

__global__ __launch_bounds__(256) void k_chol_bwd(const double* __restrict__ S, int ld, int kTop, int nsteps,
                                                  const double* __restrict__ Ld, const double* __restrict__ Winv,
                                                  double* __restrict__ yv, double* __restrict__ dp) {
  constexpr int G = BWD_GROUP;
  __shared__ double yk[G][NB];
  __shared__ double dk[G][NB];
  __shared__ double tile[NB][NB + 1];
  __shared__ double Lo[G][6][256];   // off-diagonal 16x16 blocks (b > c) of L_kb,kb : Lo[s][b*(b-1)/2 + c][col*16 + row]
  __shared__ double Ws[G][4][256];   // 16x16 inverses
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = blockIdx.x;
  const bool has_cols = cb < kTop - nsteps + 1;
  // ---- one wave of loads -------------------------------------------------------------------------------
  double tc[G][G][16];   // coupling tiles  (rows block kTop - sp, columns block kTop - sI), sp < sI
  double to[G][16];      // own tiles       (rows block kTop - sI, columns block cb)
#pragma unroll
  for (int sI = 0; sI < G; ++sI) {
#pragma unroll
    for (int sp = 0; sp < G; ++sp) {
      if (sp < sI) {
#pragma unroll
        for (int m = 0; m < 16; ++m)
          tc[sp][sI][m] = (sI < nsteps) ? S[(size_t)((kTop - sI) * NB + wave + 4 * m) * ld + (size_t)(kTop - sp) * NB + lane] : 0.0;
      }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m)
      to[sI][m] = (has_cols && sI < nsteps) ? S[(size_t)(cb * NB + wave + 4 * m) * ld + (size_t)(kTop - sI) * NB + lane] : 0.0;
  }
#pragma unroll
  for (int sI = 0; sI < G; ++sI) {
    if (sI < nsteps) {
      const int kb = kTop - sI;
      const double* Ldk = Ld + (size_t)kb * NB * NB;
      const double* Wk = Winv + (size_t)kb * 1024;
      // thread tid -> (col = tid >> 4, row = tid & 15) of each 16x16 block
#pragma unroll
      for (int b = 1; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < b; ++c)
          Lo[sI][b * (b - 1) / 2 + c][tid] = Ldk[(size_t)(16 * c + (tid >> 4)) * NB + 16 * b + (tid & 15)];
#pragma unroll
      for (int b = 0; b < 4; ++b) Ws[sI][b][tid] = Wk[(size_t)b * 256 + tid];
    }
  }
  for (int e = tid; e < nsteps * NB; e += 256) yk[e / NB][e % NB] = yv[(kTop - e / NB) * NB + e % NB];
  __syncthreads();
  // ---- dependent chain ---------------------------------------------------------------------------------
#pragma unroll
  for (int sI = 0; sI < G; ++sI) {
    if (sI < nsteps) {
      const int kb = kTop - sI;
#pragma unroll
      for (int sp = 0; sp < G; ++sp) {
        if (sp < sI) {   // y_kb -= L[kTop - sp, kb]^T d_{kTop - sp}
#pragma unroll
          for (int m = 0; m < 16; ++m) tile[wave + 4 * m][lane] = tc[sp][sI][m];
          __syncthreads();
          const double v = bwd_tile_dot(tile, dk[sp], tid);
          if ((tid & 3) == 0) yk[sI][tid >> 2] -= v;
          __syncthreads();
        }
      }
      if (wave == 0) {
        // x_b = W_b^T (y_b - sum_{c > b} L_cb^T x_c),  b = 3..0, inside the 64x64 diagonal block kb (LDS only)
#pragma unroll
        for (int b = 3; b >= 0; --b) {
          if (lane < 16) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) s += Ws[sI][b][lane * 16 + j] * yk[sI][16 * b + j];   // (W_b)[j][lane]
            dk[sI][16 * b + lane] = s;
          }
          __builtin_amdgcn_wave_barrier();
          if (b > 0 && lane < 16 * b) {
            // y[m] -= sum_n L[16b + n][m] x_n ,  m = lane in block c = lane >> 4
            const int c = lane >> 4, cc = lane & 15;
            double s = 0.0;
#pragma unroll
            for (int n = 0; n < 16; ++n) s += Lo[sI][b * (b - 1) / 2 + c][cc * 16 + n] * dk[sI][16 * b + n];
            yk[sI][lane] -= s;
          }
          __builtin_amdgcn_wave_barrier();
        }
        if (blockIdx.x == 0) dp[kb * NB + lane] = dk[sI][lane];
      }
      __syncthreads();
    }
  }
  // ---- own 64 columns (all of them lie left of the whole group) ------------------------------------------
  if (has_cols) {
    double acc = 0.0;
#pragma unroll
    for (int sI = 0; sI < G; ++sI) {
      if (sI < nsteps) {
#pragma unroll
        for (int m = 0; m < 16; ++m) tile[wave + 4 * m][lane] = to[sI][m];
        __syncthreads();
        acc += bwd_tile_dot(tile, dk[sI], tid);
        __syncthreads();
      }
    }
    if ((tid & 3) == 0) yv[cb * NB + (tid >> 2)] -= acc;
  }
}

// ------------------------------------------------------------------------------------------------
void launch_chol_step(double* S, int ld, int k, int T, double* Ld, double* Winv, int* status, int* ctr, hipStream_t s) {
  const long long nA = T - k;                                   // column-k tiles below the diagonal (+ RHS tile)
  const long long nP = (nA + 1) / 2;                            // 2x2 tile groups per side of the trailing matrix
  const long long nB = k > 0 ? nP * (nP + 1) / 2 : 0;           // groups that still owe the update of panel k-1
  static const int split = getenv("SLIDE_CHOL_SPLIT") ? atoi(getenv("SLIDE_CHOL_SPLIT")) : 0;
  if (split == 1) {
    hipLaunchKernelGGL(k_chol_a, dim3((unsigned)nA), dim3(256), 0, s, S, ld, k, T, Ld, Winv, status);
    if (nB > 0) hipLaunchKernelGGL(k_chol_b, dim3((unsigned)nB), dim3(256), 0, s, S, ld, k, T);
    return;
  }
  if (split == 2) {   // the two kinds as concurrent launches on two streams (fork / join per step)
    static hipStream_t s2 = nullptr;
    static hipEvent_t ef = nullptr, ej = nullptr;
    if (!s2) { (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); (void)hipEventCreateWithFlags(&ef, hipEventDisableTiming); (void)hipEventCreateWithFlags(&ej, hipEventDisableTiming); }
    if (nB > 0) {
      (void)hipEventRecord(ef, s);
      (void)hipStreamWaitEvent(s2, ef, 0);
    }
    hipLaunchKernelGGL(k_chol_a, dim3((unsigned)nA), dim3(256), 0, s, S, ld, k, T, Ld, Winv, status);
    if (nB > 0) {
      hipLaunchKernelGGL(k_chol_b, dim3((unsigned)nB), dim3(256), 0, s2, S, ld, k, T);
      (void)hipEventRecord(ej, s2);
      (void)hipStreamWaitEvent(s, ej, 0);
    }
    return;
  }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n_cu = pr.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  const long long extra = nB < n_cu ? nB : n_cu;                // queue workers beside the type-A workgroups
  hipLaunchKernelGGL(k_chol_step, dim3((unsigned)(nA + extra)), dim3(256), 0, s, S, ld, k, T, Ld, Winv, status, ctr, (int)nB);
}
void launch_chol_extract_y(const double* S, int ld, int T, double* yv, hipStream_t s) {
  hipLaunchKernelGGL(k_chol_extract_y, dim3((T * NB + 255) / 256), dim3(256), 0, s, S, ld, T, yv);
}
void launch_chol_bwd_all(const double* S, int ld, int T, const double* Ld, const double* Winv, double* yv, double* dp,
                         hipStream_t s) {
  for (int kTop = T - 1; kTop >= 0; kTop -= BWD_GROUP) {
    const int nsteps = kTop + 1 < BWD_GROUP ? kTop + 1 : BWD_GROUP;
    const int ncol = kTop - nsteps + 1;          // column blocks left of the group
    hipLaunchKernelGGL(k_chol_bwd, dim3(ncol > 0 ? ncol : 1), dim3(256), 0, s, S, ld, kTop, nsteps, Ld, Winv, yv, dp);
  }
}

int chol_factor_solve(double* S, int ld, int T, double* Ld, double* Winv, double* yv, double* dp, int* status, int* ctr, hipStream_t s) {
  for (int k = 0; k < T; ++k) launch_chol_step(S, ld, k, T, Ld + (size_t)k * NB * NB, Winv + (size_t)k * 1024, status, ctr, s);
  launch_chol_extract_y(S, ld, T, yv, s);
  launch_chol_bwd_all(S, ld, T, Ld, Winv, yv, dp, s);
  return 0;
}

}  // namespace sl
