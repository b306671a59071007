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

#include "graph_dev.hpp"
#include "kernels.hpp"

namespace sl {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ inline v4d mfma_f64(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// ------------------------------------------------------------------------------------------------
// Step kernel "diag + panel": every workgroup (one per row tile below the diagonal block, RHS tile included)
// redundantly factors the 64x64 diagonal block in REGISTERS (256 threads as a 16x16 grid, thread (ti, tj)
// owns A[ti + 16p][tj + 16q]; per column step the owners of column j publish it through a double-buffered
// LDS vector: one barrier per step, no integer division, no scratch), inverts its four 16x16 diagonal
// sub-blocks (one wave each), and then solves its own 64 rows X = A L^-T by blocked substitution on
// v_mfma_f64_16x16x4_f64.  The MFMA accumulator layout of X_c^T (row (lane>>4) + 4r, column lane&15) is
// exactly the B-operand layout of k-step r, so the chained products need no lane movement.
// Redundant factoring costs no latency (the blocks run concurrently) and removes a kernel boundary and the
// L / W round trip through L2 from the critical path.  Workgroup 0 also publishes L_kk and the 16x16
// inverses for the backward substitution.
__device__ inline double rsqrt_nr(double d) {
  // 1/sqrt(d): hardware estimate + two Newton steps (full double precision)
  double y = __builtin_amdgcn_rsq(d);
#ifndef EXP_NO_NR
  y = y * (1.5 - 0.5 * d * y * y);
  y = y * (1.5 - 0.5 * d * y * y);
#endif
  return y;
}

// Rank-4 elimination of the 64x64 diagonal tile held in registers (256 threads as a 16x16 grid, thread (ti, tj)
// owns A[ti + 16p][tj + 16q]).  Columns j..j+3 are published together through a double-buffered LDS panel
// (ONE barrier per FOUR columns); every thread redoes the 4x4 pivot Cholesky and the four-term substitution for
// its own rows / columns.  While block column JQ is eliminated only the register blocks JQ <= q <= p change, and
// the upper triangle of A is never read for a result, so two masks suffice (rows / columns <= j+3 of the pivot
// block).  The inverse of the 16x16 diagonal sub-block (needed by the triangular solves) rides along: each
// thread carries ONE element w of it and applies the same four-term substitution to it.
template <int JQ>
__device__ inline void diag_steps(double (&a)[4][4], double (&wfin)[4], double (*colraw)[4][NB], double (*wraw)[4][16], int ti,
                                  int tj, int* status) {
  double w = (ti == tj) ? 1.0 : 0.0;   // element (ti, tj) of the identity -> of L_bb^-1
#pragma unroll 1
  for (int jt = 0; jt < 16; jt += 4) {
    const int j = 16 * JQ + jt;
    const int buf = (jt >> 2) & 1;
    const int rc = tj - jt;            // 0..3 when this thread owns one of the four pivot columns
    if (rc >= 0 && rc < 4) {
#pragma unroll
      for (int p = JQ; p < 4; ++p) colraw[buf][rc][ti + 16 * p] = a[p][JQ];
    }
    const int rr = ti - jt;            // 0..3 when this thread owns one of the four pivot rows of W
    if (rr >= 0 && rr < 4) wraw[buf][rr][tj] = w;
#ifndef EXP_NO_BARRIER
    __syncthreads();
#endif
    const double* c0 = colraw[buf][0];
    const double* c1 = colraw[buf][1];
    const double* c2 = colraw[buf][2];
    const double* c3 = colraw[buf][3];
    // 4x4 pivot block (lower part)
    double P00 = c0[j], P10 = c0[j + 1], P20 = c0[j + 2], P30 = c0[j + 3];
    double P11 = c1[j + 1], P21 = c1[j + 2], P31 = c1[j + 3];
    double P22 = c2[j + 2], P32 = c2[j + 3], P33 = c3[j + 3];
    bool bad = !(P00 > 0.0);
    if (bad) P00 = 1.0;
    const double i0 = rsqrt_nr(P00);
    const double l10 = P10 * i0, l20 = P20 * i0, l30 = P30 * i0;
    double d1 = P11 - l10 * l10;
    if (!(d1 > 0.0)) { bad = true; d1 = 1.0; }
    const double i1 = rsqrt_nr(d1);
    const double l21 = (P21 - l20 * l10) * i1, l31 = (P31 - l30 * l10) * i1;
    double d2 = P22 - l20 * l20 - l21 * l21;
    if (!(d2 > 0.0)) { bad = true; d2 = 1.0; }
    const double i2 = rsqrt_nr(d2);
    const double l32 = (P32 - l30 * l20 - l31 * l21) * i2;
    double d3 = P33 - l30 * l30 - l31 * l31 - l32 * l32;
    if (!(d3 > 0.0)) { bad = true; d3 = 1.0; }
    const double i3 = rsqrt_nr(d3);
    if (bad && ti == 0 && tj == 0 && blockIdx.x == 0) atomicOr(&status[1], 1);
    // four-term forward substitution  (v0..v3) -> (x0..x3) = L_piv^-1 v
#define SUBST4(v0, v1, v2, v3, x0, x1, x2, x3)          \
    const double x0 = (v0) * i0;                          \
    const double x1 = ((v1) - x0 * l10) * i1;             \
    const double x2 = ((v2) - x0 * l20 - x1 * l21) * i2;  \
    const double x3 = ((v3) - x0 * l30 - x1 * l31 - x2 * l32) * i3;
    double X0[4], X1[4], X2[4], X3[4], Y0[4], Y1[4], Y2[4], Y3[4];
#pragma unroll
    for (int p = JQ; p < 4; ++p) {
      const int i = ti + 16 * p;
      SUBST4(c0[i], c1[i], c2[i], c3[i], x0, x1, x2, x3)
      X0[p] = x0; X1[p] = x1; X2[p] = x2; X3[p] = x3;
    }
#pragma unroll
    for (int q = JQ; q < 4; ++q) {
      const int c = tj + 16 * q;
      SUBST4(c0[c], c1[c], c2[c], c3[c], y0, y1, y2, y3)
      Y0[q] = y0; Y1[q] = y1; Y2[q] = y2; Y3[q] = y3;
    }
    const bool rlive = ti > jt + 3, clive = tj > jt + 3;
    const double x0m = rlive ? X0[JQ] : 0.0, x1m = rlive ? X1[JQ] : 0.0, x2m = rlive ? X2[JQ] : 0.0, x3m = rlive ? X3[JQ] : 0.0;
    const double y0m = clive ? Y0[JQ] : 0.0, y1m = clive ? Y1[JQ] : 0.0, y2m = clive ? Y2[JQ] : 0.0, y3m = clive ? Y3[JQ] : 0.0;
#pragma unroll
    for (int p = JQ; p < 4; ++p)
#pragma unroll
      for (int q = JQ; q <= p; ++q) {
        const double e0 = (p == JQ ? x0m : X0[p]), e1 = (p == JQ ? x1m : X1[p]), e2 = (p == JQ ? x2m : X2[p]), e3 = (p == JQ ? x3m : X3[p]);
        const double f0 = (q == JQ ? y0m : Y0[q]), f1 = (q == JQ ? y1m : Y1[q]), f2 = (q == JQ ? y2m : Y2[q]), f3 = (q == JQ ? y3m : Y3[q]);
        a[p][q] -= e0 * f0 + e1 * f1 + e2 * f2 + e3 * f3;
      }
    // inverse of the diagonal sub-block: rows j..j+3 of W become L_piv^-1 (rows), rows below take the update
    {
      const double* w0 = wraw[buf][0];
      SUBST4(w0[tj], wraw[buf][1][tj], wraw[buf][2][tj], wraw[buf][3][tj], z0, z1, z2, z3)
      if (rlive) w -= X0[JQ] * z0 + X1[JQ] * z1 + X2[JQ] * z2 + X3[JQ] * z3;
      if (rr == 0) w = z0; else if (rr == 1) w = z1; else if (rr == 2) w = z2; else if (rr == 3) w = z3;
    }
#undef SUBST4
    if (rc >= 0 && rc < 4) {
#pragma unroll
      for (int p = JQ; p < 4; ++p) a[p][JQ] = (rc == 0) ? X0[p] : (rc == 1) ? X1[p] : (rc == 2) ? X2[p] : X3[p];   // columns j..j+3 of L
    }
  }
  wfin[JQ] = w;
}

#ifdef SLIDE_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i)
#endif
constexpr int LSTR = 80;   // LDS column stride of the factored block: two adjacent columns fall in disjoint banks

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
__global__ __launch_bounds__(256) void k_chol_step(double* __restrict__ S, int ld, int k, int T, double* __restrict__ Ld,
                                                   double* __restrict__ Winv, int* status) {
  __shared__ double colraw[2][4][NB];
  __shared__ double wraw[2][4][16];
  __shared__ double Ls[NB * LSTR];        // Ls[c * LSTR + r] = L[r][c]  (first used to re-shape the updated diagonal tile)
  __shared__ double Wi[4][16 * 16];       // Wi[b][c * 16 + r] = (L_bb^-1)[r][c]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int nA = T - k;
  if ((int)blockIdx.x >= nA) {
    // ---------------- type B: trailing update of one 64x64 tile with panel k-1 ----------------
    const long long t = (long long)blockIdx.x - nA;
    long long ii = (long long)floor((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (ii * (ii + 1) / 2 > t) --ii;
    while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
    const int jj = (int)(t - ii * (ii + 1) / 2);
    const int i = k + 1 + (int)ii, j = k + 1 + jj;
    const int m0 = 32 * (wave >> 1), n0 = 32 * (wave & 1);
    const double* pj = S + (size_t)((k - 1) * NB) * ld + (size_t)j * NB + n0 + lr;
    const double* pi = S + (size_t)((k - 1) * NB) * ld + (size_t)i * NB + m0 + lr;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const size_t off = (size_t)(4 * ks + lk) * ld;
      const double a0 = pj[off], a1 = pj[off + 16];
      const double b0 = pi[off], b1 = pi[off + 16];
      acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
      acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
      acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
      acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
    }
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double* c = S + (size_t)(j * NB + n0 + 16 * pb + lk + 4 * r) * ld + (size_t)i * NB + m0 + 16 * qb + lr;
          *c -= acc[pb][qb][r];
        }
    return;
  }
  // ---------------- type A ----------------
  STAMP(0);
  const int it = k + 1 + blockIdx.x;
  const int ti = tid & 15, tj = tid >> 4;
  // own rows (16 per wave) of tile (it, k) and of the diagonal tile (k, k), both in the transposed MFMA layout
  // t[b][r] = A[row 16*wave + lr][col 16b + lk + 4r]
  double* col = S + (size_t)(k * NB) * ld + (size_t)it * NB + 16 * wave + lr;
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
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const size_t off = (size_t)(4 * ks + lk) * ld;
      const double bi = pi[off], bd = pd[off];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const double aop = -pk[off + 16 * b];
        tt[b] = mfma_f64(aop, bi, tt[b]);
        dd[b] = mfma_f64(aop, bd, dd[b]);
      }
    }
  }
  STAMP(2);
  // re-shape the (updated) diagonal tile through LDS into the factorisation layout a[p][q] = A[ti + 16p][tj + 16q]
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) Ls[(16 * b + lk + 4 * r) * LSTR + 16 * wave + lr] = dd[b][r];
  __syncthreads();
  double a[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = ti + 16 * p, c = tj + 16 * q;
      a[p][q] = (i >= c) ? Ls[c * LSTR + i] : 0.0;
    }
  __syncthreads();
  STAMP(3);
  double wfin[4];
  diag_steps<0>(a, wfin, colraw, wraw, ti, tj, status);
  diag_steps<1>(a, wfin, colraw, wraw, ti, tj, status);
  diag_steps<2>(a, wfin, colraw, wraw, ti, tj, status);
  diag_steps<3>(a, wfin, colraw, wraw, ti, tj, status);
  STAMP(4);
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = ti + 16 * p, c = tj + 16 * q;
      const double v = (i >= c) ? a[p][q] : 0.0;
      Ls[c * LSTR + i] = v;
      if (blockIdx.x == 0) Ld[(size_t)c * NB + i] = v;
    }
#pragma unroll
  for (int bq = 0; bq < 4; ++bq) {
    const double v = (ti >= tj) ? wfin[bq] : 0.0;          // (L_bb^-1)[ti][tj]
    Wi[bq][tj * 16 + ti] = v;
    if (blockIdx.x == 0) Winv[(size_t)bq * 256 + tj * 16 + ti] = v;
  }
  __syncthreads();
  STAMP(5);
  // blocked triangular solve of this workgroup's 64 rows (16 per wave), operands already in registers
  v4d xt[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    v4d t = tt[b];
#pragma unroll
    for (int c = 0; c < b; ++c)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const double aop = -Ls[(16 * c + 4 * s4 + lk) * LSTR + 16 * b + lr];       // -L[16b + n][16c + j]
        t = mfma_f64(aop, xt[c][s4], t);
      }
    v4d x = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const double aop = Wi[b][(4 * s4 + lk) * 16 + lr];                             // (L_bb^-1)[n][j]
      x = mfma_f64(aop, t[s4], x);
    }
    xt[b] = x;
#pragma unroll
    for (int r = 0; r < 4; ++r) col[(size_t)(16 * b + lk + 4 * r) * ld] = x[r];
  }
  STAMP(6);
}
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
void launch_chol_step(double* S, int ld, int k, int T, double* Ld, double* Winv, int* status, hipStream_t s) {
  const long long nA = T - k;                                   // column-k tiles below the diagonal (+ RHS tile)
  const long long nB = k > 0 ? nA * (nA + 1) / 2 - 1 : 0;       // trailing tiles that still owe the update of panel k-1
  hipLaunchKernelGGL(k_chol_step, dim3((unsigned)(nA + nB)), dim3(256), 0, s, S, ld, k, T, Ld, Winv, status);
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

int chol_factor_solve(double* S, int ld, int T, double* Ld, double* Winv, double* yv, double* dp, int* status, hipStream_t s) {
  for (int k = 0; k < T; ++k) launch_chol_step(S, ld, k, T, Ld + (size_t)k * NB * NB, Winv + (size_t)k * 1024, status, s);
  launch_chol_extract_y(S, ld, T, yv, s);
  launch_chol_bwd_all(S, ld, T, Ld, Winv, yv, dp, s);
  return 0;
}

}  // namespace sl
