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
template <int JQ>
__device__ inline void diag_steps(double (&a)[4][4], double (*colraw)[NB], int ti, int tj, int* status) {
  // Only the 16x16 register blocks with JQ <= q <= p can change while columns 16JQ..16JQ+15 are eliminated
  // (blocks above / left of it are finished), and the upper triangle of A is never read for a result, so
  // the update needs just two masks, both inside the pivot block: rows <= j and columns <= j are skipped.
#pragma unroll 1
  for (int jt = 0; jt < 16; ++jt) {
    const int j = 16 * JQ + jt;
    const int buf = j & 1;
    if (tj == jt) {
#pragma unroll
      for (int p = JQ; p < 4; ++p) colraw[buf][ti + 16 * p] = a[p][JQ];
    }
    __syncthreads();
    double d = colraw[buf][j];
    if (!(d > 0.0)) {
      if (ti == 0 && tj == 0 && blockIdx.x == 0) atomicOr(&status[1], 1);
      d = 1.0;
    }
    // 1/sqrt(d): hardware estimate + two Newton steps (full double precision)
    double inv = __builtin_amdgcn_rsq(d);
    inv = inv * (1.5 - 0.5 * d * inv * inv);
    inv = inv * (1.5 - 0.5 * d * inv * inv);
    double li[4], lk[4];
#pragma unroll
    for (int p = JQ; p < 4; ++p) li[p] = colraw[buf][ti + 16 * p] * inv;
#pragma unroll
    for (int q = JQ; q < 4; ++q) lk[q] = colraw[buf][tj + 16 * q] * inv;
    const double li0 = (ti > jt) ? li[JQ] : 0.0;      // rows <= j of the pivot block take no update
    const double lk0 = (tj > jt) ? lk[JQ] : 0.0;      // columns <= j are finished
#pragma unroll
    for (int p = JQ; p < 4; ++p)
#pragma unroll
      for (int q = JQ; q <= p; ++q) a[p][q] -= (p == JQ ? li0 : li[p]) * (q == JQ ? lk0 : lk[q]);
    if (tj == jt) {
#pragma unroll
      for (int p = JQ; p < 4; ++p) a[p][JQ] = li[p];   // column j of L (rows >= j; rows above stay unread)
    }
  }
}

constexpr int LSTR = 80;   // LDS column stride of the factored block: two adjacent columns fall in disjoint banks

__global__ __launch_bounds__(256) void k_chol_dp(double* __restrict__ S, int ld, int k, double* __restrict__ Ld,
                                                 double* __restrict__ Winv, int* status) {
  __shared__ double colraw[2][NB];
  __shared__ double Ls[NB * LSTR];        // Ls[c * LSTR + r] = L[r][c]
  __shared__ double Wi[4][16 * 16];       // Wi[b][c * 16 + r] = (L_bb^-1)[r][c]
  const int tid = threadIdx.x, ti = tid & 15, tj = tid >> 4;
  const double* dg = S + (size_t)(k * NB) * ld + (size_t)k * NB;
  double a[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = ti + 16 * p, c = tj + 16 * q;
      a[p][q] = (i >= c) ? dg[(size_t)c * ld + i] : 0.0;
    }
  diag_steps<0>(a, colraw, ti, tj, status);
  diag_steps<1>(a, colraw, ti, tj, status);
  diag_steps<2>(a, colraw, ti, tj, status);
  diag_steps<3>(a, colraw, ti, tj, status);
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = ti + 16 * p, c = tj + 16 * q;
      const double v = (i >= c) ? a[p][q] : 0.0;
      Ls[c * LSTR + i] = v;
      if (blockIdx.x == 0) Ld[(size_t)c * NB + i] = v;
    }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  {
    // wave b inverts the 16x16 lower-triangular block L_bb: lane c solves L x = e_c (forward substitution,
    // fully unrolled so x[] stays in registers)
    const int bb = wave;
    if (lane < 16) {
      const int c = lane;
      double x[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int m = 0; m < i; ++m) s -= Ls[(16 * bb + m) * LSTR + 16 * bb + i] * x[m];
        x[i] = (i >= c) ? s / Ls[(16 * bb + i) * LSTR + 16 * bb + i] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        Wi[bb][c * 16 + i] = x[i];
        if (blockIdx.x == 0) Winv[(size_t)bb * 256 + c * 16 + i] = x[i];
      }
    }
  }
  __syncthreads();
  // blocked triangular solve of this workgroup's 64 rows (16 per wave)
  const int it = k + 1 + blockIdx.x;
  const int lr = lane & 15, lk = lane >> 4;
  double* col = S + (size_t)(k * NB) * ld + (size_t)it * NB + 16 * wave + lr;
  v4d xt[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    v4d t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = col[(size_t)(16 * b + lk + 4 * r) * ld];      // Tmp^T[n = lk + 4r][m = lr]
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
}

// C_ij -= L_ik L_jk^T over trailing tiles: rows i = i0 + blockIdx.x, columns j = j0 + blockIdx.y (i >= j;
// blocks above the diagonal exit at once).  i == T is the RHS tile.
__global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ S, int ld, int k, int i0, int j0) {
  const int i = i0 + blockIdx.x, j = j0 + blockIdx.y;
  if (i < j) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = 32 * (wave >> 1), n0 = 32 * (wave & 1);
  const int lr = lane & 15, lk = lane >> 4;
  const double* pj = S + (size_t)(k * NB) * ld + (size_t)j * NB + n0 + lr;
  const double* pi = S + (size_t)(k * NB) * ld + (size_t)i * NB + m0 + lr;
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
  // D[p][q]: C[m = m0 + 16 qb + q][n = n0 + 16 pb + p], p = lk + 4 r, q = lr
#pragma unroll
  for (int pb = 0; pb < 2; ++pb)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* c = S + (size_t)(j * NB + n0 + 16 * pb + lk + 4 * r) * ld + (size_t)i * NB + m0 + 16 * qb + lr;
        *c -= acc[pb][qb][r];
      }
}

__global__ void k_chol_extract_y(const double* __restrict__ S, int ld, int T, double* __restrict__ yv) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < T * NB) yv[c] = S[(size_t)c * ld + (size_t)T * NB];
}

// backward substitution step k:  d_k = L_kk^-T y_k (blocked, with the 16x16 inverses) ;
// y_c -= L[k-block, c]^T d_k for every column c < k*NB.  One workgroup per 64 columns: the 64x64 tile of L is read
// in whole 512-byte column runs and reduced over rows through an LDS transpose, 4 threads per column.
__global__ __launch_bounds__(256) void k_chol_bwd(const double* __restrict__ S, int ld, int k, const double* __restrict__ Ld,
                                                  const double* __restrict__ Winv, double* __restrict__ yv,
                                                  double* __restrict__ dp) {
  __shared__ double yk[NB];
  __shared__ double dk[NB];
  __shared__ double tile[NB][NB + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * NB;
  const bool has_cols = c0 < k * NB;
  if (has_cols) {
    for (int cc = wave; cc < NB; cc += 4) tile[cc][lane] = S[(size_t)(c0 + cc) * ld + (size_t)k * NB + lane];
  }
  if (tid < NB) yk[tid] = yv[k * NB + tid];
  __syncthreads();
  if (wave == 0) {
    // x_b = W_b^T (y_b - sum_{c > b} L_cb^T x_c),  b = 3..0 ;  one wave, LDS traffic only within the wave
#pragma unroll 1
    for (int b = 3; b >= 0; --b) {
      if (lane < 16) {
        double s = 0.0;
        const double* w = Winv + (size_t)b * 256 + lane * 16;     // column `lane` of W_b: (W_b)[j][lane]
#pragma unroll
        for (int j = 0; j < 16; ++j) s += w[j] * yk[16 * b + j];
        dk[16 * b + lane] = s;
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < 16 * b) {
        // y[m] -= sum_n L[16b + n][m] x_n   (column m of L_kk, rows 16b..16b+15 contiguous)
        const double* lc = Ld + (size_t)lane * NB + 16 * b;
        double s = 0.0;
#pragma unroll
        for (int n = 0; n < 16; ++n) s += lc[n] * dk[16 * b + n];
        yk[lane] -= s;
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (blockIdx.x == 0) dp[k * NB + lane] = dk[lane];
  }
  __syncthreads();
  if (has_cols) {
    const int c = tid >> 2, part = tid & 3;
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += tile[c][16 * part + r] * dk[16 * part + r];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (part == 0) yv[c0 + c] -= s;
  }
}

// ------------------------------------------------------------------------------------------------
void launch_chol_dp(double* S, int ld, int k, int T, double* Ld, double* Winv, int* status, hipStream_t s) {
  const int nt = T - k;  // row tiles below the diagonal block, incl. the RHS tile (>= 1)
  hipLaunchKernelGGL(k_chol_dp, dim3(nt), dim3(256), 0, s, S, ld, k, Ld, Winv, status);
}
// part 0: every trailing tile; part 1: only column k+1 (what the next diag+panel step needs);
// part 2: columns >= k+2 (can run beside the next diag+panel step on a second stream)
void launch_chol_update(double* S, int ld, int k, int T, int part, hipStream_t s) {
  int j0 = k + 1, j1 = T - 1;            // column tiles [j0, j1]
  if (part == 1) j1 = k + 1;
  if (part == 2) j0 = k + 2;
  if (j0 > j1 || j0 > T - 1) return;
  const int rows = T - j0 + 1;           // row tiles j0 .. T (RHS tile included)
  hipLaunchKernelGGL(k_chol_update, dim3(rows, j1 - j0 + 1), dim3(256), 0, s, S, ld, k, j0, j0);
}
void launch_chol_extract_y(const double* S, int ld, int T, double* yv, hipStream_t s) {
  hipLaunchKernelGGL(k_chol_extract_y, dim3((T * NB + 255) / 256), dim3(256), 0, s, S, ld, T, yv);
}
void launch_chol_bwd(const double* S, int ld, int k, const double* Ld, const double* Winv, double* yv, double* dp,
                     hipStream_t s) {
  hipLaunchKernelGGL(k_chol_bwd, dim3(k > 0 ? k : 1), dim3(256), 0, s, S, ld, k, Ld, Winv, yv, dp);
}

int chol_factor_solve(double* S, int ld, int T, double* Ld, double* Winv, double* yv, double* dp, int* status, hipStream_t s) {
  for (int k = 0; k < T; ++k) {
    launch_chol_dp(S, ld, k, T, Ld + (size_t)k * NB * NB, Winv + (size_t)k * 1024, status, s);
    launch_chol_update(S, ld, k, T, 0, s);
  }
  launch_chol_extract_y(S, ld, T, yv, s);
  for (int k = T - 1; k >= 0; --k) launch_chol_bwd(S, ld, k, Ld + (size_t)k * NB * NB, Winv + (size_t)k * 1024, yv, dp, s);
  return 0;
}

}  // namespace sl
