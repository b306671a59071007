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
// POTRF of one 64x64 diagonal block plus its explicit inverse W = L^-1, entirely in registers:
// 256 threads as a 16x16 grid, thread (ti, tj) owns A[ti + 16p][tj + 16q] and W[..][..], p, q in 0..3.
// Per column step the owners of column j (of A) and of row j (of W) publish them through a
// double-buffered LDS vector; one barrier per step, no integer division, no scratch.
template <int JQ>
__device__ inline void diag_steps(double (&a)[4][4], double (&w)[4][4], double (*colraw)[NB], double (*wraw)[NB], int ti, int tj,
                                  int* status) {
#pragma unroll 1
  for (int jt = 0; jt < 16; ++jt) {
    const int j = 16 * JQ + jt;
    const int buf = j & 1;
    if (tj == jt) {
#pragma unroll
      for (int p = 0; p < 4; ++p) colraw[buf][ti + 16 * p] = a[p][JQ];
    }
    if (ti == jt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) wraw[buf][tj + 16 * q] = w[JQ][q];
    }
    __syncthreads();
    double d = colraw[buf][j];
    if (!(d > 0.0)) {
      if (ti == 0 && tj == 0) atomicOr(&status[1], 1);
      d = 1.0;
    }
    const double inv = 1.0 / sqrt(d);
    double li[4], lk[4], wr[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) li[p] = colraw[buf][ti + 16 * p] * inv;
#pragma unroll
    for (int q = 0; q < 4; ++q) { lk[q] = colraw[buf][tj + 16 * q] * inv; wr[q] = wraw[buf][tj + 16 * q] * inv; }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int i = ti + 16 * p;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = tj + 16 * q;
        if (i > j && c > j && c <= i) a[p][q] -= li[p] * lk[q];   // trailing update
        if (i > j && c <= j) w[p][q] -= li[p] * wr[q];            // forward substitution on the identity
      }
    }
    if (tj == jt) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (ti + 16 * p >= j) a[p][JQ] = li[p];                   // column j of L is final
    }
    if (ti == jt) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (tj + 16 * q <= j) w[JQ][q] = wr[q];                   // row j of W is final
    }
  }
}

__global__ __launch_bounds__(256) void k_chol_diag(double* __restrict__ S, int ld, int k, double* __restrict__ W,
                                                   int* status) {
  __shared__ double colraw[2][NB];
  __shared__ double wraw[2][NB];
  const int tid = threadIdx.x, ti = tid & 15, tj = tid >> 4;
  double* base = S + (size_t)(k * NB) * ld + (size_t)k * NB;
  double a[4][4], w[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = ti + 16 * p, c = tj + 16 * q;
      a[p][q] = (i >= c) ? base[(size_t)c * ld + i] : 0.0;
      w[p][q] = (i == c) ? 1.0 : 0.0;
    }
  diag_steps<0>(a, w, colraw, wraw, ti, tj, status);
  diag_steps<1>(a, w, colraw, wraw, ti, tj, status);
  diag_steps<2>(a, w, colraw, wraw, ti, tj, status);
  diag_steps<3>(a, w, colraw, wraw, ti, tj, status);
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = ti + 16 * p, c = tj + 16 * q;
      if (i >= c) base[(size_t)c * ld + i] = a[p][q];
      W[(size_t)c * NB + i] = (i >= c) ? w[p][q] : 0.0;
    }
}

// X = A_ik W^T  for every row tile below the diagonal block (incl. the RHS tile); 16 rows per wave.
__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ S, int ld, int k, const double* __restrict__ W) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int it = k + 1 + blockIdx.x;
  const int lr = lane & 15, lk = lane >> 4;
  double* col = S + (size_t)(k * NB) * ld + (size_t)it * NB + 16 * wave + lr;
  double a[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) a[ks] = col[(size_t)(4 * ks + lk) * ld];
  v4d acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) acc[nb] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const double w = W[(size_t)(4 * ks + lk) * NB + 16 * nb + lr];   // W[n = 16 nb + lr][j = 4 ks + lk]
      acc[nb] = mfma_f64(w, a[ks], acc[nb]);                          // D[p = n][q = m]
    }
  }
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) col[(size_t)(16 * nb + lk + 4 * r) * ld] = acc[nb][r];
}

// C_ij -= L_ik L_jk^T over the trailing tiles (k < j <= i, j < T; i == T is the RHS tile)
__global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ S, int ld, int k, int T) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long t = blockIdx.x;
  long long ii = (long long)floor((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (ii * (ii + 1) / 2 > t) --ii;
  while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
  const int jj = (int)(t - ii * (ii + 1) / 2);
  const int i = k + 1 + (int)ii, j = k + 1 + jj;
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

// backward substitution step k:  d_k = W_k^T y_k ;  y_c -= L[k-block, c]^T d_k  for every column c < k*NB.
// One workgroup per 64 columns: the 64x64 tile of L is read in whole 512-byte column runs and reduced
// over rows through an LDS transpose (pad 1), 4 threads per column.
__global__ __launch_bounds__(256) void k_chol_bwd(const double* __restrict__ S, int ld, int k, const double* __restrict__ W,
                                                  double* __restrict__ yv, double* __restrict__ dp) {
  __shared__ double yk[NB];
  __shared__ double dk[NB];
  __shared__ double tile[NB][NB + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < NB) yk[tid] = yv[k * NB + tid];
  __syncthreads();
  // d_k[c] = sum_r W[r][c] y[r],  W[r][c] at W[c*NB + r]: one wave per column, lanes over r
  for (int c = wave; c < NB; c += 4) {
    double v = W[(size_t)c * NB + lane] * yk[lane];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) {
      dk[c] = v;
      if (blockIdx.x == 0) dp[k * NB + c] = v;
    }
  }
  const int c0 = blockIdx.x * NB;
  if (c0 < k * NB) {
    for (int cc = wave; cc < NB; cc += 4) tile[cc][lane] = S[(size_t)(c0 + cc) * ld + (size_t)k * NB + lane];
  }
  __syncthreads();
  if (c0 < k * NB) {
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
void launch_chol_diag(double* S, int ld, int k, double* W, int* status, hipStream_t s) {
  hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), 0, s, S, ld, k, W, status);
}
void launch_chol_panel(double* S, int ld, int k, int T, const double* W, hipStream_t s) {
  const int nt = T - k;  // row tiles below the diagonal block, incl. the RHS tile
  if (nt > 0) hipLaunchKernelGGL(k_chol_panel, dim3(nt), dim3(256), 0, s, S, ld, k, W);
}
void launch_chol_update(double* S, int ld, int k, int T, hipStream_t s) {
  const long long nt = T - k;
  const long long cnt = nt * (nt + 1) / 2 - 1;  // lower tile pairs minus the (RHS, RHS) corner
  if (cnt > 0) hipLaunchKernelGGL(k_chol_update, dim3((unsigned)cnt), dim3(256), 0, s, S, ld, k, T);
}
void launch_chol_extract_y(const double* S, int ld, int T, double* yv, hipStream_t s) {
  hipLaunchKernelGGL(k_chol_extract_y, dim3((T * NB + 255) / 256), dim3(256), 0, s, S, ld, T, yv);
}
void launch_chol_bwd(const double* S, int ld, int k, const double* W, double* yv, double* dp, hipStream_t s) {
  hipLaunchKernelGGL(k_chol_bwd, dim3(k > 0 ? k : 1), dim3(256), 0, s, S, ld, k, W, yv, dp);
}

int chol_factor_solve(double* S, int ld, int T, double* W, double* yv, double* dp, int* status, hipStream_t s) {
  for (int k = 0; k < T; ++k) {
    double* Wk = W + (size_t)k * NB * NB;
    launch_chol_diag(S, ld, k, Wk, status, s);
    launch_chol_panel(S, ld, k, T, Wk, s);
    launch_chol_update(S, ld, k, T, s);
  }
  launch_chol_extract_y(S, ld, T, yv, s);
  for (int k = T - 1; k >= 0; --k) launch_chol_bwd(S, ld, k, W + (size_t)k * NB * NB, yv, dp, s);
  return 0;
}

}  // namespace sl
