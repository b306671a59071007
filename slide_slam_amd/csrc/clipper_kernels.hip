// CLIPPER dense-clique solver on the device (clipper_semantic_object/src/clipper.cpp:172-323, DSD_HEU rounding :302-310).
//
// The reference keeps the affinity matrix M and the constraint pattern C (= sparsity pattern of M, clipper.cpp:55-64) as Eigen
// sparse matrices and runs projected gradient ascent with backtracking on F(u) = u^T (M + I - d (11^T - C - I)) u; d grows in an
// outer loop until no constraint is active.  Here: M in CSR (built on the device from the upper-filled dense affinity matrix), and
// the WHOLE solve — every product, reduction, clamp, normalisation, line-search and stopping decision — in ONE persistent workgroup
// of 1024 threads (k_clq_solve): the iteration is strictly sequential with O(nnz) work per evaluation, so there is nothing for more
// than one CU to do that would outweigh a grid-wide barrier per evaluation, and no host round trip remains inside the loop (the
// round-1 version copied three vectors over PCIe per evaluation).  Loops are bounded by maxoliters x maxiniters x maxlsiters.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace sl {

// row i of the symmetric matrix held as its upper triangle, dense row-major: nonzeros counted (rowcnt) or written (col / val) in
// ascending column order; one wave per row, ballot ranks keep the order
template <bool EMIT>
__global__ __launch_bounds__(256) void k_clq_csr(const double* __restrict__ Mup, int n, int* __restrict__ rowcnt, const int* __restrict__ rowptr,
                                                 int* __restrict__ col, double* __restrict__ val) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  int base = EMIT ? rowptr[i] : 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    double a = 0.0;
    if (j < n && j != i) a = j > i ? Mup[(size_t)i * n + j] : Mup[(size_t)j * n + i];
    const unsigned long long m = __ballot(a != 0.0);
    if (EMIT && a != 0.0) {
      const int k = base + __popcll(m & ((1ull << lane) - 1ull));
      col[k] = j;
      val[k] = a;
    }
    base += __popcll(m);
  }
  if (!EMIT && lane == 0) rowcnt[i] = base;
}


__device__ __forceinline__ double clq_block_sum(double x, double* sh) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
  __syncthreads();                      // (sh may still be read from the previous reduction)
  if ((tid & 63) == 0) sh[tid >> 6] = x;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];      // same order in every thread: identical result everywhere
  return s;
}
// Mu = M v, Cu = C v (C = pattern of M): wave per row
__device__ __forceinline__ void clq_prod(const ClqSolve& A, const double* __restrict__ v) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int i = wave; i < A.n; i += nw) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = A.rowptr[i] + lane; k < A.rowptr[i + 1]; k += 64) {
      const double x = v[A.col[k]];
      s1 += A.val[k] * x;
      s2 += x;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
    if (lane == 0) { A.Mu[i] = s1; A.Cu[i] = s2; }
  }
  __syncthreads();
}
// gradient of F at v for the current d into gout; returns F = v . gout      (clipper.cpp:224-226, 246-248)
__device__ __forceinline__ double clq_grad(const ClqSolve& A, const double* __restrict__ v, double d, double* __restrict__ gout, double* sh) {
  double part = 0.0;
  for (int i = threadIdx.x; i < A.n; i += blockDim.x) part += v[i];
  const double su = clq_block_sum(part, sh);
  clq_prod(A, v);
  part = 0.0;
  for (int i = threadIdx.x; i < A.n; i += blockDim.x) {
    const double gi = (1.0 + d) * v[i] - d * su + A.Mu[i] + A.Cu[i] * d;
    gout[i] = gi;
    part += v[i] * gi;
  }
  const double F = clq_block_sum(part, sh);
  __syncthreads();
  return F;
}
// mean over the active constraints of (Mu + u) / Cbu, Cbu = sum(u) - Cu - u   (clipper.cpp:201-216, 283-295)
__device__ __forceinline__ double clq_d_terms(const ClqSolve& A, const double* __restrict__ v, bool absval, int* cnt_out, double* sh) {
  double part = 0.0;
  for (int i = threadIdx.x; i < A.n; i += blockDim.x) part += v[i];
  const double su = clq_block_sum(part, sh);
  clq_prod(A, v);
  double acc = 0.0, cnt = 0.0;
  for (int i = threadIdx.x; i < A.n; i += blockDim.x) {
    const double cbu = su - A.Cu[i] - v[i];
    if (cbu > A.eps && v[i] > A.eps) {
      const double q = (A.Mu[i] + v[i]) / cbu;
      acc += absval ? fabs(q) : q;
      cnt += 1.0;
    }
  }
  const double sa = clq_block_sum(acc, sh);
  const double sc = clq_block_sum(cnt, sh);
  *cnt_out = (int)sc;
  return sc > 0.0 ? sa / sc : 0.0;
}

__device__ __forceinline__ void clq_solve_body(const ClqSolve& A) {
  __shared__ double sh[16];
  const int tid = threadIdx.x, nt = blockDim.x, n = A.n;
  // u <- normalised (M u0 + u0) or u0   (clipper.cpp:186-199)
  if (A.rescale) {
    clq_prod(A, A.u0);
    for (int i = tid; i < n; i += nt) A.u[i] = A.Mu[i] + A.u0[i];
  } else {
    for (int i = tid; i < n; i += nt) A.u[i] = A.u0[i];
  }
  __syncthreads();
  {
    double part = 0.0;
    for (int i = tid; i < n; i += nt) part += A.u[i] * A.u[i];
    const double nn = sqrt(clq_block_sum(part, sh));
    for (int i = tid; i < n; i += nt) A.u[i] /= nn;
    __syncthreads();
  }
  double d = 0.0, F = 0.0, evals = 0.0;
  int cnt = 0, outer = 0;
  {
    const double t = clq_d_terms(A, A.u, false, &cnt, sh);
    if (cnt > 0) d = t;
  }
  double *u = A.u, *unew = A.unew, *g = A.g, *gnew = A.gnew;
  for (outer = 0; outer < A.maxol; ++outer) {
    F = clq_grad(A, u, d, g, sh);
    evals += 1.0;
    for (int j = 0; j < A.maxin; ++j) {
      double alpha = 1.0, Fnew = 0.0, deltaF = 0.0;
      for (int k = 0; k < A.maxls; ++k) {
        // step, project onto u >= 0, back onto the sphere   (clipper.cpp:236-245)
        double part = 0.0;
        for (int i = tid; i < n; i += nt) {
          const double x = fmax(u[i] + alpha * g[i], 0.0);
          unew[i] = x;
          part += x * x;
        }
        const double nn = sqrt(clq_block_sum(part, sh));
        for (int i = tid; i < n; i += nt) unew[i] /= nn;
        __syncthreads();
        Fnew = clq_grad(A, unew, d, gnew, sh);
        evals += 1.0;
        deltaF = Fnew - F;
        if (deltaF < -A.eps) alpha *= A.beta;        // backtracking line search (every thread holds the same numbers)
        else break;
      }
      double part = 0.0;
      for (int i = tid; i < n; i += nt) { const double e = unew[i] - u[i]; part += e * e; }
      const double du = sqrt(clq_block_sum(part, sh));
      F = Fnew;
      double* t = u; u = unew; unew = t;
      t = g; g = gnew; gnew = t;
      __syncthreads();
      if (du < A.tol_u || fabs(deltaF) < A.tol_F) break;
    }
    const double deltad = clq_d_terms(A, u, true, &cnt, sh);
    if (cnt > 0) d += deltad;
    else break;
  }
  // the iterate ends up in A.u for the caller
  if (u != A.u) {
    for (int i = tid; i < n; i += nt) A.u[i] = u[i];
  }
  if (tid == 0) { A.out[0] = F; A.out[1] = d; A.out[2] = evals; A.out[3] = (double)outer; }
}

__global__ __launch_bounds__(1024) void k_clq_solve(ClqSolve A) { clq_solve_body(A); }
// several independent problems (the robot pairs of a multi-robot job, SURVEY 8e: 28 at eight robots), one persistent workgroup each
__global__ __launch_bounds__(1024) void k_clq_solve_b(const ClqSolve* __restrict__ jobs) {
  const ClqSolve A = jobs[blockIdx.x];
  if (A.n > 0) clq_solve_body(A);
}
void launch_clq_solve_batch(const ClqSolve* d_jobs, int n_jobs, hipStream_t s) {
  if (n_jobs > 0) hipLaunchKernelGGL(k_clq_solve_b, dim3(n_jobs), dim3(1024), 0, s, d_jobs);
}

void launch_clq_csr_count(const double* Mup, int n, int* rowcnt, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_clq_csr<false>, dim3((n + 3) / 4), dim3(256), 0, s, Mup, n, rowcnt, nullptr, nullptr, nullptr);
}
void launch_clq_csr_fill(const double* Mup, int n, const int* rowptr, int* col, double* val, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_clq_csr<true>, dim3((n + 3) / 4), dim3(256), 0, s, Mup, n, nullptr, rowptr, col, val);
}
void launch_clq_solve(const int* rowptr, const int* col, const double* val, int n, const double* u0, double* work6n, double tol_u, double tol_F,
                      double beta, double eps, int maxin, int maxol, int maxls, int rescale, double* out4, hipStream_t s) {
  ClqSolve A;
  A.rowptr = rowptr; A.col = col; A.val = val; A.n = n; A.u0 = u0;
  A.u = work6n; A.unew = work6n + n; A.g = work6n + 2 * (size_t)n; A.gnew = work6n + 3 * (size_t)n; A.Mu = work6n + 4 * (size_t)n; A.Cu = work6n + 5 * (size_t)n;
  A.tol_u = tol_u; A.tol_F = tol_F; A.beta = beta; A.eps = eps; A.maxin = maxin; A.maxol = maxol; A.maxls = maxls; A.rescale = rescale;
  A.out = out4;
  hipLaunchKernelGGL(k_clq_solve, dim3(1), dim3(1024), 0, s, A);
}

}  // namespace sl
