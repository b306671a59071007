// CLIPPER dense-clique solver on the device (clipper_semantic_object/src/clipper.cpp:172-323, DSD_HEU rounding :302-310).
//
// The reference keeps the affinity matrix M and the constraint pattern C (= sparsity pattern of M, clipper.cpp:55-64) as Eigen
// sparse matrices and runs projected gradient ascent with backtracking on F(u) = u^T (M + I - d (11^T - C - I)) u; d grows in an
// outer loop until no constraint is active.  Here: M in CSR (built on the device from the upper-filled dense affinity matrix), and
// the WHOLE solve — every product, reduction, clamp, normalisation, line-search and stopping decision — in ONE launch with no host
// round trip inside the loop: a persistent workgroup of 1024 threads per problem (k_clq_solve, k_clq_solve_b: the iteration is
// strictly sequential with O(nnz) work per evaluation, and the problems sloam meets — tens to hundreds of associations — leave a
// second CU nothing to do that outweighs a grid barrier), and for ONE LARGE problem (m in the thousands) k_clq_solve_coop:
// the rows of the product over the waves of up to 128 co-resident workgroups, one grid barrier per evaluation, iterates bit-identical
// to the one-workgroup kernel's.  Loops are bounded by maxoliters x maxiniters x maxlsiters.
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

// ---- one LARGE problem on several workgroups (COOP) ---------------------------------------------------------------------------------
// The rows of the product are dealt to the waves of ALL workgroups; everything else — the step, the projection, every reduction,
// the line search and the stopping decisions — each workgroup repeats for the whole vector on PRIVATE copies of u / unew / g / gnew,
// with the very code and thread layout of the one-workgroup solve.  So every workgroup holds bit-identical iterates, takes the same
// branches and reaches the same grid barriers, the iterates equal the one-workgroup kernel's bit for bit, and the only data crossing
// workgroups are Mu / Cu: written with agent-scope (write-through) stores, read with agent-scope loads (the per-XCD L2s are not
// coherent: profiles/r04_tile_hop_bench.txt has the protocol measurements), double-buffered by product parity because a fast
// workgroup may start the next product while a slow one still reads this one's result — ONE grid barrier per product.
struct ClqGrid {
  int* bar;       // [0] arrivals (monotonic), [1] abort word
  int n_wg;
  int epoch;      // barriers passed (same in every thread of every workgroup)
  int par;        // parity of the product being computed
};
template <bool COOP> __device__ __forceinline__ double clq_ld(const double* p) {
  if (COOP) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
// false: another workgroup gave up (or this one waited too long) — every caller returns at once, so the grid drains
__device__ __forceinline__ bool clq_grid_sync(ClqGrid& G) {
  __shared__ int s_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's Mu / Cu stores have left
  __syncthreads();
  G.epoch += 1;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(G.bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int target = G.epoch * G.n_wg;
    int good = 1, spins = 0;
    while (__hip_atomic_load(G.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (__hip_atomic_load(G.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || ++spins > (1 << 24)) {
        __hip_atomic_store(G.bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        good = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    s_ok = good;
  }
  __syncthreads();
  return s_ok != 0;
}

// Mu = M v, Cu = C v (C = pattern of M): wave per row.  COOP: rows over the waves of the whole grid, results into the buffer of this
// product's parity, then the grid barrier; returns false when the grid is being abandoned.
template <bool COOP>
__device__ __forceinline__ bool clq_prod(const ClqSolve& A, ClqGrid& G, const double* __restrict__ v, double*& Mu, double*& Cu) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  Mu = A.Mu + (COOP ? (size_t)G.par * 2 * A.n : 0);
  Cu = A.Cu + (COOP ? (size_t)G.par * 2 * A.n : 0);
  const int first = COOP ? (int)blockIdx.x * nw + wave : wave, step = COOP ? G.n_wg * nw : nw;
  for (int i = first; i < A.n; i += step) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = A.rowptr[i] + lane; k < A.rowptr[i + 1]; k += 64) {
      const double x = v[A.col[k]];
      s1 += A.val[k] * x;
      s2 += x;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
    if (lane == 0) {
      if (COOP) {
        __hip_atomic_store(Mu + i, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(Cu + i, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        Mu[i] = s1; Cu[i] = s2;
      }
    }
  }
  if (COOP) {
    G.par ^= 1;
    return clq_grid_sync(G);
  }
  __syncthreads();
  return true;
}
// gradient of F at v for the current d into gout; F = v . gout      (clipper.cpp:224-226, 246-248)
template <bool COOP>
__device__ __forceinline__ bool clq_grad(const ClqSolve& A, ClqGrid& G, const double* __restrict__ v, double d, double* __restrict__ gout, double* sh,
                                         double* F_out) {
  double part = 0.0;
  for (int i = threadIdx.x; i < A.n; i += blockDim.x) part += v[i];
  const double su = clq_block_sum(part, sh);
  double *Mu, *Cu;
  if (!clq_prod<COOP>(A, G, v, Mu, Cu)) return false;
  part = 0.0;
  for (int i = threadIdx.x; i < A.n; i += blockDim.x) {
    const double gi = (1.0 + d) * v[i] - d * su + clq_ld<COOP>(Mu + i) + clq_ld<COOP>(Cu + i) * d;
    gout[i] = gi;
    part += v[i] * gi;
  }
  *F_out = clq_block_sum(part, sh);
  __syncthreads();
  return true;
}
// mean over the active constraints of (Mu + u) / Cbu, Cbu = sum(u) - Cu - u   (clipper.cpp:201-216, 283-295)
template <bool COOP>
__device__ __forceinline__ bool clq_d_terms(const ClqSolve& A, ClqGrid& G, const double* __restrict__ v, bool absval, int* cnt_out, double* sh,
                                            double* mean_out) {
  double part = 0.0;
  for (int i = threadIdx.x; i < A.n; i += blockDim.x) part += v[i];
  const double su = clq_block_sum(part, sh);
  double *Mu, *Cu;
  if (!clq_prod<COOP>(A, G, v, Mu, Cu)) return false;
  double acc = 0.0, cnt = 0.0;
  for (int i = threadIdx.x; i < A.n; i += blockDim.x) {
    const double cbu = su - clq_ld<COOP>(Cu + i) - v[i];
    if (cbu > A.eps && v[i] > A.eps) {
      const double q = (clq_ld<COOP>(Mu + i) + v[i]) / cbu;
      acc += absval ? fabs(q) : q;
      cnt += 1.0;
    }
  }
  const double sa = clq_block_sum(acc, sh);
  const double sc = clq_block_sum(cnt, sh);
  *cnt_out = (int)sc;
  *mean_out = sc > 0.0 ? sa / sc : 0.0;
  return true;
}

// uw: this workgroup's u / unew / g / gnew (4 n doubles; A.u itself for the one-workgroup solve)
template <bool COOP>
__device__ __forceinline__ void clq_solve_body(const ClqSolve& A, ClqGrid& G, double* uw) {
  __shared__ double sh[16];
  const int tid = threadIdx.x, nt = blockDim.x, n = A.n;
  double *u = uw, *unew = uw + n, *g = uw + 2 * (size_t)n, *gnew = uw + 3 * (size_t)n;
  // u <- normalised (M u0 + u0) or u0   (clipper.cpp:186-199)
  if (A.rescale) {
    double *Mu, *Cu;
    if (!clq_prod<COOP>(A, G, A.u0, Mu, Cu)) return;
    for (int i = tid; i < n; i += nt) u[i] = clq_ld<COOP>(Mu + i) + A.u0[i];
  } else {
    for (int i = tid; i < n; i += nt) u[i] = A.u0[i];
  }
  __syncthreads();
  {
    double part = 0.0;
    for (int i = tid; i < n; i += nt) part += u[i] * u[i];
    const double nn = sqrt(clq_block_sum(part, sh));
    for (int i = tid; i < n; i += nt) u[i] /= nn;
    __syncthreads();
  }
  double d = 0.0, F = 0.0, evals = 0.0;
  int cnt = 0, outer = 0;
  {
    double t;
    if (!clq_d_terms<COOP>(A, G, u, false, &cnt, sh, &t)) return;
    if (cnt > 0) d = t;
  }
  for (outer = 0; outer < A.maxol; ++outer) {
    if (!clq_grad<COOP>(A, G, u, d, g, sh, &F)) return;
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
        if (!clq_grad<COOP>(A, G, unew, d, gnew, sh, &Fnew)) return;
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
    double deltad;
    if (!clq_d_terms<COOP>(A, G, u, true, &cnt, sh, &deltad)) return;
    if (cnt > 0) d += deltad;
    else break;
  }
  // the iterate ends up in A.u for the caller (COOP: workgroup 0 hands over its copy)
  if (!COOP || blockIdx.x == 0) {
    if (u != A.u) {
      for (int i = tid; i < n; i += nt) A.u[i] = u[i];
    }
    if (tid == 0) { A.out[0] = F; A.out[1] = d; A.out[2] = evals; A.out[3] = (double)outer; }
  }
}

__global__ __launch_bounds__(1024) void k_clq_solve(ClqSolve A) {
  ClqGrid G{nullptr, 1, 0, 0};
  clq_solve_body<false>(A, G, A.u);
}
// several independent problems (the robot pairs of a multi-robot job, SURVEY 8e: 28 at eight robots), one persistent workgroup each
__global__ __launch_bounds__(1024) void k_clq_solve_b(const ClqSolve* __restrict__ jobs) {
  const ClqSolve A = jobs[blockIdx.x];
  ClqGrid G{nullptr, 1, 0, 0};
  if (A.n > 0) clq_solve_body<false>(A, G, A.u);
}
// one large problem on n_wg co-resident workgroups (cooperative launch); priv: n_wg x 4 n doubles, bar: 2 ints, zero at launch;
// A.Mu: 4 n doubles, [parity][Mu | Cu] (A.Cu = A.Mu + n); A.u receives the result
__global__ __launch_bounds__(1024) void k_clq_solve_coop(ClqSolve A, double* priv, int* bar, int n_wg) {
  ClqGrid G{bar, n_wg, 0, 0};
  clq_solve_body<true>(A, G, priv + (size_t)blockIdx.x * 4 * A.n);
  // a workgroup that left through a failed barrier reports it (out[2] < 0: the host turns it into an error)
  if (threadIdx.x == 0 && blockIdx.x == 0 && __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) A.out[2] = -1.0;
}
void launch_clq_solve_batch(const ClqSolve* d_jobs, int n_jobs, hipStream_t s) {
  if (n_jobs > 0) hipLaunchKernelGGL(k_clq_solve_b, dim3(n_jobs), dim3(1024), 0, s, d_jobs);
}
// false: the cooperative launch was refused (the caller falls back to the one-workgroup kernel)
bool launch_clq_solve_coop(const ClqSolve& A, double* priv, int* bar2, int n_wg, hipStream_t s) {
  if (hipMemsetAsync(bar2, 0, 2 * sizeof(int), s) != hipSuccess) return false;
  ClqSolve a = A;
  void* args[] = {&a, &priv, &bar2, &n_wg};
  const hipError_t e = hipLaunchCooperativeKernel((const void*)k_clq_solve_coop, dim3(n_wg), dim3(1024), args, 0, s);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return true;
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
