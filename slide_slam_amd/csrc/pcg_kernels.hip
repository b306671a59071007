// Joint Gauss-Newton step of several robot sub-graphs that share landmarks: preconditioned conjugate gradients on the GLOBAL
// landmark-eliminated pose system, preconditioned by the robots' own Cholesky factors.
//
// The reference solves the joint graph of all robots in every host replica (graph.cpp:260-272 on a graph that holds every
// robot's poses, sloamNode.cpp:912-1002).  Here robot a owns the block S_a = H_pp,a - W_a H_ll^-1 W_a^T of the reduced system
// (H_ll already the all-reduced GLOBAL landmark blocks); the coupling between robots is the Schur fill through the shared
// landmarks,   S = blockdiag(S_a) - [W_a H_ll^-1 W_b^T]_{a != b}.   One pass factors every S_a (chol_kernels.hip) and then runs a
// fixed number of PCG iterations on S delta = b, in the Chronopoulos-Gear form (one scalar reduction per iteration):
//     u = M^-1 r                      M = blockdiag(S_a): forward + backward chain on the factors
//     t_l = W_a,l^T u_a               per shared landmark, packed into the exchange buffer            -> all-reduce(sum): u_l
//     w_a = S_a u_a - sum_l W_a,l H_l^-1 (u_l - t_l)        dense symmetric product with the saved S_a + a gather over the factors
//     gamma = (r, u), delta = (w, u)                                                                   -> all-reduce(sum) of 2 scalars
//     beta = gamma / gamma_old,  alpha = gamma / (delta - beta gamma / alpha_old)     (first: beta = 0, alpha = gamma / delta)
//     p = u + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s
// Block-Jacobi alone (x = M^-1 b, what a pass did before) does not converge once the robots share more than a few landmarks: its
// iteration matrix has eigenvalue pairs +-mu with mu -> 1 (measured on C3: stalls 1.3e-3 from the joint optimum).
// Every kernel is batched over the robots of the GPU (blockIdx.z), deterministic (no floating-point atomics), scratch-free.
#include <hip/hip_runtime.h>

#include "graph_dev.hpp"
#include "kernels.hpp"

namespace sl {

enum { PV_R = 0, PV_U = 1, PV_W = 2, PV_P = 3, PV_S = 4, PV_X = 5, PV_Y = 6, PV_COUNT = 7 };
// pcg_scal: [0] gamma_old, [1] alpha_old, [2] alpha, [3] beta, [4] |r0|^2-like first gamma (diagnostic), [5] last gamma
__device__ __forceinline__ double* pvec(const GraphDev& G, int i) { return G.pcg + (size_t)i * G.T * NB; }

// r = b (the reduced gradient, -g padded with zeros), u = M^-1 b (the block solve the factorisation already delivered), x = 0
__global__ __launch_bounds__(256) void k_pcg_init(const GraphDev* __restrict__ Gs) {
  const GraphDev G = Gs[blockIdx.z];
  const int i = blockIdx.x * 256 + threadIdx.x, nT = G.T * NB;
  if (i >= nT) return;
  pvec(G, PV_R)[i] = i < 6 * G.P ? -G.pose_g[i] : 0.0;
  pvec(G, PV_U)[i] = G.dp[i];
  pvec(G, PV_X)[i] = 0.0;
  pvec(G, PV_P)[i] = 0.0;
  pvec(G, PV_S)[i] = 0.0;
  if (i < 8) G.pcg_scal[i] = 0.0;
}

struct PcgBufs { double* p[8]; };

// t_l = sum_f E_f^T v_pose(f) for the landmarks of the shared slots, written to lm_t and packed (9 per slot) into the robot's
// exchange buffer; slots this robot does not observe contribute zeros.  One wave per slot.
__device__ __forceinline__ void pcg_tl_body(const GraphDev& G, const PcgBufs& B, int blk, int vec) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sidx = blk * 4 + wave;
  if (sidx >= G.n_slots) return;
  double* out = B.p[blockIdx.z] + 9 * (size_t)sidx;
  const int l = G.sh_lid[sidx];
  if (l < 0) {
    if (lane < 9) out[lane] = 0.0;
    return;
  }
  const double* v = pvec(G, vec);
  const int D = lm_dim(G.lm_type[l]);
  double rhs[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) rhs[k] = 0.0;
  for (int q = G.lm_ptr[l] + lane; q < G.lm_ptr[l + 1]; q += 64) {
    const int f = G.lm_fids[q];
    const double* E = G.ebuf + G.lf_eoff[f];
    const double* d = v + 6 * (size_t)G.lf_pose[f];
    double dd[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) dd[a] = d[a];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (k < D) {
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) s += E[a * D + k] * dd[a];
        rhs[k] += s;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) rhs[k] += __shfl_xor(rhs[k], m);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      G.lm_t[9 * (size_t)l + k] = rhs[k];
      out[k] = rhs[k];
    }
  }
}

// out = S0 * in for the symmetric S0 held as its lower triangle (column-major, leading dimension ld): one workgroup per block row
// i of 64: tiles (i, j <= i) as they lie, tiles (j > i, i) transposed.  Deterministic (no atomics): every block row is summed by
// one workgroup in a fixed order.
__device__ __forceinline__ void pcg_symv_body(const GraphDev& G, int bi, int vin, int vout, int in_lds) {
  if (bi >= G.T) return;
  extern __shared__ double xs_lds[];      // the whole input vector (T * 64 doubles) when it fits: no barrier inside the tile loops
  __shared__ double red[4][NB];
  __shared__ double redt[NB][5];
  const int tid = threadIdx.x, row = tid & 63, cp = tid >> 6;
  const double* x = pvec(G, vin);
  const double* S0 = G.S0;
  const int ld = G.ld, nT = G.T * NB;
  if (in_lds) {
    for (int i = tid; i < nT; i += 256) xs_lds[i] = x[i];
    __syncthreads();
  }
  const double* xs = in_lds ? xs_lds : x;
  double acc = 0.0;
  // tiles (bi, j), j < bi: y[row] += sum_c tile[row][c] x_j[c]; thread (row, cp) takes columns 16 cp ..; the next tile's sixteen
  // loads are issued before this tile's products (two register sets): the loop is a pure stream, latency is all there is to hide
  {
    double ta[16], tb[16];
    const double* tp0 = S0 + (size_t)(16 * cp) * ld + (size_t)bi * NB + row;
    const int jf = G.first ? G.first[bi] : 0;      // tiles left of the profile are structurally zero
    if (jf < bi) {
      const double* tp = tp0 + (size_t)(jf * NB) * ld;
#pragma unroll
      for (int r = 0; r < 16; ++r) ta[r] = tp[(size_t)r * ld];
    }
    for (int j = jf; j < bi; j += 2) {
      if (j + 1 < bi) {
        const double* tp = tp0 + (size_t)((j + 1) * NB) * ld;
#pragma unroll
        for (int r = 0; r < 16; ++r) tb[r] = tp[(size_t)r * ld];
      }
      {
        const double* xj = xs + j * NB + 16 * cp;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += ta[r] * xj[r];
      }
      if (j + 2 < bi) {
        const double* tp = tp0 + (size_t)((j + 2) * NB) * ld;
#pragma unroll
        for (int r = 0; r < 16; ++r) ta[r] = tp[(size_t)r * ld];
      }
      if (j + 1 < bi) {
        const double* xj = xs + (j + 1) * NB + 16 * cp;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += tb[r] * xj[r];
      }
    }
  }
  // diagonal tile: lower triangle only
  {
    const double* tp = S0 + (size_t)(bi * NB) * ld + (size_t)bi * NB;
    const double* xj = xs + bi * NB;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = 16 * cp + r;
      const double v = c <= row ? tp[(size_t)c * ld + row] : tp[(size_t)row * ld + c];
      acc += v * xj[c];
    }
  }
  red[cp][row] = acc;
  // tiles (j, bi), j > bi, transposed: y[c] += sum_r tile[r][c] x_j[r]; thread (col = tid >> 2, part = tid & 3) takes rows 16 part ..
  const int col = tid >> 2, part = tid & 3;
  double acct = 0.0;
  {
    double ta[16], tb[16];
    const double* tp0 = S0 + (size_t)(bi * NB + col) * ld + 16 * part;
    const int j0 = bi + 1, T = G.prof ? G.prof[bi] + 1 : G.T;      // tiles below the profile of column bi: zero
    if (j0 < T) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ta[r] = tp0[(size_t)j0 * NB + r];
    }
    for (int j = j0; j < T; j += 2) {
      if (j + 1 < T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tb[r] = tp0[(size_t)(j + 1) * NB + r];
      }
      {
        const double* xj = xs + j * NB + 16 * part;
#pragma unroll
        for (int r = 0; r < 16; ++r) acct += ta[r] * xj[r];
      }
      if (j + 2 < T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ta[r] = tp0[(size_t)(j + 2) * NB + r];
      }
      if (j + 1 < T) {
        const double* xj = xs + (j + 1) * NB + 16 * part;
#pragma unroll
        for (int r = 0; r < 16; ++r) acct += tb[r] * xj[r];
      }
    }
  }
  redt[col][part] = acct;
  __syncthreads();
  if (tid < NB) {
    const double a = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    const double b = (redt[tid][0] + redt[tid][1]) + (redt[tid][2] + redt[tid][3]);
    pvec(G, vout)[(size_t)bi * NB + tid] = a + b;
  }
}

// t_l(v) and w = S0 v in ONE launch: both need v only (the products do not wait for the exchange of t_l), so the first n_row_blocks
// workgroups of a robot take the block rows of the product, the others four shared slots each
__global__ __launch_bounds__(256) void k_pcg_tl_symv(const GraphDev* __restrict__ Gs, PcgBufs B, int vec, int vout, int n_row_blocks, int in_lds) {
  const GraphDev G = Gs[blockIdx.z];
  if ((int)blockIdx.x < n_row_blocks) pcg_symv_body(G, blockIdx.x, vec, vout, in_lds);
  else pcg_tl_body(G, B, (int)blockIdx.x - n_row_blocks, vec);
}

// w_p -= sum_{f at pose p} F_f c_lm(f): the cross-robot Schur fill through the shared landmarks, c_l = (sum over all robots of t_l,
// from the exchange buffer) - own t_l; a landmark without a slot contributes nothing.  One wave per pose, lanes over its landmark factors.
// nsum > 1 (whole-pass graphs of one GPU): the exchange buffers of the nsum robots still hold their own t_l — the sum is taken
// here, in the order k_sum_bcast takes it (same arithmetic); nsum = 1: the buffer holds the all-reduced sum already.
__global__ __launch_bounds__(256) void k_pcg_cross(const GraphDev* __restrict__ Gs, PcgBufs B, int vout, int nsum) {
  const GraphDev G = Gs[blockIdx.z];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + wave;
  if (p >= G.P) return;
  double y[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) y[a] = 0.0;
  for (int q = G.pose_ptr[p] + lane; q < G.pose_ptr[p + 1]; q += 64) {
    const int l = G.pose_lms[q];
    const int sl = G.lm_slot[l];
    if (sl < 0) continue;
    const long long ed = G.pose_ed[q];
    const int D = (int)(ed & 15);
    const double* town = G.lm_t + 9 * (size_t)l;
    double cc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cc[k] = 0.0;
    if (nsum > 1) {
      // all loads of one component first (independent), then its sum in buffer order
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        if (k < D) {
          double t[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) t[r] = r < nsum ? B.p[r][9 * (size_t)sl + k] : 0.0;
          double acc = 0.0;
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (r < nsum) acc += t[r];
          cc[k] = acc;
        }
      }
    } else {
      const double* tsum = B.p[blockIdx.z] + 9 * (size_t)sl;
#pragma unroll
      for (int k = 0; k < 9; ++k) cc[k] = k < D ? tsum[k] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) cc[k] = k < D ? cc[k] - town[k] : 0.0;
    const double* F = G.ebuf + (ed >> 4) + 6 * D;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if (k < D) s += F[a * D + k] * cc[k];
      y[a] += s;
    }
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) y[a] += __shfl_xor(y[a], m);
  }
  if (lane < 6) {
    double v = 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a) v = lane == a ? y[a] : v;
    pvec(G, vout)[6 * (size_t)p + lane] -= v;
  }
}

// partial dot products of one robot, (r, u) and (w, u), into the first two doubles of its exchange buffer (one workgroup per
// robot: fixed summation order)
__global__ __launch_bounds__(256) void k_pcg_dots(const GraphDev* __restrict__ Gs, PcgBufs B) {
  const GraphDev G = Gs[blockIdx.z];
  __shared__ double sh[2][256];
  const int tid = threadIdx.x, nT = G.T * NB;
  const double* r = pvec(G, PV_R);
  const double* u = pvec(G, PV_U);
  const double* w = pvec(G, PV_W);
  double a = 0.0, b = 0.0;
  for (int i = tid; i < nT; i += 256) {
    const double ui = u[i];
    a += r[i] * ui;
    b += w[i] * ui;
  }
  sh[0][tid] = a;
  sh[1][tid] = b;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) { sh[0][tid] += sh[0][tid + st]; sh[1][tid] += sh[1][tid + st]; }
    __syncthreads();
  }
  if (tid == 0) { B.p[blockIdx.z][0] = sh[0][0]; B.p[blockIdx.z][1] = sh[1][0]; }
}

// alpha, beta of this iteration from the all-reduced (gamma, delta) in the exchange buffer; every robot computes the same numbers
__global__ void k_pcg_scalars(const GraphDev* __restrict__ Gs, PcgBufs B, int nsum) {
  const GraphDev G = Gs[blockIdx.x];
  if (threadIdx.x != 0) return;
  double gamma = B.p[blockIdx.x][0], delta = B.p[blockIdx.x][1];
  if (nsum > 1) {              // the partial dot products of the nsum robots of this GPU, summed in k_sum_bcast's order
    gamma = 0.0; delta = 0.0;
    for (int r = 0; r < nsum; ++r) { gamma += B.p[r][0]; delta += B.p[r][1]; }
  }
  double* sc = G.pcg_scal;
  const double gamma_old = sc[0], alpha_old = sc[1];
  double beta = 0.0, alpha;
  if (gamma_old > 0.0) {
    beta = gamma / gamma_old;
    alpha = gamma / (delta - beta * gamma / alpha_old);
  } else {
    alpha = gamma / delta;
    sc[4] = gamma;
  }
  // converged: gamma has fallen to tol^2 (at least eps^2) of its first value — r is at rounding level, the recurrence's denominator
  // would only cancel; the step and every later one are no-ops (alpha = beta = 0 leaves x and r)
  const double t2 = G.pcg_tol2 > 4.930380657631324e-32 ? G.pcg_tol2 : 4.930380657631324e-32;
  if (!(gamma > 0.0) || (gamma_old > 0.0 && gamma <= t2 * sc[4])) { alpha = 0.0; beta = 0.0; }          // r = 0 already (or NaN upstream): a no-op step, x stays
  else if (!(alpha > 0.0) || !(alpha < 1e300)) { alpha = 0.0; beta = 0.0; atomicOr(&G.status[1], 4); }     // breakdown: not positive definite
  else sc[6] += 1.0;
  sc[0] = gamma > 0.0 ? gamma : gamma_old;
  sc[1] = alpha > 0.0 ? alpha : alpha_old;
  sc[2] = alpha;
  sc[3] = beta;
  sc[5] = gamma;
}

// p = u + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s.  Also prepares the forward substitution that follows (u = M^-1 r):
// its output y pre-filled with the sentinel of the chained kernels (chol_kernels.hip BWD_SENT), its ticket counter (status[5]) cleared.
__global__ __launch_bounds__(256) void k_pcg_update(const GraphDev* __restrict__ Gs) {
  const GraphDev G = Gs[blockIdx.z];
  const int i = blockIdx.x * 256 + threadIdx.x, nT = G.T * NB;
  if (i == 0) G.status[5] = 0;
  if (i >= nT) return;
  pvec(G, PV_Y)[i] = __longlong_as_double((long long)CHAIN_SENTINEL);
  const double alpha = G.pcg_scal[2], beta = G.pcg_scal[3];
  const double p = pvec(G, PV_U)[i] + beta * pvec(G, PV_P)[i];
  const double s = pvec(G, PV_W)[i] + beta * pvec(G, PV_S)[i];
  pvec(G, PV_P)[i] = p;
  pvec(G, PV_S)[i] = s;
  pvec(G, PV_X)[i] += alpha * p;
  pvec(G, PV_R)[i] -= alpha * s;
}

// the joint step replaces the block solve: dp = x
__global__ __launch_bounds__(256) void k_pcg_finish(const GraphDev* __restrict__ Gs) {
  const GraphDev G = Gs[blockIdx.z];
  const int i = blockIdx.x * 256 + threadIdx.x, nT = G.T * NB;
  if (i < nT) G.dp[i] = pvec(G, PV_X)[i];
}

static inline unsigned nblk(long long n, int bs) { return (unsigned)((n + bs - 1) / bs); }
static PcgBufs bufs_of(double* const* bufs, int n) {
  PcgBufs B{};
  for (int i = 0; i < n; ++i) B.p[i] = bufs[i];
  return B;
}
static void maxima(const GraphDev* h, int n, int* nT, int* P, int* slots) {
  *nT = *P = *slots = 0;
  for (int i = 0; i < n; ++i) {
    *nT = std::max(*nT, h[i].T * NB);
    *P = std::max(*P, h[i].P);
    *slots = std::max(*slots, h[i].n_slots);
  }
}

void launch_pcg_init(const GraphDev* d, const GraphDev* h, int n, hipStream_t s) {
  int nT, P, slots;
  maxima(h, n, &nT, &P, &slots);
  if (nT > 0) hipLaunchKernelGGL(k_pcg_init, dim3(nblk(nT, 256), 1, n), dim3(256), 0, s, d);
}
// t_l of vector `vec` (PV_U during the iterations) packed into every robot's exchange buffer (9 doubles per slot)
// t_l(v) -> bufs[i][9 slot ..] and, in the same launch, w = S0 v (launch_pcg_matvec_dots continues with the cross-robot part of w)
void launch_pcg_tl(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, int vec, hipStream_t s) {
  int nT, P, slots;
  maxima(h, n, &nT, &P, &slots);
  const int in_lds = (size_t)nT * sizeof(double) <= 48 * 1024 ? 1 : 0;      // (beyond that the vector is read from L2)
  const unsigned nrb = nT / NB, nsl = slots > 0 ? nblk(slots, 4) : 0;
  if (nrb + nsl > 0)
    hipLaunchKernelGGL(k_pcg_tl_symv, dim3(nrb + nsl, 1, n), dim3(256), in_lds ? (size_t)nT * sizeof(double) : 0, s, d, bufs_of(bufs, n), vec, (int)PV_W,
                       (int)nrb, in_lds);
}
// after the exchange of t_l: w = S u (own block + cross-robot fill), then the two partial dot products into bufs[i][0..1]
// after the exchange of t_l (the product w = S0 u ran with t_l, launch_pcg_tl): the cross-robot part of w, then the two partial dot
// products into bufs[i][0..1]
void launch_pcg_matvec_dots(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, int nsum, hipStream_t s) {
  int nT, P, slots;
  maxima(h, n, &nT, &P, &slots);
  const PcgBufs B = bufs_of(bufs, n);
  if (P > 0 && slots > 0) hipLaunchKernelGGL(k_pcg_cross, dim3(nblk(P, 4), 1, n), dim3(256), 0, s, d, B, (int)PV_W, nsum);
  hipLaunchKernelGGL(k_pcg_dots, dim3(1, 1, n), dim3(256), 0, s, d, B);
}
// after the exchange of (gamma, delta): alpha, beta, the four vector updates
void launch_pcg_update(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, int nsum, hipStream_t s) {
  int nT, P, slots;
  maxima(h, n, &nT, &P, &slots);
  hipLaunchKernelGGL(k_pcg_scalars, dim3(n), dim3(64), 0, s, d, bufs_of(bufs, n), nsum);
  if (nT > 0) hipLaunchKernelGGL(k_pcg_update, dim3(nblk(nT, 256), 1, n), dim3(256), 0, s, d);
}
void launch_pcg_finish(const GraphDev* d, const GraphDev* h, int n, hipStream_t s) {
  int nT, P, slots;
  maxima(h, n, &nT, &P, &slots);
  if (nT > 0) hipLaunchKernelGGL(k_pcg_finish, dim3(nblk(nT, 256), 1, n), dim3(256), 0, s, d);
}

}  // namespace sl
