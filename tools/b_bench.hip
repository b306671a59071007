// Microbenchmark of the type-B trailing-update item (rank-128 pass over a 2x1 tile group) in isolation:
//   V0 direct operand loads (the k_chol_step form), V1 operands staged through LDS, plus ablations.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/b_bench.hip -o tools/bin/b_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int NB = 64;
__device__ __forceinline__ v4d mfma_f64(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ long long tri_row(long long t) {
  long long ii = (long long)floor((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (ii * (ii + 1) / 2 > t) --ii;
  while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
  return ii;
}
__device__ __forceinline__ void decode(int tt, int kb, int nP, long long nG, int& i, int& j0) {
  const long long t = nG - 1 - (tt >> 1);
  const long long u = tri_row(t);
  const int v = (int)(t - u * (u + 1) / 2);
  i = kb + 1 + 2 * (nP - 1 - v) + (tt & 1);
  j0 = kb + 1 + 2 * (nP - 1 - (int)u);
}

// ---- V0: direct ------------------------------------------------------------------------------------------------------
template <int KS, int RD, bool NOC, bool NOOP>
__global__ __launch_bounds__(512) void k_v0(double* __restrict__ S, int ld, int kb, int T, int nP, int g0, int g1) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const long long nG = (long long)nP * (nP + 1) / 2;
  for (int g = g0 + blockIdx.x; g < g1; g += gridDim.x) {
    int i, j0;
    decode(g, kb, nP, nG, i, j0);
    const int j = j0 + (wave >> 2);
    if (i > T || j > T - 1 || i < j) continue;
    const int ch = (wave >> 1) & 1, rh = wave & 1;
    const double* pjh = S + (size_t)((kb - 2) * NB) * ld + (size_t)j * NB + 32 * ch + lr;
    const double* pih = S + (size_t)((kb - 2) * NB) * ld + (size_t)i * NB + 32 * rh + lr;
    double* cbh = S + (size_t)(j * NB + 32 * ch + lk) * ld + (size_t)i * NB + 32 * rh + lr;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] = NOC ? 1.0 : cbh[(size_t)(16 * a + 4 * r) * ld + 16 * b];
    double pa[RD][2], pb[RD][2];
#pragma unroll
    for (int pre = 0; pre < RD - 1; ++pre) {
      const size_t off = (size_t)(4 * pre + lk) * ld;
      pa[pre][0] = pjh[off]; pa[pre][1] = pjh[off + 16];
      pb[pre][0] = pih[off]; pb[pre][1] = pih[off + 16];
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + RD - 1 < KS && !NOOP) {
        const size_t off = (size_t)(4 * (ks + RD - 1) + lk) * ld;
        pa[(ks + RD - 1) % RD][0] = pjh[off]; pa[(ks + RD - 1) % RD][1] = pjh[off + 16];
        pb[(ks + RD - 1) % RD][0] = pih[off]; pb[(ks + RD - 1) % RD][1] = pih[off + 16];
      }
      __builtin_amdgcn_sched_barrier(0);
      const int sl = NOOP ? (ks % (RD - 1)) : (ks % RD);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const double na = -pa[sl][a];
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = mfma_f64(na, pb[sl][b], acc[a][b]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!NOC) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) cbh[(size_t)(16 * a + 4 * r) * ld + 16 * b] = acc[a][b][r];
    } else if (acc[0][0][0] == 123.456) {
      cbh[0] = acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3];
    }
  }
}

// ---- V1: operands through LDS ------------------------------------------------------------------------------------------
constexpr int KC = 4;               // k-steps per staged chunk (16 panel columns)
constexpr int RS = 208;             // LDS column stride (doubles): 192 rows + pad, = 32 banks mod 64
template <int KS, bool NOC>
__global__ __launch_bounds__(512) void k_v1(double* __restrict__ S, int ld, int kb, int T, int nP, int g0, int g1) {
  __shared__ double Bs[2][4 * KC][RS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const long long nG = (long long)nP * (nP + 1) / 2;
  typedef double v2d __attribute__((ext_vector_type(2)));
  for (int g = g0 + blockIdx.x; g < g1; g += gridDim.x) {
    int i, j0;
    decode(g, kb, nP, nG, i, j0);
    if (i > T || j0 > T - 1 || i < j0) continue;      // uniform per workgroup
    const int t = wave >> 2, j = j0 + t;
    const bool ok = !(j > T - 1 || i < j);
    const int ch = (wave >> 1) & 1, rh = wave & 1;
    double* cbh = S + (size_t)(j * NB + 32 * ch + lk) * ld + (size_t)i * NB + 32 * rh + lr;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] = (NOC || !ok) ? 1.0 : cbh[(size_t)(16 * a + 4 * r) * ld + 16 * b];
    // staging map: element pair e = tid + 512 q, q = 0..2 : column e / 96, row pair e % 96 of the 192 staged rows
    // (rows 0..63 = tile row i, 64..127 = j0, 128..191 = j0 + 1)
    const double* src[3];
    int dst[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int e = tid + 512 * q, col = e / 96, rp = e % 96, blk = rp >> 5, within = (rp & 31) * 2;
      const int tile = blk == 0 ? i : (j0 + blk - 1);
      src[q] = S + (size_t)((kb - 2) * NB + col) * ld + (size_t)tile * NB + within;
      dst[q] = col * RS + blk * 64 + within;
    }
    v2d st[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) st[q] = *(const v2d*)(src[q]);
    __syncthreads();                      // previous item done with the buffers
#pragma unroll
    for (int q = 0; q < 3; ++q) *(v2d*)(&Bs[0][0][0] + dst[q]) = st[q];
    __syncthreads();
    constexpr int NCH = KS / KC;
    const int arow = 64 * (1 + t) + 32 * ch + lr, brow = 32 * rh + lr;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c + 1 < NCH) {
#pragma unroll
        for (int q = 0; q < 3; ++q) st[q] = *(const v2d*)(src[q] + (size_t)(4 * KC * (c + 1)) * ld);
      }
      const double* B = &Bs[c & 1][0][0];
#pragma unroll
      for (int ks = 0; ks < KC; ++ks) {
        const double a0 = B[(4 * ks + lk) * RS + arow], a1 = B[(4 * ks + lk) * RS + arow + 16];
        const double b0 = B[(4 * ks + lk) * RS + brow], b1 = B[(4 * ks + lk) * RS + brow + 16];
        acc[0][0] = mfma_f64(-a0, b0, acc[0][0]);
        acc[0][1] = mfma_f64(-a0, b1, acc[0][1]);
        acc[1][0] = mfma_f64(-a1, b0, acc[1][0]);
        acc[1][1] = mfma_f64(-a1, b1, acc[1][1]);
      }
      if (c + 1 < NCH) {
#pragma unroll
        for (int q = 0; q < 3; ++q) *(v2d*)(&Bs[(c + 1) & 1][0][0] + dst[q]) = st[q];
      }
      __syncthreads();
    }
    if (!NOC && ok) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) cbh[(size_t)(16 * a + 4 * r) * ld + 16 * b] = acc[a][b][r];
    } else if (acc[0][0][0] == 123.456) {
      cbh[0] = acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3];
    }
  }
}


// ---- V2: software-pipelined across items: operand ring of the next item filled in the tail, C added at the end -----------
template <int KS, int RD>
__global__ __launch_bounds__(512) void k_v2(double* __restrict__ S, int ld, int kb, int T, int nP, int g0, int g1) {
  static_assert(KS % RD == 0, "ring slots must line up across items");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lk = lane >> 4;
  const long long nG = (long long)nP * (nP + 1) / 2;
  const int ch = (wave >> 1) & 1, rh = wave & 1;
  const unsigned lo = (unsigned)(lk * ld + lr);
  auto dec = [&](int g, int& i, int& j, bool& ok) {
    if (g >= g1) { ok = false; i = j = 0; return; }
    int j0;
    decode(g, kb, nP, nG, i, j0);
    j = j0 + (wave >> 2);
    ok = !(i > T || j > T - 1 || i < j);
  };
  double pa[RD][2], pb[RD][2];
  int g = g0 + blockIdx.x, ci, cj; bool cok;
  dec(g, ci, cj, cok);
  if (cok) {
    const double* pju = S + (size_t)((kb - 2) * NB) * ld + (size_t)cj * NB + 32 * ch;
    const double* piu = S + (size_t)((kb - 2) * NB) * ld + (size_t)ci * NB + 32 * rh;
#pragma unroll
    for (int pre = 0; pre < RD - 1; ++pre) {
      const size_t off = (size_t)(4 * pre) * ld;
      pa[pre][0] = (pju + off)[lo]; pa[pre][1] = (pju + off + 16)[lo];
      pb[pre][0] = (piu + off)[lo]; pb[pre][1] = (piu + off + 16)[lo];
    }
  }
  while (g < g1) {
    const int gn = g + gridDim.x;
    int ni, nj; bool nok;
    dec(gn, ni, nj, nok);
    const double* npju = S + (size_t)((kb - 2) * NB) * ld + (size_t)nj * NB + 32 * ch;
    const double* npiu = S + (size_t)((kb - 2) * NB) * ld + (size_t)ni * NB + 32 * rh;
    if (cok) {
      const double* pju = S + (size_t)((kb - 2) * NB) * ld + (size_t)cj * NB + 32 * ch;
      const double* piu = S + (size_t)((kb - 2) * NB) * ld + (size_t)ci * NB + 32 * rh;
      double* cu = S + (size_t)(cj * NB + 32 * ch) * ld + (size_t)ci * NB + 32 * rh;
      v4d cin[2][2], acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
          for (int r = 0; r < 4; ++r) cin[a][b][r] = (cu + (size_t)(16 * a + 4 * r) * ld + 16 * b)[lo];
          acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
        }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int f = ks + RD - 1;
        if (f < KS) {
          const size_t off = (size_t)(4 * f) * ld;
          pa[f % RD][0] = (pju + off)[lo]; pa[f % RD][1] = (pju + off + 16)[lo];
          pb[f % RD][0] = (piu + off)[lo]; pb[f % RD][1] = (piu + off + 16)[lo];
        } else if (nok) {
          const size_t off = (size_t)(4 * (f - KS)) * ld;
          pa[f % RD][0] = (npju + off)[lo]; pa[f % RD][1] = (npju + off + 16)[lo];
          pb[f % RD][0] = (npiu + off)[lo]; pb[f % RD][1] = (npiu + off + 16)[lo];
        }
        __builtin_amdgcn_sched_barrier(0);
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
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) (cu + (size_t)(16 * a + 4 * r) * ld + 16 * b)[lo] = cin[a][b][r] + acc[a][b][r];
    } else if (nok) {
#pragma unroll
      for (int pre = 0; pre < RD - 1; ++pre) {
        const size_t off = (size_t)(4 * pre) * ld;
        pa[pre][0] = (npju + off)[lo]; pa[pre][1] = (npju + off + 16)[lo];
        pb[pre][0] = (npiu + off)[lo]; pb[pre][1] = (npiu + off + 16)[lo];
      }
    }
    g = gn; ci = ni; cj = nj; cok = nok;
  }
}


// ---- V3: V0 with the row <-> lane map permuted so that every access is 16 B per lane (rows 2 lr, 2 lr + 1) --------------
typedef double v2dd __attribute__((ext_vector_type(2)));
__device__ unsigned long long g_clk[2];
__device__ unsigned long long g_span[3] = {~0ull, 0ull, 0ull};   // min start, max start, max end (100 MHz ticks)      // summed shader cycles (s_memtime) and 100 MHz wall ticks (s_memrealtime) of wave 0 of every workgroup
template <int KS, int RD>
__global__ __launch_bounds__(512) void k_v3(double* __restrict__ S, int ld, int kb, int T, int nP, int g0, int g1) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const long long nG = (long long)nP * (nP + 1) / 2;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { atomicMin(&g_span[0], w0); atomicMax(&g_span[1], w0); }
  struct Fin { unsigned long long c0, w0; int tid; __device__ ~Fin() { if (tid == 0) { const unsigned long long w1 = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_clk[0], __builtin_amdgcn_s_memtime() - c0); atomicAdd(&g_clk[1], w1 - w0); atomicMax(&g_span[2], w1); } } } fin{c0, w0, tid};
  for (int g = g0 + blockIdx.x; g < g1; g += gridDim.x) {
    int i, j0;
    decode(g, kb, nP, nG, i, j0);
    const int j = j0 + (wave >> 2);
    if (i > T || j > T - 1 || i < j) continue;
    const int ch = (wave >> 1) & 1, rh = wave & 1;
    const double* pjh = S + (size_t)((kb - 2) * NB + lk) * ld + (size_t)j * NB + 32 * ch + 2 * lr;   // rows 32 ch + 2 lr + a of panel tiles (j, ..)
    const double* pih = S + (size_t)((kb - 2) * NB + lk) * ld + (size_t)i * NB + 32 * rh + 2 * lr;   // rows 32 rh + 2 lr + b of panel tiles (i, ..)
    // acc[a][b] register r of lane (lr, lk) = C[row 32 rh + 2 lr + b][column 32 ch + 2 (lk + 4 r) + a]
    double* cbh = S + (size_t)(j * NB + 32 * ch + 2 * lk) * ld + (size_t)i * NB + 32 * rh + 2 * lr;   // + (8 r + a) ld + b
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const v2dd c2 = *(const v2dd*)(cbh + (size_t)(8 * r + a) * ld);
        acc[a][0][r] = c2[0]; acc[a][1][r] = c2[1];
      }
    v2dd pa[RD], pb[RD];
#pragma unroll
    for (int pre = 0; pre < RD - 1; ++pre) {
      const size_t off = (size_t)(4 * pre) * ld;
      pa[pre] = *(const v2dd*)(pjh + off);
      pb[pre] = *(const v2dd*)(pih + off);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + RD - 1 < KS) {
        const size_t off = (size_t)(4 * (ks + RD - 1)) * ld;
        pa[(ks + RD - 1) % RD] = *(const v2dd*)(pjh + off);
        pb[(ks + RD - 1) % RD] = *(const v2dd*)(pih + off);
      }
      __builtin_amdgcn_sched_barrier(0);
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
        v2dd c2; c2[0] = acc[a][0][r]; c2[1] = acc[a][1][r];
        *(v2dd*)(cbh + (size_t)(8 * r + a) * ld) = c2;
      }
  }
}


// ---- V4: V3 layout + V2 pipelining (next item's operand ring filled in the tail, C added at the end) --------------------
template <int KS, int RD, bool XPF>
__global__ __launch_bounds__(512) void k_v4(double* __restrict__ S, int ld, int kb, int T, int nP, int g0, int g1) {
  static_assert(KS % RD == 0, "ring slots must line up across items");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lk = lane >> 4;
  const long long nG = (long long)nP * (nP + 1) / 2;
  const int ch = (wave >> 1) & 1, rh = wave & 1;
  auto dec = [&](int g, int& i, int& j, bool& ok) {
    if (g >= g1) { ok = false; i = j = 0; return; }
    int j0;
    decode(g, kb, nP, nG, i, j0);
    j = j0 + (wave >> 2);
    ok = !(i > T || j > T - 1 || i < j);
  };
  v2dd pa[RD], pb[RD];
  int g = g0 + blockIdx.x, ci, cj; bool cok;
  dec(g, ci, cj, cok);
  auto pjp = [&](int j) { return S + (size_t)((kb - 2) * NB + lk) * ld + (size_t)j * NB + 32 * ch + 2 * lr; };
  auto pip = [&](int i) { return S + (size_t)((kb - 2) * NB + lk) * ld + (size_t)i * NB + 32 * rh + 2 * lr; };
  if (cok && XPF) {
    const double* pjh = pjp(cj); const double* pih = pip(ci);
#pragma unroll
    for (int pre = 0; pre < RD - 1; ++pre) { pa[pre] = *(const v2dd*)(pjh + (size_t)(4 * pre) * ld); pb[pre] = *(const v2dd*)(pih + (size_t)(4 * pre) * ld); }
  }
  while (g < g1) {
    const int gn = g + gridDim.x;
    int ni, nj; bool nok;
    dec(gn, ni, nj, nok);
    const double* npjh = pjp(nj); const double* npih = pip(ni);
    if (cok) {
      const double* pjh = pjp(cj); const double* pih = pip(ci);
      double* cbh = S + (size_t)(cj * NB + 32 * ch + 2 * lk) * ld + (size_t)ci * NB + 32 * rh + 2 * lr;
      if (!XPF) {
#pragma unroll
        for (int pre = 0; pre < RD - 1; ++pre) { pa[pre] = *(const v2dd*)(pjh + (size_t)(4 * pre) * ld); pb[pre] = *(const v2dd*)(pih + (size_t)(4 * pre) * ld); }
      }
      v2dd cin[2][4];
      v4d acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cin[a][r] = *(const v2dd*)(cbh + (size_t)(8 * r + a) * ld);
        acc[a][0] = v4d{0.0, 0.0, 0.0, 0.0}; acc[a][1] = v4d{0.0, 0.0, 0.0, 0.0};
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int f = ks + RD - 1;
        if (f < KS) {
          pa[f % RD] = *(const v2dd*)(pjh + (size_t)(4 * f) * ld); pb[f % RD] = *(const v2dd*)(pih + (size_t)(4 * f) * ld);
        } else if (XPF && nok) {
          pa[f % RD] = *(const v2dd*)(npjh + (size_t)(4 * (f - KS)) * ld); pb[f % RD] = *(const v2dd*)(npih + (size_t)(4 * (f - KS)) * ld);
        }
        __builtin_amdgcn_sched_barrier(0);
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
          v2dd c2; c2[0] = cin[a][r][0] + acc[a][0][r]; c2[1] = cin[a][r][1] + acc[a][1][r];
          *(v2dd*)(cbh + (size_t)(8 * r + a) * ld) = c2;
        }
    } else if (nok && XPF) {
#pragma unroll
      for (int pre = 0; pre < RD - 1; ++pre) { pa[pre] = *(const v2dd*)(npjh + (size_t)(4 * pre) * ld); pb[pre] = *(const v2dd*)(npih + (size_t)(4 * pre) * ld); }
    }
    g = gn; ci = ni; cj = nj; cok = nok;
  }
}

// ---- V5: V4 with the C loads issued after the last operand refill of the item (next item's operand ring filled in the tail, C added at the end) --------------------
template <int KS, int RD, bool XPF>
__global__ __launch_bounds__(512) void k_v5(double* __restrict__ S, int ld, int kb, int T, int nP, int g0, int g1) {
  static_assert(KS % RD == 0, "ring slots must line up across items");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lk = lane >> 4;
  const long long nG = (long long)nP * (nP + 1) / 2;
  const int ch = (wave >> 1) & 1, rh = wave & 1;
  auto dec = [&](int g, int& i, int& j, bool& ok) {
    if (g >= g1) { ok = false; i = j = 0; return; }
    int j0;
    decode(g, kb, nP, nG, i, j0);
    j = j0 + (wave >> 2);
    ok = !(i > T || j > T - 1 || i < j);
  };
  v2dd pa[RD], pb[RD];
  int g = g0 + blockIdx.x, ci, cj; bool cok;
  dec(g, ci, cj, cok);
  auto pjp = [&](int j) { return S + (size_t)((kb - 2) * NB + lk) * ld + (size_t)j * NB + 32 * ch + 2 * lr; };
  auto pip = [&](int i) { return S + (size_t)((kb - 2) * NB + lk) * ld + (size_t)i * NB + 32 * rh + 2 * lr; };
  if (cok && XPF) {
    const double* pjh = pjp(cj); const double* pih = pip(ci);
#pragma unroll
    for (int pre = 0; pre < RD - 1; ++pre) { pa[pre] = *(const v2dd*)(pjh + (size_t)(4 * pre) * ld); pb[pre] = *(const v2dd*)(pih + (size_t)(4 * pre) * ld); }
  }
  while (g < g1) {
    const int gn = g + gridDim.x;
    int ni, nj; bool nok;
    dec(gn, ni, nj, nok);
    const double* npjh = pjp(nj); const double* npih = pip(ni);
    if (cok) {
      const double* pjh = pjp(cj); const double* pih = pip(ci);
      double* cbh = S + (size_t)(cj * NB + 32 * ch + 2 * lk) * ld + (size_t)ci * NB + 32 * rh + 2 * lr;
      if (!XPF) {
#pragma unroll
        for (int pre = 0; pre < RD - 1; ++pre) { pa[pre] = *(const v2dd*)(pjh + (size_t)(4 * pre) * ld); pb[pre] = *(const v2dd*)(pih + (size_t)(4 * pre) * ld); }
      }
      v2dd cin[2][4];
      v4d acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a) { acc[a][0] = v4d{0.0, 0.0, 0.0, 0.0}; acc[a][1] = v4d{0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int f = ks + RD - 1;
        if (f == KS) {
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) cin[a][r] = *(const v2dd*)(cbh + (size_t)(8 * r + a) * ld);
        }
        if (f < KS) {
          pa[f % RD] = *(const v2dd*)(pjh + (size_t)(4 * f) * ld); pb[f % RD] = *(const v2dd*)(pih + (size_t)(4 * f) * ld);
        } else if (XPF && nok) {
          pa[f % RD] = *(const v2dd*)(npjh + (size_t)(4 * (f - KS)) * ld); pb[f % RD] = *(const v2dd*)(npih + (size_t)(4 * (f - KS)) * ld);
        }
        __builtin_amdgcn_sched_barrier(0);
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
          v2dd c2; c2[0] = cin[a][r][0] + acc[a][0][r]; c2[1] = cin[a][r][1] + acc[a][1][r];
          *(v2dd*)(cbh + (size_t)(8 * r + a) * ld) = c2;
        }
    } else if (nok && XPF) {
#pragma unroll
      for (int pre = 0; pre < RD - 1; ++pre) { pa[pre] = *(const v2dd*)(npjh + (size_t)(4 * pre) * ld); pb[pre] = *(const v2dd*)(npih + (size_t)(4 * pre) * ld); }
    }
    g = gn; ci = ni; cj = nj; cok = nok;
  }
}


// ---- V6: V3 with 1024-thread workgroups (four waves per SIMD): a whole 2x2 group per workgroup ---------------------------
template <int KS, int RD>
__global__ __launch_bounds__(1024) void k_v6(double* __restrict__ S, int ld, int kb, int T, int nP, int g0, int g1) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const long long nG = (long long)nP * (nP + 1) / 2;
  for (int gg = g0 / 2 + blockIdx.x; gg < g1 / 2; gg += gridDim.x) {
    int i, j0;
    decode(2 * gg + ((wave >> 3) & 1), kb, nP, nG, i, j0);
    const int j = j0 + ((wave >> 2) & 1);
    if (i > T || j > T - 1 || i < j) continue;
    const int ch = (wave >> 1) & 1, rh = wave & 1;
    const double* pjh = S + (size_t)((kb - 2) * NB + lk) * ld + (size_t)j * NB + 32 * ch + 2 * lr;
    const double* pih = S + (size_t)((kb - 2) * NB + lk) * ld + (size_t)i * NB + 32 * rh + 2 * lr;
    double* cbh = S + (size_t)(j * NB + 32 * ch + 2 * lk) * ld + (size_t)i * NB + 32 * rh + 2 * lr;
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const v2dd c2 = *(const v2dd*)(cbh + (size_t)(8 * r + a) * ld);
        acc[a][0][r] = c2[0]; acc[a][1][r] = c2[1];
      }
    v2dd pa[RD], pb[RD];
#pragma unroll
    for (int pre = 0; pre < RD - 1; ++pre) {
      pa[pre] = *(const v2dd*)(pjh + (size_t)(4 * pre) * ld);
      pb[pre] = *(const v2dd*)(pih + (size_t)(4 * pre) * ld);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + RD - 1 < KS) {
        pa[(ks + RD - 1) % RD] = *(const v2dd*)(pjh + (size_t)(4 * (ks + RD - 1)) * ld);
        pb[(ks + RD - 1) % RD] = *(const v2dd*)(pih + (size_t)(4 * (ks + RD - 1)) * ld);
      }
      __builtin_amdgcn_sched_barrier(0);
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
        v2dd c2; c2[0] = acc[a][0][r]; c2[1] = acc[a][1][r];
        *(v2dd*)(cbh + (size_t)(8 * r + a) * ld) = c2;
      }
  }
}


// ---- V7: operands through LDS (each panel element fetched once per workgroup), 16-byte LDS reads, cross-item software pipeline:
//          the next item's C and first operand chunk are requested while the current item's last chunk is in the matrix pipe -----
constexpr int KC7 = 4;                  // k-steps per chunk (16 panel columns)
constexpr int RS7 = 192 + 8;            // LDS column stride in doubles (rows 0..63 tile row i, 64..127 j0, 128..191 j0 + 1)
template <int KS>
__global__ __launch_bounds__(512) void k_v7(double* __restrict__ S, int ld, int kb, int T, int nP, int g0, int g1) {
  __shared__ __attribute__((aligned(16))) double Bs[2][4 * KC7][RS7];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lk = lane >> 4;
  const long long nG = (long long)nP * (nP + 1) / 2;
  const int t = wave >> 2, ch = (wave >> 1) & 1, rh = wave & 1;
  constexpr int NCH = KS / KC7;
  // staging map: element pair e = tid + 512 q, q = 0..2: column e / 96, row pair e % 96 of the 192 staged rows
  int scol[3], srow[3], sblk[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) { const int e = tid + 512 * q; scol[q] = e / 96; const int rp = e % 96; sblk[q] = rp >> 5; srow[q] = (rp & 31) * 2; }
  auto item = [&](int g, int& i, int& j0, bool& any) {
    any = false; i = j0 = 0;
    if (g >= g1) return;
    decode(g, kb, nP, nG, i, j0);
    any = !(i > T || j0 > T - 1 || i < j0);
  };
  auto gsrc = [&](int q, int i, int j0, int c) {
    const int tile = sblk[q] == 0 ? i : (j0 + sblk[q] - 1);
    return S + (size_t)((kb - 2) * NB + 4 * KC7 * c + scol[q]) * ld + (size_t)tile * NB + srow[q];
  };
  int g = g0 + blockIdx.x, ci, cj0; bool cany;
  item(g, ci, cj0, cany);
  v2dd st[3], cin[2][4];
  bool cok = false;
  auto load_c = [&](int i, int j0, bool any) {
    const int j = j0 + t;
    cok = any && !(j > T - 1 || i < j);
    if (cok) {
      const double* cbh = S + (size_t)(j * NB + 32 * ch + 2 * lk) * ld + (size_t)i * NB + 32 * rh + 2 * lr;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) cin[a][r] = *(const v2dd*)(cbh + (size_t)(8 * r + a) * ld);
    }
  };
  // prologue of the first item: chunk 0 -> LDS buffer 0
  if (cany) {
#pragma unroll
    for (int q = 0; q < 3; ++q) st[q] = *(const v2dd*)gsrc(q, ci, cj0, 0);
    load_c(ci, cj0, cany);
#pragma unroll
    for (int q = 0; q < 3; ++q) *(v2dd*)(&Bs[0][scol[q]][sblk[q] * 64 + srow[q]]) = st[q];
  }
  __syncthreads();
  int buf = 0;
  while (g < g1) {
    const int gn = g + gridDim.x;
    int ni, nj0; bool nany;
    item(gn, ni, nj0, nany);
    if (cany) {
      const bool my = cok;
      const int j = cj0 + t;
      v4d acc[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a) { acc[a][0] = v4d{0, 0, 0, 0}; acc[a][1] = v4d{0, 0, 0, 0}; }
      v2dd cmine[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) cmine[a][r] = cin[a][r];
      const int arow = 64 * (1 + t) + 32 * ch + 2 * lr, brow = 32 * rh + 2 * lr;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        // request the next chunk (of this item, or the first of the next item together with its C)
        const bool more = c + 1 < NCH;
        if (more) {
#pragma unroll
          for (int q = 0; q < 3; ++q) st[q] = *(const v2dd*)gsrc(q, ci, cj0, c + 1);
        } else if (nany) {
#pragma unroll
          for (int q = 0; q < 3; ++q) st[q] = *(const v2dd*)gsrc(q, ni, nj0, 0);
          load_c(ni, nj0, nany);
        }
        const double* B = &Bs[buf][0][0];
        if (my) {
#pragma unroll
          for (int ks = 0; ks < KC7; ++ks) {
            const v2dd a2 = *(const v2dd*)(B + (4 * ks + lk) * RS7 + arow);
            const v2dd b2 = *(const v2dd*)(B + (4 * ks + lk) * RS7 + brow);
            acc[0][0] = mfma_f64(-a2[0], b2[0], acc[0][0]);
            acc[0][1] = mfma_f64(-a2[0], b2[1], acc[0][1]);
            acc[1][0] = mfma_f64(-a2[1], b2[0], acc[1][0]);
            acc[1][1] = mfma_f64(-a2[1], b2[1], acc[1][1]);
          }
        }
        if (more || nany) {
#pragma unroll
          for (int q = 0; q < 3; ++q) *(v2dd*)(&Bs[buf ^ 1][scol[q]][sblk[q] * 64 + srow[q]]) = st[q];
        }
        __syncthreads();
        buf ^= 1;
      }
      if (my) {
        double* cbh = S + (size_t)(j * NB + 32 * ch + 2 * lk) * ld + (size_t)ci * NB + 32 * rh + 2 * lr;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v2dd c2; c2[0] = cmine[a][r][0] + acc[a][0][r]; c2[1] = cmine[a][r][1] + acc[a][1][r];
            *(v2dd*)(cbh + (size_t)(8 * r + a) * ld) = c2;
          }
      }
    } else if (nany) {
#pragma unroll
      for (int q = 0; q < 3; ++q) st[q] = *(const v2dd*)gsrc(q, ni, nj0, 0);
      load_c(ni, nj0, nany);
#pragma unroll
      for (int q = 0; q < 3; ++q) *(v2dd*)(&Bs[buf][scol[q]][sblk[q] * 64 + srow[q]]) = st[q];
      __syncthreads();
    }
    g = gn; ci = ni; cj0 = nj0; cany = nany;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int T = 59, ld = (T + 1) * NB;
  const size_t n = (size_t)ld * T * NB;
  std::vector<double> h(n);
  unsigned long long s = 12345;
  for (size_t i = 0; i < n; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i] = ((double)(s >> 11) / 9007199254740992.0 - 0.5) * 1e-3; }
  double *S, *S2;
  CK(hipMalloc(&S, n * 8)); CK(hipMalloc(&S2, n * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = argc > 1 ? atoi(argv[1]) : 256;
  for (int kb : {2, 10, 20, 30}) {
    const int nP = (T - kb + 1) / 2;
    const long long nG = (long long)nP * (nP + 1) / 2;
    const int g0 = 0, g1 = (int)nG;      // half of the 2 nG items of the pass
    auto run = [&](const char* name, auto launch, double* out) {
      CK(hipMemcpy(S, h.data(), n * 8, hipMemcpyHostToDevice));
      launch();
      CK(hipDeviceSynchronize());
      if (out) CK(hipMemcpy(out, S, n * 8, hipMemcpyDeviceToDevice));
      float best = 1e9;
      for (int rep = 0; rep < 10; ++rep) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("kb %2d items %4d grid %d  %-22s %8.2f us  (%.2f us per item-round)\n", kb, g1 - g0, grid, name, best * 1e3,
             best * 1e3 / ((g1 - g0 + grid - 1) / grid));
    };
    run("v0 direct RD6", [&] { hipLaunchKernelGGL((k_v0<32, 6, false, false>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, S2);
    run("v0 direct RD10", [&] { hipLaunchKernelGGL((k_v0<32, 10, false, false>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v0 no C", [&] { hipLaunchKernelGGL((k_v0<32, 6, true, false>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v0 no operand loads", [&] { hipLaunchKernelGGL((k_v0<32, 6, false, true>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v0 neither", [&] { hipLaunchKernelGGL((k_v0<32, 6, true, true>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v1 lds", [&] { hipLaunchKernelGGL((k_v1<32, false>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    {   // check v1 == v0 after one application
      std::vector<double> a(n), b(n);
      CK(hipMemcpy(S, h.data(), n * 8, hipMemcpyHostToDevice));
      hipLaunchKernelGGL((k_v1<32, false>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(a.data(), S, n * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), S2, n * 8, hipMemcpyDeviceToHost));
      double md = 0; size_t nd = 0;
      for (size_t i = 0; i < n; ++i) { double d = fabs(a[i] - b[i]); if (d > md) md = d; if (d != 0) ++nd; }
      printf("   v1 vs v0: max |diff| %.3e, %zu differing\n", md, nd);
    }
    run("v2 pipelined RD8", [&] { hipLaunchKernelGGL((k_v2<32, 8>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    {
      std::vector<double> a(n), b(n);
      CK(hipMemcpy(S, h.data(), n * 8, hipMemcpyHostToDevice));
      hipLaunchKernelGGL((k_v2<32, 8>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(a.data(), S, n * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), S2, n * 8, hipMemcpyDeviceToHost));
      double md = 0; size_t nd = 0;
      for (size_t i = 0; i < n; ++i) { double d = fabs(a[i] - b[i]); if (d > md) md = d; if (d != 0) ++nd; }
      printf("   v2 vs v0: max |diff| %.3e, %zu differing\n", md, nd);
    }
    run("v2 pipelined RD16", [&] { hipLaunchKernelGGL((k_v2<32, 16>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v2 pipelined RD4", [&] { hipLaunchKernelGGL((k_v2<32, 4>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    {
      unsigned long long z[2] = {0, 0}, o[2];
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z)));
      { unsigned long long init0[3] = {~0ull, 0ull, 0ull}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_span), init0, sizeof(init0))); }
      hipLaunchKernelGGL((k_v3<32, 6>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1);
      CK(hipDeviceSynchronize());
      CK(hipMemcpyFromSymbol(o, HIP_SYMBOL(g_clk), sizeof(o)));
      unsigned long long sp[3], init[3] = {~0ull, 0ull, 0ull};
      CK(hipMemcpyFromSymbol(sp, HIP_SYMBOL(g_span), sizeof(sp)));
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_span), init, sizeof(init)));
      printf("   v3 spans: last workgroup starts %.2f us after the first, last end %.2f us after the first start\n", (sp[1] - sp[0]) / 100.0, (sp[2] - sp[0]) / 100.0);
      printf("   v3 clocks: %.0f shader cycles per workgroup, %.2f us wall (100 MHz ticks) -> %.2f GHz effective\n", (double)o[0] / grid,
             (double)o[1] / grid / 100.0, (double)o[0] / ((double)o[1] * 10.0));
    }
    run("v3 16B RD6", [&] { hipLaunchKernelGGL((k_v3<32, 6>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    {
      std::vector<double> a(n), b(n);
      CK(hipMemcpy(S, h.data(), n * 8, hipMemcpyHostToDevice));
      hipLaunchKernelGGL((k_v3<32, 6>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(a.data(), S, n * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), S2, n * 8, hipMemcpyDeviceToHost));
      double md = 0; size_t nd = 0;
      for (size_t i = 0; i < n; ++i) { double d = fabs(a[i] - b[i]); if (d > md) md = d; if (d != 0) ++nd; }
      printf("   v3 vs v0: max |diff| %.3e, %zu differing\n", md, nd);
    }
    run("v3 16B RD4", [&] { hipLaunchKernelGGL((k_v3<32, 4>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v3 16B RD10", [&] { hipLaunchKernelGGL((k_v3<32, 10>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v0 direct RD4", [&] { hipLaunchKernelGGL((k_v0<32, 4, false, false>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v4 16B pipelined RD8", [&] { hipLaunchKernelGGL((k_v4<32, 8, true>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v4 16B pipelined RD4", [&] { hipLaunchKernelGGL((k_v4<32, 4, true>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v4 16B C-at-end RD4", [&] { hipLaunchKernelGGL((k_v4<32, 4, false>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v4 16B C-at-end RD8", [&] { hipLaunchKernelGGL((k_v4<32, 8, false>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v5 late C RD8", [&] { hipLaunchKernelGGL((k_v5<32, 8, true>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    {
      std::vector<double> a(n), b(n);
      CK(hipMemcpy(S, h.data(), n * 8, hipMemcpyHostToDevice));
      hipLaunchKernelGGL((k_v5<32, 8, true>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(a.data(), S, n * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), S2, n * 8, hipMemcpyDeviceToHost));
      double md = 0; size_t nd = 0;
      for (size_t i = 0; i < n; ++i) { double d = fabs(a[i] - b[i]); if (d > md) md = d; if (d != 0) ++nd; }
      printf("   v5 vs v0: max |diff| %.3e, %zu differing\n", md, nd);
    }
    run("v5 late C RD4", [&] { hipLaunchKernelGGL((k_v5<32, 4, true>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v5 late C RD16", [&] { hipLaunchKernelGGL((k_v5<32, 16, true>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v6 1024 threads RD4", [&] { hipLaunchKernelGGL((k_v6<32, 4>), dim3(grid), dim3(1024), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    {
      std::vector<double> a(n), b(n);
      CK(hipMemcpy(S, h.data(), n * 8, hipMemcpyHostToDevice));
      hipLaunchKernelGGL((k_v6<32, 4>), dim3(grid), dim3(1024), 0, 0, S, ld, kb, T, nP, g0, g1);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(a.data(), S, n * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), S2, n * 8, hipMemcpyDeviceToHost));
      double md = 0; size_t nd = 0;
      for (size_t i = 0; i < n; ++i) { double d = fabs(a[i] - b[i]); if (d > md) md = d; if (d != 0) ++nd; }
      printf("   v6 vs v0: max |diff| %.3e, %zu differing\n", md, nd);
    }
    run("v6 1024 threads RD6", [&] { hipLaunchKernelGGL((k_v6<32, 6>), dim3(grid), dim3(1024), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    run("v7 lds pipelined", [&] { hipLaunchKernelGGL((k_v7<32>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
    {
      std::vector<double> a(n), b(n);
      CK(hipMemcpy(S, h.data(), n * 8, hipMemcpyHostToDevice));
      hipLaunchKernelGGL((k_v7<32>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(a.data(), S, n * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(b.data(), S2, n * 8, hipMemcpyDeviceToHost));
      double md = 0; size_t nd = 0;
      for (size_t i = 0; i < n; ++i) { double d = fabs(a[i] - b[i]); if (d > md) md = d; if (d != 0) ++nd; }
      printf("   v7 vs v0: max |diff| %.3e, %zu differing\n", md, nd);
    }
    run("v1 lds no C", [&] { hipLaunchKernelGGL((k_v1<32, true>), dim3(grid), dim3(512), 0, 0, S, ld, kb, T, nP, g0, g1); }, nullptr);
  }
  return 0;
}
