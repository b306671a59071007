// Microbenchmark of the border product (k_border_syrk of chol_kernels.hip) in isolation: R robots, T column blocks, nbr border tile rows,
// first non-zero column block of row i = 0.75 i T / nbr (the C4 workload's average sum length is ~0.48 T).
//   V0  the product kernel's form: 64x64 tile per 256-thread workgroup, a 32x32 quadrant per wave, operands as 16-byte global loads
//   V0n V0 without the loads in the loop (matrix pipe only)      V0l V0 without the MFMAs (loads only)
//   V1  operands staged through LDS per workgroup (each panel element fetched once per workgroup instead of twice)
//   V2  128x128 tile per 256-thread workgroup (64x64 per wave: sixteen accumulators), operands through LDS
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/syrk_bench.hip -o tools/bin/syrk_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int NB = 64;
__device__ __forceinline__ v4d mfma_f64(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args {
  const double* S; int ld; int T; int nbr; double* bord; int ldb; const int* bfirst; const int* jobs;
  size_t S_stride, bord_stride;     // per robot
};

// ---- V0 ------------------------------------------------------------------------------------------------------------------------
template <int MODE, int RD = 4, int SAME = 0>
__global__ __launch_bounds__(256) void k_v0(Args A) {
  const int j = A.jobs[blockIdx.x];
  int r = j >> 20, ib = (j >> 10) & 1023, jb = j & 1023;
  const int nbr = A.nbr, T = A.T, ld = A.ld, ldb = A.ldb;
  const int c0 = max(A.bfirst[ib], A.bfirst[jb]), c1 = T;
  if (c0 >= c1) return;
  const double* S = A.S + r * A.S_stride;
  const int wq = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int ch = (wq >> 1) & 1, rh = wq & 1;
  if (ib == nbr && rh == 1) return;
  const double* pjh = S + (size_t)(c0 * NB + lk) * ld + (size_t)(T + jb) * NB + 32 * ch + 2 * lr;
  const double* pih = S + (size_t)(c0 * NB + lk) * ld + (size_t)(T + ib) * NB + 32 * rh + 2 * lr;
  if (SAME == 1) {        // every job reads robot 0's first two row panels, from the top: the loads hit L2 (and mostly L1)
    pjh = A.S + (size_t)lk * ld + (size_t)T * NB + 32 * ch + 2 * lr;
    pih = A.S + (size_t)lk * ld + (size_t)(T + 1) * NB + 32 * rh + 2 * lr;
  }
  if (SAME == 2) {        // every job of an XCD (blockIdx % 8) reads the same robot's panels: an XCD's L2 holds what its jobs read
    pjh = A.S + (blockIdx.x % 8) * A.S_stride + (size_t)lk * ld + (size_t)(T + (blockIdx.x / 8) % 4) * NB + 32 * ch + 2 * lr;
    pih = A.S + (blockIdx.x % 8) * A.S_stride + (size_t)lk * ld + (size_t)(T + 4 + (blockIdx.x / 32) % 4) * NB + 32 * rh + 2 * lr;
  }
  long long kstride = 4LL * ld;
  if (SAME == 3) {        // from the last column block down to c0: all jobs of a launch end at T, so jobs started together read the same columns together
    pjh += (size_t)((c1 - c0) * NB - 4) * ld;
    pih += (size_t)((c1 - c0) * NB - 4) * ld;
    kstride = -kstride;
  }
  double* cbh = A.bord + r * A.bord_stride + (size_t)(jb * NB + 32 * ch + 2 * lk) * ldb + (size_t)ib * NB + 32 * rh + 2 * lr;
  v4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v2d c2 = *(const v2d*)(cbh + (size_t)(8 * e + a) * ldb);
      acc[a][0][e] = c2[0]; acc[a][1][e] = c2[1];
    }
  const int KS = (c1 - c0) * 16;
  v2d pa[RD], pb[RD];
#pragma unroll
  for (int pre = 0; pre < RD - 1; ++pre) {
    const long long off = pre * kstride;
    pa[pre] = *(const v2d*)(pjh + off);
    pb[pre] = *(const v2d*)(pih + off);
  }
  pa[RD - 1] = pa[0]; pb[RD - 1] = pb[0];
  for (int ks0 = 0; ks0 < KS; ks0 += RD) {
#pragma unroll
    for (int u = 0; u < RD; ++u) {
      const int ks = ks0 + u;
      if (MODE != 1 && ks + RD - 1 < KS) {
        const long long off = (ks + RD - 1) * kstride;
        pa[(u + RD - 1) % RD] = *(const v2d*)(pjh + off);
        pb[(u + RD - 1) % RD] = *(const v2d*)(pih + off);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 2) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const double na = -pa[u][a];
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = mfma_f64(na, pb[u][b], acc[a][b]);
        }
      } else {
        asm volatile("" : : "v"(pa[u]), "v"(pb[u]));
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

// ---- V1 / V2: operands through LDS ------------------------------------------------------------------------------------------------
// A workgroup of 256 threads owns a (64 MT) x (64 MT) block of border tiles (MT = 1, 2): rows [ib0, ib0 + MT), columns [jb0, jb0 + MT).
// Per chunk of KC = 16 columns of the band, the two panels (64 MT rows x 16 columns each) go global -> registers -> LDS (double
// buffered), every wave computes a (32 MT) x (32 MT) block: MT^2 x 4 accumulators.  LDS layout of a panel: [column][row] (row
// contiguous), so an operand fetch of "rows 2 lr, 2 lr + 1 of column lk" is one 16-byte read.
template <int MT>
struct V12 {
  static constexpr int KC = 16, ROWS = 64 * MT;
  struct Lds { double a[2][KC][ROWS]; double b[2][KC][ROWS]; };
};
template <int MT>
__global__ __launch_bounds__(256) void k_v12(Args A) {
  using Cfg = V12<MT>;
  constexpr int KC = Cfg::KC, ROWS = Cfg::ROWS;
  __shared__ typename Cfg::Lds L;
  const int j = A.jobs[blockIdx.x];
  const int r = j >> 20, ibt = (j >> 10) & 1023, jbt = j & 1023;      // block coordinates in units of MT tiles
  const int nbr = A.nbr, T = A.T, ld = A.ld, ldb = A.ldb;
  const int ib0 = ibt * MT, jb0 = jbt * MT;
  // the sum starts where the block's earliest row pair starts (bfirst is non-decreasing: the first row of the row block, the first of the column block)
  const int c0 = max(A.bfirst[min(ib0, nbr)], A.bfirst[min(jb0, nbr)]), c1 = T;
  if (c0 >= c1) return;
  const double* S = A.S + r * A.S_stride;
  const int tid = threadIdx.x, wq = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int ch = (wq >> 1) & 1, rh = wq & 1;
  // loader: thread t fetches 16-byte pieces: panel rows 2 (t % (ROWS / 2)) .., column t / (ROWS / 2) + s * (256 / (ROWS / 2))
  constexpr int PR = ROWS / 2, CPT = 256 / PR, NL = KC / CPT;       // pieces per column, columns per sweep, loads per thread per panel
  const int l_row = 2 * (tid % PR), l_col = tid / PR;
  const double* ga = S + (size_t)(T + jb0) * NB + l_row;             // "a": the column block's rows (jb)
  const double* gb = S + (size_t)(T + ib0) * NB + l_row;             // "b": the row block's rows (ib)
  v2d ra[NL], rb[NL];
  auto fetch = [&](int c) {            // chunk c: columns [c KC, (c + 1) KC) counted from column block c0
#pragma unroll
    for (int s = 0; s < NL; ++s) {
      const size_t col = (size_t)c0 * NB + (size_t)c * KC + l_col + s * CPT;
      ra[s] = *(const v2d*)(ga + col * ld);
      rb[s] = *(const v2d*)(gb + col * ld);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int s = 0; s < NL; ++s) {
      *(v2d*)&L.a[buf][l_col + s * CPT][l_row] = ra[s];
      *(v2d*)&L.b[buf][l_col + s * CPT][l_row] = rb[s];
    }
  };
  v4d acc[2 * MT][2 * MT];
#pragma unroll
  for (int a = 0; a < 2 * MT; ++a)
#pragma unroll
    for (int b = 0; b < 2 * MT; ++b) acc[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
  const int NC = (c1 - c0) * NB / KC;
  fetch(0);
  stash(0);
  __syncthreads();
  for (int c = 0; c < NC; ++c) {
    const int buf = c & 1;
    if (c + 1 < NC) fetch(c + 1);
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      v2d pa[MT], pb[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        pa[m] = *(const v2d*)&L.a[buf][4 * ks + lk][32 * MT * ch + 32 * m + 2 * lr];
        pb[m] = *(const v2d*)&L.b[buf][4 * ks + lk][32 * MT * rh + 32 * m + 2 * lr];
      }
#pragma unroll
      for (int ma = 0; ma < MT; ++ma)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const double na = -pa[ma][a];
#pragma unroll
          for (int mb = 0; mb < MT; ++mb)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[2 * ma + a][2 * mb + b] = mfma_f64(na, pb[mb][b], acc[2 * ma + a][2 * mb + b]);
        }
    }
    if (c + 1 < NC) stash(buf ^ 1);
    __syncthreads();
  }
  // C: acc[2 ma + a][2 mb + b][e] = element (row 32 MT rh + 32 mb + 2 lr + b, column 32 MT ch + 32 ma + 2 lk + 8 e + a)
  double* cb = A.bord + r * A.bord_stride + (size_t)(jb0 * NB) * ldb + (size_t)ib0 * NB;
#pragma unroll
  for (int ma = 0; ma < MT; ++ma)
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) {
      const int row = 32 * MT * rh + 32 * mb + 2 * lr, colb = 32 * MT * ch + 32 * ma + 2 * lk;
      const int tr = ib0 + row / NB, tc = jb0 + colb / NB;
      if (tr > nbr || tc >= nbr || tr < tc) continue;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          double* p = cb + (size_t)(colb + 8 * e + a) * ldb + row;
          v2d c2 = *(const v2d*)p;
          c2[0] += acc[2 * ma + a][2 * mb][e]; c2[1] += acc[2 * ma + a][2 * mb + 1][e];
          *(v2d*)p = c2;
        }
    }
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 59, nbr = argc > 3 ? atoi(argv[3]) : 20;
  const int reps = 20;
  const int ld = (T + nbr + 1) * NB, ldb = (nbr + 1) * NB;
  const size_t S_stride = (size_t)ld * T * NB, bord_stride = (size_t)ldb * nbr * NB;
  double *S, *bord, *bord0;
  CK(hipMalloc(&S, (R * S_stride + 65536) * sizeof(double)));
  CK(hipMemset(S, 0, (R * S_stride + 65536) * sizeof(double)));
  CK(hipMalloc(&bord, R * bord_stride * sizeof(double)));
  CK(hipMalloc(&bord0, R * bord_stride * sizeof(double)));
  std::vector<double> h(S_stride);
  srand(1);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3;
  {
    // border rows are zero before their first column block (as W^T is), the right-hand-side row tile holds one row
    std::vector<int> bf0(nbr + 1);
    for (int i = 0; i < nbr; ++i) bf0[i] = (int)(0.75 * i * T / nbr);
    bf0[nbr] = 0;
    for (int c = 0; c < T * NB; ++c)
      for (int i = 0; i <= nbr; ++i)
        for (int w = 0; w < NB; ++w)
          if (c / NB < bf0[i] || (i == nbr && w > 0)) h[(size_t)c * ld + (size_t)(T + i) * NB + w] = 0.0;
  }
  for (int r = 0; r < R; ++r) CK(hipMemcpy(S + r * S_stride, h.data(), S_stride * sizeof(double), hipMemcpyHostToDevice));
  CK(hipMemset(bord0, 0, R * bord_stride * sizeof(double)));
  std::vector<int> bf(nbr + 1);
  for (int i = 0; i < nbr; ++i) bf[i] = (int)(0.75 * i * T / nbr);
  bf[nbr] = 0;
  int* d_bf;
  CK(hipMalloc(&d_bf, (nbr + 1) * sizeof(int)));
  CK(hipMemcpy(d_bf, bf.data(), (nbr + 1) * sizeof(int), hipMemcpyHostToDevice));
  // jobs for MT = 1 and MT = 2 (blocks of 2 x 2 tiles; the right-hand-side row is tile row nbr)
  auto make_jobs = [&](int MT, double* flops) {
    std::vector<std::pair<int, int>> jl;
    const int nbt = (nbr + 1 + MT - 1) / MT;
    double fl = 0.0;
    for (int r = 0; r < R; ++r)
      for (int jbt = 0; jbt * MT < nbr; ++jbt)
        for (int ibt = jbt; ibt < nbt; ++ibt) {
          const int K = T - std::max(bf[std::min(ibt * MT, nbr)], bf[std::min(jbt * MT, nbr)]);
          jl.emplace_back(K, r << 20 | ibt << 10 | jbt);
        }
    // useful flops: the lower tiles (diagonal tiles counted whole), right-hand-side rows as one row
    for (int r = 0; r < R; ++r)
      for (int jb = 0; jb < nbr; ++jb)
        for (int ib = jb; ib <= nbr; ++ib) fl += 2.0 * (ib == nbr ? 1 : 64) * 64.0 * 64.0 * (T - std::max(bf[ib], bf[jb]));
    *flops = fl;
    std::stable_sort(jl.begin(), jl.end(), [](auto& a, auto& b) { return a.first > b.first; });
    std::vector<int> codes;
    for (auto& p : jl) codes.push_back(p.second);
    return codes;
  };
  // robot-per-XCD order: position p of the table runs on XCD p % 8; it takes the next (longest) job of robot p % 8 while that robot has
  // jobs left, else of the robot with the most jobs left
  auto xcd_jobs = [&](const std::vector<int>& sorted) {
    std::vector<std::vector<int>> q(R);
    for (int c : sorted) q[c >> 20].push_back(c);
    std::vector<size_t> head(R, 0);
    std::vector<int> out;
    for (size_t p = 0; p < sorted.size(); ++p) {
      int r = (int)(p % 8) % R;
      if (head[r] >= q[r].size()) {
        size_t best = 0;
        for (int t = 0; t < R; ++t)
          if (q[t].size() - head[t] > best) { best = q[t].size() - head[t]; r = t; }
      }
      out.push_back(q[r][head[r]++]);
    }
    return out;
  };
  double fl1, fl2;
  std::vector<int> j1 = make_jobs(1, &fl1), j2 = make_jobs(2, &fl2);
  int *d_j1, *d_j2;
  CK(hipMalloc(&d_j1, j1.size() * sizeof(int))); CK(hipMemcpy(d_j1, j1.data(), j1.size() * sizeof(int), hipMemcpyHostToDevice));
  CK(hipMalloc(&d_j2, j2.size() * sizeof(int))); CK(hipMemcpy(d_j2, j2.data(), j2.size() * sizeof(int), hipMemcpyHostToDevice));
  Args A{S, ld, T, nbr, bord, ldb, d_bf, d_j1, S_stride, bord_stride};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> ref(R * bord_stride), got(R * bord_stride);
  auto run = [&](const char* name, auto launch, bool check) {
    CK(hipMemcpy(bord, bord0, R * bord_stride * sizeof(double), hipMemcpyDeviceToDevice));
    launch();
    CK(hipDeviceSynchronize());
    double err = -1.0;
    if (check) {
      CK(hipMemcpy(got.data(), bord, got.size() * sizeof(double), hipMemcpyDeviceToHost));
      err = 0.0;
      for (int r = 0; r < R; ++r)
        for (int jb = 0; jb < nbr; ++jb)
          for (int ib = jb; ib < nbr; ++ib)       // (tiles of the border proper; the right-hand-side row tile's idle rows differ by design)
            for (int c = 0; c < NB; ++c)
              for (int w = 0; w < NB; ++w) {
                if (ib == jb && w < c) continue;
                const size_t o = r * bord_stride + (size_t)(jb * NB + c) * ldb + (size_t)ib * NB + w;
                err = std::max(err, std::abs(got[o] - ref[o]));
              }
    }
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-28s %8.3f ms  %6.2f TFLOP/s useful  max|diff| %g\n", name, ms, fl1 / ms * 1e-9, err);
    fflush(stdout);
  };
  // reference result: V0
  CK(hipMemcpy(bord, bord0, R * bord_stride * sizeof(double), hipMemcpyDeviceToDevice));
  hipLaunchKernelGGL(k_v0<0>, dim3(j1.size()), dim3(256), 0, 0, A);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ref.data(), bord, ref.size() * sizeof(double), hipMemcpyDeviceToHost));
  printf("R %d T %d nbr %d: %zu jobs (1x1), %zu jobs (2x2), useful %.2f GFLOP\n", R, T, nbr, j1.size(), j2.size(), fl1 * 1e-9);
  for (int lds : {0, 65536}) {
    char nm[64];
    snprintf(nm, sizeof nm, "V0 direct (lds pad %d)", lds);
    run(nm, [&] { A.jobs = d_j1; hipLaunchKernelGGL(k_v0<0>, dim3(j1.size()), dim3(256), lds, 0, A); }, true);
  }
  run("V0n no loads", [&] { A.jobs = d_j1; hipLaunchKernelGGL(k_v0<1>, dim3(j1.size()), dim3(256), 0, 0, A); }, false);
  run("V0l loads only", [&] { A.jobs = d_j1; hipLaunchKernelGGL(k_v0<2>, dim3(j1.size()), dim3(256), 0, 0, A); }, false);
  run("V0 RD=8", [&] { A.jobs = d_j1; hipLaunchKernelGGL((k_v0<0, 8>), dim3(j1.size()), dim3(256), 0, 0, A); }, true);
  run("V0 RD=2", [&] { A.jobs = d_j1; hipLaunchKernelGGL((k_v0<0, 2>), dim3(j1.size()), dim3(256), 0, 0, A); }, true);
  run("V0l loads only RD=8", [&] { A.jobs = d_j1; hipLaunchKernelGGL((k_v0<2, 8>), dim3(j1.size()), dim3(256), 0, 0, A); }, false);
  run("V0l loads only, same panels", [&] { A.jobs = d_j1; hipLaunchKernelGGL((k_v0<2, 4, 1>), dim3(j1.size()), dim3(256), 0, 0, A); }, false);
  run("V0 same panels", [&] { A.jobs = d_j1; hipLaunchKernelGGL((k_v0<0, 4, 1>), dim3(j1.size()), dim3(256), 0, 0, A); }, false);
  run("V0l loads only, XCD-local", [&] { A.jobs = d_j1; hipLaunchKernelGGL((k_v0<2, 4, 2>), dim3(j1.size()), dim3(256), 0, 0, A); }, false);
  run("V0 XCD-local panels", [&] { A.jobs = d_j1; hipLaunchKernelGGL((k_v0<0, 4, 2>), dim3(j1.size()), dim3(256), 0, 0, A); }, false);
  {
    std::vector<int> jx = xcd_jobs(j1);
    int* d_jx;
    CK(hipMalloc(&d_jx, jx.size() * sizeof(int))); CK(hipMemcpy(d_jx, jx.data(), jx.size() * sizeof(int), hipMemcpyHostToDevice));
    run("V0 reverse columns", [&] { A.jobs = d_j1; hipLaunchKernelGGL((k_v0<0, 4, 3>), dim3(j1.size()), dim3(256), 0, 0, A); }, true);
    run("V0 robot per XCD", [&] { A.jobs = d_jx; hipLaunchKernelGGL((k_v0<0, 4, 0>), dim3(j1.size()), dim3(256), 0, 0, A); }, true);
    run("V0 robot per XCD + reverse", [&] { A.jobs = d_jx; hipLaunchKernelGGL((k_v0<0, 4, 3>), dim3(j1.size()), dim3(256), 0, 0, A); }, true);
    run("V0 robot/XCD + rev, RD=8", [&] { A.jobs = d_jx; hipLaunchKernelGGL((k_v0<0, 8, 3>), dim3(j1.size()), dim3(256), 0, 0, A); }, true);
    run("V0l robot/XCD + rev (loads)", [&] { A.jobs = d_jx; hipLaunchKernelGGL((k_v0<2, 4, 3>), dim3(j1.size()), dim3(256), 0, 0, A); }, false);
    for (int lds : {32768, 49152}) {
      char nm[64];
      snprintf(nm, sizeof nm, "V0 robot/XCD + rev, pad %d", lds);
      run(nm, [&] { A.jobs = d_jx; hipLaunchKernelGGL((k_v0<0, 4, 3>), dim3(j1.size()), dim3(256), lds, 0, A); }, true);
    }
  }
  run("V1 LDS 64x64", [&] { A.jobs = d_j1; hipLaunchKernelGGL(k_v12<1>, dim3(j1.size()), dim3(256), 0, 0, A); }, true);
  run("V2 LDS 128x128", [&] { A.jobs = d_j2; hipLaunchKernelGGL(k_v12<2>, dim3(j2.size()), dim3(256), 0, 0, A); }, true);
  return 0;
}
