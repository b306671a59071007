// Times factor64_mfma (the in-register 64x64 diagonal factorisation of the Cholesky step kernel) in isolation.
#include "../slide_slam_amd/csrc/chol_kernels.hip"
#include <stdio.h>
using namespace sl;
__global__ __launch_bounds__(256) void k_fb(double* out, unsigned long long* cyc, int reps) {
  __shared__ double Wi[16][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  v4d Lt[10];
  unsigned long long tot = 0;
  double acc = 0;
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * I + lr, c = 16 * J + lk + 4 * r;
          Lt[tidx(I, J)][r] = (i == c) ? 70.0 + i : 1.0 / (1.0 + i + c) + rep * 1e-9;   // diagonally dominant SPD
        }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 10; ++t) asm volatile("" : "+v"(Lt[t]));
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    const bool bad = factor64_mfma(Lt, Wi + 4 * wave, nullptr, lr, lk);
#pragma unroll
    for (int t = 0; t < 10; ++t) asm volatile("" : "+v"(Lt[t]));
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    tot += t1 - t0;
#pragma unroll
    for (int t = 0; t < 10; ++t) acc += Lt[t][0] + Lt[t][3];
    acc += bad;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc + Wi[0][threadIdx.x];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = tot / reps;
}
int main() {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&cyc, 64);
  hipLaunchKernelGGL(k_fb, dim3(1), dim3(256), 0, 0, out, cyc, 20);
  (void)hipDeviceSynchronize();
  unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  double o; (void)hipMemcpy(&o, out, 8, hipMemcpyDeviceToHost);
  printf("factor64_mfma: %llu ticks per call (check %g)\n", h, o);
  return 0;
}
