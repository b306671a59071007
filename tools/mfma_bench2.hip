// Clean issue-rate micro-benchmark (inline asm, no compiler copies): v_mfma_f64_16x16x4_f64 and v_fma_f64, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int mode>
__global__ __launch_bounds__(512) void k_bench(double* out, unsigned long long* cyc, int iters) {
  v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-6, z0 = x, z1 = x, z2 = x, z3 = x;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
   for (int u = 0; u < 8; ++u) {
    if (mode == 0)
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n v_mfma_f64_16x16x4_f64 %3, %4, %5, %3"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));
    else if (mode == 1)
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n v_mfma_f64_16x16x4_f64 %0, %1, %2, %0"
                   : "+v"(a0) : "v"(x), "v"(y));
    else if (mode == 2)
      asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                   : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3) : "v"(y), "v"(x));
    else if (mode == 3)
      asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2"
                   : "+v"(z0) : "v"(y), "v"(x));
    else if (mode == 4)
      asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                   : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3) : "v"(y));
    else if (mode == 5)
      asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                   : "+v"(*(float*)&z0), "+v"(*(float*)&z1), "+v"(*(float*)&z2), "+v"(*(float*)&z3) : "v"((float)y), "v"((float)x));
  }}
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  v4d s = a0 + a1 + a2 + a3;
  out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + z0 + z1 + z2 + z3;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[mode] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 1024 * 512 * 8); (void)hipMalloc(&cyc, 64 * 8); (void)hipMemset(cyc, 0, 64 * 8);
  const int iters = 4000;
  const char* names[] = {"mfma f64 4 indep accum", "mfma f64 dependent accum", "v_fma_f64 4 indep", "v_fma_f64 dependent", "v_mul/add_f64 indep", "v_fma_f32 4 indep"};
  for (int threads : {256, 512}) {
    for (int mode = 0; mode < 6; ++mode) {
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, 0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_bench<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
        case 1: hipLaunchKernelGGL(k_bench<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
        case 2: hipLaunchKernelGGL(k_bench<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
        case 3: hipLaunchKernelGGL(k_bench<3>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
        case 4: hipLaunchKernelGGL(k_bench<4>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
        case 5: hipLaunchKernelGGL(k_bench<5>, dim3(256), dim3(threads), 0, 0, out, cyc, iters); break;
      }
      (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[64]; (void)hipMemcpy(h, cyc, 64 * 8, hipMemcpyDeviceToHost);
      printf("waves/SIMD %d  %-26s %.1f ticks per instr (per wave), wall %.3f ms -> %.2f ns per instr\n", threads / 256, names[mode], (double)h[mode] / (32.0 * iters), ms, ms * 1e6 / (32.0 * iters));
    }
  }
  return 0;
}
