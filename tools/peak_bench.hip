// Sustained v_mfma_f64_16x16x4_f64 rate, wall clock: grid x 512 threads, every wave issues N MFMAs on four accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k_peak(double* out, int n) {
  v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = threadIdx.x * 1e-9, y = 1.0 + threadIdx.x * 1e-12;
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
  }
  if (a0[0] + a1[1] + a2[2] + a3[3] == 123.0) out[0] = 1.0;
}
int main() {
  double* d; hipMalloc(&d, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {256, 512}) for (int grid : {256, 512, 1024}) for (int n : {64, 256, 2048}) {
    hipLaunchKernelGGL(k_peak, dim3(grid), dim3(threads), 0, 0, d, n);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0); hipLaunchKernelGGL(k_peak, dim3(grid), dim3(threads), 0, 0, d, n); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double mf = (double)grid * (threads / 64) * n * 4.0;      // MFMAs
    printf("threads %d grid %4d n %5d: %9.2f us  %.1f TFLOP/s  (%.1f ns per MFMA per SIMD at full occupancy of 256 CUs)\n", threads, grid, n, best * 1e3,
           mf * 2048.0 / (best * 1e-3) / 1e12, best * 1e6 / (mf / 1024.0));
  }
  return 0;
}
