// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access pattern of the Cholesky kernels: every lane moves 8 bytes,
// 16 consecutive lanes one 128-byte run, the four 16-lane groups of a wave four runs one matrix column apart.
// Reads (mode 0) or writes (mode 1) exactly `bytes` bytes once; compare with the counter of the same dispatch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(256) void k_read(const double* __restrict__ A, size_t ld, size_t ntile_rows, double* out) {
  // tile (tr, tc) of 64x64 doubles, column-major with leading dimension ld; one workgroup per tile, wave w: 16 columns
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  const size_t tr = blockIdx.x % ntile_rows, tc = blockIdx.x / ntile_rows;
  const double* p = A + (tc * 64 + 16 * wave + lk) * ld + tr * 64 + lr;
  double s = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int b = 0; b < 4; ++b) s += p[(size_t)(4 * r) * ld + 16 * b];
  if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(double* __restrict__ A, size_t ld, size_t ntile_rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  const size_t tr = blockIdx.x % ntile_rows, tc = blockIdx.x / ntile_rows;
  double* p = A + (tc * 64 + 16 * wave + lk) * ld + tr * 64 + lr;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int b = 0; b < 4; ++b) p[(size_t)(4 * r) * ld + 16 * b] = 1.0;
}
int main() {
  const size_t T = 96, ld = T * 64;                 // 6144 x 6144 doubles = 302 MB (past the 256 MB Infinity Cache)
  double *A, *out;
  (void)hipMalloc(&A, ld * ld * 8); (void)hipMalloc(&out, 8);
  (void)hipMemset(A, 0, ld * ld * 8);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(k_read, dim3(T * T), dim3(256), 0, 0, A, ld, T, out);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(k_write, dim3(T * T), dim3(256), 0, 0, A, ld, T);
  (void)hipDeviceSynchronize();
  printf("bytes per kernel: %zu\n", ld * ld * 8);
  return 0;
}
