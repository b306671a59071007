// Micro-benchmark: v_mfma_f64_16x16x4_f64 issue rate / dependent latency, DP FMA dependent latency (diagnostic).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma(double* out, unsigned long long* cyc, int iters, int mode) {
  v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-6;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mode == 0) {
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
  } else if (mode == 1) {
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    }
  } else if (mode == 2) {   // result feeds the next B operand (register 0)
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, a0[0], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, a0[0], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, a0[0], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, a0[0], a1, 0, 0, 0);
    }
  } else if (mode == 3) {   // dependent DP FMA chain
    for (int i = 0; i < iters; ++i) {
      x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9); x = __builtin_fma(x, y, 1e-9);
    }
    a0[0] = x;
  } else if (mode == 4) {   // independent DP FMAs
    double x1 = x + 1, x2 = x + 2, x3 = x + 3;
    for (int i = 0; i < iters; ++i) {
      x = __builtin_fma(x, y, 1e-9); x1 = __builtin_fma(x1, y, 1e-9); x2 = __builtin_fma(x2, y, 1e-9); x3 = __builtin_fma(x3, y, 1e-9);
    }
    a0[0] = x + x1 + x2 + x3;
  } else if (mode == 5) {   // dependent rsq chain
    for (int i = 0; i < iters; ++i) {
      x = __builtin_amdgcn_rsq(x + 1.0); x = __builtin_amdgcn_rsq(x + 1.0); x = __builtin_amdgcn_rsq(x + 1.0); x = __builtin_amdgcn_rsq(x + 1.0);
    }
    a0[0] = x;
  } else if (mode == 6) {   // MFMA -> readlane -> VALU -> MFMA round trip
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        int lo = __builtin_amdgcn_readlane(__double2loint(a0[0]), 5), hi = __builtin_amdgcn_readlane(__double2hiint(a0[0]), 5);
        y = __hiloint2double(hi, lo) * 1e-3 + 1.0;
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  v4d s = a0 + a1 + a2 + a3;
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[mode] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 8); hipMalloc(&cyc, 64 * 8); hipMemset(cyc, 0, 64 * 8);
  const int iters = 1000;
  const char* names[] = {"mfma f64 indep x4", "mfma f64 dep accum", "mfma f64 dep via B operand", "fma f64 dependent", "fma f64 indep x4", "rsq f64 dependent(+add)", "mfma->readlane->fma->mfma"};
  for (int grid : {1, 256}) {
    for (int mode = 0; mode < 7; ++mode) {
      hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, out, cyc, iters, mode);
      hipDeviceSynchronize();
    }
    unsigned long long h[64]; hipMemcpy(h, cyc, 64 * 8, hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 7; ++mode) printf("grid %d  %-28s %.1f cycles per op\n", grid, names[mode], (double)h[mode] / (4.0 * iters));
  }
  return 0;
}
