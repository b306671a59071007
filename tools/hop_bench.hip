// Publish -> poll latency of a chain of workgroups: workgroup t waits for the word of t-1, stamps the wall clock, publishes its own.
// Question: is a hop cheaper when every workgroup of the chain sits on ONE XCD (the word can then be served by that XCD's L2), and
// which cache-policy bits of the polling load get that?  Every poll variant falls back to a device-scope atomic load every 8th
// try, so no variant can spin for ever whatever the dispatcher's workgroup -> XCD map is.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/hop_bench tools/hop_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }   // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ unsigned long long ld_sc0(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned long long ld_nt(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned long long ld_sc0_nt(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned long long ld_inv_plain(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("buffer_inv sc0\n global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// mode: 0 agent-scope atomic load; 1 sc0; 2 sc1; 3 nt; 4 sc0 nt; 5 buffer_inv sc0 + plain; 6 workgroup-scope atomic load
// store: 0 agent-scope atomic store; 1 system-scope atomic store; 2 atomic exchange (agent)
__global__ __launch_bounds__(64) void k_hop(unsigned long long* flags, int n, int mode, int store, int xcc_want, int* ticket,
                                            unsigned long long* t_out, int* xcc_out, int* fb_out) {
  const int xcc = xcc_id();
  if (xcc_want >= 0 && xcc != xcc_want) return;
  __shared__ int s_t;
  if (threadIdx.x == 0) s_t = atomicAdd(ticket, 1);
  __syncthreads();
  const int t = s_t;
  if (t >= n) return;
  if (threadIdx.x == 0) {
    int fallback = 0;
    if (t > 0) {
      const unsigned long long* p = flags + (size_t)(t - 1) * 16;      // 128 bytes apart
      int spins = 0;
      for (;;) {
        unsigned long long v;
        const bool fb = (spins & 7) == 7;
        if (mode == 0 || fb) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (mode == 1) v = ld_sc0(p);
        else if (mode == 2) v = ld_sc1(p);
        else if (mode == 3) v = ld_nt(p);
        else if (mode == 4) v = ld_sc0_nt(p);
        else if (mode == 5) v = ld_inv_plain(p);
        else v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v != 0) { fallback = fb && mode != 0; break; }
        if (++spins > (1 << 20)) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    unsigned long long* q = flags + (size_t)t * 16;
    if (store == 0) __hip_atomic_store(q, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (store == 1) __hip_atomic_store(q, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else atomicExch(q, 1ull);
    t_out[t] = now;
    xcc_out[t] = xcc;
    fb_out[t] = fallback;
  }
}

int main() {
  const int n = 48;
  unsigned long long *flags, *t_out;
  int *ticket, *xcc_out, *fb_out;
  hipMalloc(&flags, n * 128);
  hipMalloc(&t_out, n * 8);
  hipMalloc(&ticket, 4);
  hipMalloc(&xcc_out, n * 4);
  hipMalloc(&fb_out, n * 4);
  std::vector<unsigned long long> t(n);
  std::vector<int> xc(n), fb(n);
  const char* mname[] = {"agent atomic", "sc0", "sc1", "nt", "sc0 nt", "buffer_inv sc0 + plain", "workgroup atomic"};
  const char* sname[] = {"agent store", "system store", "atomic exch"};
  for (int xw : {-1, 0, 3}) {
    for (int store = 0; store < 3; ++store) {
      for (int mode = 0; mode < 7; ++mode) {
        double best = 1e30;
        int nfb = 0, nx = 0;
        for (int rep = 0; rep < 6; ++rep) {
          hipMemset(flags, 0, n * 128);
          hipMemset(ticket, 0, 4);
          hipMemset(t_out, 0, n * 8);
          hipDeviceSynchronize();
          hipLaunchKernelGGL(k_hop, dim3(xw < 0 ? n : 8 * n + 64), dim3(64), 0, 0, flags, n, mode, store, xw, ticket, t_out, xcc_out, fb_out);
          if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
          hipMemcpy(t.data(), t_out, n * 8, hipMemcpyDeviceToHost);
          hipMemcpy(xc.data(), xcc_out, n * 4, hipMemcpyDeviceToHost);
          hipMemcpy(fb.data(), fb_out, n * 4, hipMemcpyDeviceToHost);
          if (t[n - 1] == 0) { best = -1; break; }
          const double hop = (double)(t[n - 1] - t[0]) / (n - 1) * 10.0;      // 100 MHz ticks -> ns
          if (rep > 0 && hop < best) {
            best = hop;
            nfb = 0;
            for (int i = 0; i < n; ++i) nfb += fb[i];
            nx = 0;
            for (int i = 1; i < n; ++i) nx += xc[i] != xc[i - 1];
          }
        }
        printf("xcd %2d  %-12s  poll %-24s: %7.1f ns per hop   (%d of %d hops seen by the fallback load, %d hops cross XCDs)\n", xw, sname[store],
               mname[mode], best, nfb, n - 1, nx);
        fflush(stdout);
      }
    }
  }
  return 0;
}
