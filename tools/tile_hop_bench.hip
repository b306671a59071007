// Hand-over of a 64x64 f64 TILE (32 KB) between workgroups inside one launch: what the left-looking persistent Cholesky
// (chol_kernels.hip, k_chol_ll) pays per link of its diagonal chain, and which store / fence / load protocol is CORRECT across XCDs.
// A chain of n workgroups (ticket order): workgroup t waits for the flag of t-1, reads t-1's tile, checks every value, writes its own
// tile (values depend on repetition, t and index), publishes its flag.  Variants:
//   store 0: plain stores + release fence (agent)      1: agent-scope atomic stores (write-through) + release fence
//         2: agent-scope atomic stores + s_waitcnt vmcnt(0) only (no L2 write-back instruction)
//   load  0: acquire fence (agent) + plain loads        1: plain loads, no fence      2: agent-scope atomic loads, no fence
//   preread 1: every workgroup reads its predecessor's tile BEFORE waiting (stale lines in its L1 / L2 on purpose)
// Output: ns per hop, split into flag wait -> loads done -> stores + fence done, and the number of stale values seen.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/tile_hop_bench tools/tile_hop_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }
__device__ __forceinline__ double val(int rep, int t, int i) { return (double)(rep * 1000003 + t * 4099 + i); }

__global__ __launch_bounds__(256) void k_tile_hop(double* tiles, int* flags, int n, int rep, int store, int load, int preread, int* ticket,
                                                  unsigned long long* t_out, int* bad_out, int* xcc_out, double* sink) {
  __shared__ int s_t;
  __shared__ double lds[4096];
  if (threadIdx.x == 0) s_t = atomicAdd(ticket, 1);
  __syncthreads();
  const int t = s_t;
  if (t >= n) return;
  const int tid = threadIdx.x;
  double* mine = tiles + (size_t)t * 4096;
  const double* prev = tiles + (size_t)(t > 0 ? t - 1 : 0) * 4096;
  double acc = 0.0;
  if (preread && t > 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += prev[q * 256 + tid];
  }
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  int bad = 0;
  if (t > 0) {
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(flags + (size_t)(t - 1) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != rep + 1) {
        if (++spins > (1 << 22)) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    t0 = __builtin_amdgcn_s_memrealtime();
    if (load == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double v[16];
    if (load == 2) {
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = __hip_atomic_load(prev + q * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = prev[q * 256 + tid];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      lds[q * 256 + tid] = v[q];
      bad += v[q] != val(rep, t - 1, q * 256 + tid);
    }
    __syncthreads();
    t1 = __builtin_amdgcn_s_memrealtime();
  } else {
    t0 = t1 = __builtin_amdgcn_s_memrealtime();
  }
  if (store == 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) mine[q * 256 + tid] = val(rep, t, q * 256 + tid);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) __hip_atomic_store(mine + q * 256 + tid, val(rep, t, q * 256 + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (store == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(flags + (size_t)t * 32, rep + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t2 = __builtin_amdgcn_s_memrealtime();
    t_out[3 * t] = t0; t_out[3 * t + 1] = t1; t_out[3 * t + 2] = t2;
    xcc_out[t] = xcc_id();
  }
  if (bad) atomicAdd(bad_out, bad);
  if (acc == 1.2345e300) sink[0] = acc + lds[tid];
}

int main() {
  const int n = 64;
  double *tiles, *sink;
  int *flags, *ticket, *bad, *xcc;
  unsigned long long* t_out;
  hipMalloc(&tiles, (size_t)n * 4096 * 8);
  hipMalloc(&sink, 8);
  hipMalloc(&flags, n * 128);
  hipMalloc(&ticket, 4);
  hipMalloc(&bad, 4);
  hipMalloc(&xcc, n * 4);
  hipMalloc(&t_out, n * 24);
  hipMemset(tiles, 0, (size_t)n * 4096 * 8);
  hipMemset(flags, 0, n * 128);
  std::vector<unsigned long long> t(3 * n);
  std::vector<int> xc(n);
  const char* sname[] = {"plain + release fence", "atomic wt + release fence", "atomic wt + vmcnt(0)"};
  const char* lname[] = {"acquire fence + plain", "plain (no fence)", "atomic loads"};
  int rep = 0;
  for (int preread = 0; preread < 2; ++preread)
    for (int store = 0; store < 3; ++store)
      for (int load = 0; load < 3; ++load) {
        double hop = 0, w = 0, l = 0, st = 0;
        int nbad = 0, nx = 0, runs = 0;
        for (int r = 0; r < 6; ++r, ++rep) {
          hipMemset(ticket, 0, 4);
          hipMemset(bad, 0, 4);
          hipDeviceSynchronize();
          hipLaunchKernelGGL(k_tile_hop, dim3(n), dim3(256), 0, 0, tiles, flags, n, rep, store, load, preread, ticket, t_out, bad, xcc, sink);
          if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
          int b = 0;
          hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
          hipMemcpy(t.data(), t_out, n * 24, hipMemcpyDeviceToHost);
          hipMemcpy(xc.data(), xcc, n * 4, hipMemcpyDeviceToHost);
          nbad += b;
          if (r == 0) continue;
          ++runs;
          hop += (double)(t[3 * (n - 1) + 2] - t[2]) / (n - 1) * 10.0;
          for (int i = 1; i < n; ++i) {
            w += (double)(t[3 * i] - t[3 * (i - 1) + 2]) * 10.0 / (n - 1);
            l += (double)(t[3 * i + 1] - t[3 * i]) * 10.0 / (n - 1);
            st += (double)(t[3 * i + 2] - t[3 * i + 1]) * 10.0 / (n - 1);
            if (r == 1) nx += xc[i] != xc[i - 1];
          }
        }
        printf("preread %d  store %-26s load %-22s: %7.1f ns per hop (flag %6.1f | loads %6.1f | stores + fence %6.1f)  stale values %d  (%d of %d hops cross XCDs)\n",
               preread, sname[store], lname[load], hop / runs, w / runs, l / runs, st / runs, nbad, nx, n - 1);
        fflush(stdout);
      }
  return 0;
}
