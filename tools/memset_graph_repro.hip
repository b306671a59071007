// Stand-alone check of DESIGN §4 finding 6: a hipMemsetAsync of a few words captured as the FIRST node of a stream capture, the graph
// replayed many times.  Every replay's kernel logs the words it finds (they must be zero: the memset ran before it) and then dirties them.
//   hipcc --offload-arch=gfx950 -O2 tools/memset_graph_repro.hip -o /tmp/memset_graph_repro && /tmp/memset_graph_repro
// Variants: small (32 B) and larger (4 KB) fills, memset first / behind a kernel node, blocking / non-blocking stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
__global__ void k_log_and_dirty(int* words, int n, int* log, int* iter) {
  const int it = *iter;
  int bad = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (words[i] != 0) bad = 1;
    words[i] = 0x5a5a0000 + it;
  }
  if (__syncthreads_or(bad) && threadIdx.x == 0) log[it] = 1;
  __syncthreads();
  if (threadIdx.x == 0) *iter = it + 1;
}
__global__ void k_nop(int* p) { if (threadIdx.x == 1024) *p = 1; }
static int run(size_t bytes, bool memset_first, unsigned flags, int replays) {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, flags));
  int *words, *log, *iter, *dummy;
  CK(hipMalloc(&words, bytes)); CK(hipMalloc(&log, replays * sizeof(int))); CK(hipMalloc(&iter, sizeof(int))); CK(hipMalloc(&dummy, sizeof(int)));
  CK(hipMemset(words, 0xff, bytes)); CK(hipMemset(log, 0, replays * sizeof(int))); CK(hipMemset(iter, 0, sizeof(int)));
  CK(hipDeviceSynchronize());
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  if (!memset_first) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, dummy);
  CK(hipMemsetAsync(words, 0, bytes, s));
  hipLaunchKernelGGL(k_log_and_dirty, dim3(1), dim3(256), 0, s, words, (int)(bytes / 4), log, iter);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  std::vector<int> h(replays);
  CK(hipMemcpy(h.data(), log, replays * sizeof(int), hipMemcpyDeviceToHost));
  int bad = 0, first = -1;
  for (int i = 0; i < replays; ++i) if (h[i]) { ++bad; if (first < 0) first = i; }
  printf("bytes %6zu  memset %s  stream %s: %d of %d replays found the words NOT cleared (first: %d)\n", bytes, memset_first ? "first node " : "second node",
         flags ? "non-blocking" : "blocking    ", bad, replays, first);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(words)); CK(hipFree(log)); CK(hipFree(iter)); CK(hipFree(dummy)); CK(hipStreamDestroy(s));
  return bad ? 1 : 0;
}
int main() {
  int rc = 0;
  for (unsigned fl : {0u, (unsigned)hipStreamNonBlocking})
    for (bool first : {true, false})
      for (size_t b : {(size_t)32, (size_t)4096, (size_t)(1 << 20)}) rc |= run(b, first, fl, 200);
  printf(rc ? "ANOMALY REPRODUCED\n" : "all replays cleared\n");
  return 0;
}
