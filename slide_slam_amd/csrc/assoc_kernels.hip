// Per-frame semantic data association on gfx950 — the "association sweep" of BASELINE.json.
// Reference: *MapManager::getSubmap (backend/sloam/src/core/cubeMapManager.cpp:36-75,
// cylinderMapManager.cpp:213-243, ellipsoidMapManager.cpp:40-80), sloam::projectModels + match*Models
// (src/core/sloam.cpp:73-217), object distances (src/objects/cube.cpp:22-24, ellipsoid.cpp:24-26,
// cylinder.cpp:187-224).
//
// One workgroup per (frame, class).  The K-NN gate is an exact float32 brute-force scan of the first-seen cloud — three SoA
// float streams, read once, coalesced — whose squared distances stay in LDS, followed by a K-SELECT, not a sort of the whole
// cloud: an MSD radix select (8-bit digits, LDS histogram with wave-aggregated atomics) finds the K-th smallest
// (distance bits, map index) key, the K keys at or below it are compacted into LDS and only those are sorted (bitonic, K padded to
// a power of two) — the reference's nearest-first submap order, ties by map index.  The nearest-neighbour match is one wavefront
// per detection with a lexicographic (distance, submap index) shuffle reduction = the reference's strict-'<' first-index-wins rule.
// The cloud may be of any size (beyond the LDS cache the distances are recomputed per select pass); K is bounded by the sort
// buffer (ASSOC_MAX_K).  This file is compiled with -ffp-contract=off: distances are compared against thresholds, so the
// arithmetic must round exactly like the reference's un-fused x86-64 build.
#include <hip/hip_runtime.h>
#include <limits.h>

#include <algorithm>

#include "kernels.hpp"
#include "sl_math.hpp"

namespace sl {

#ifdef SLIDE_STAMPS
// experiment builds (python -m slide_slam_amd.build --stamps): phase time stamps of workgroup 0 (100 MHz wall clock)
__device__ unsigned long long g_assoc_stamps[16];
#define ASTAMP(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) g_assoc_stamps[i] = wall_clock64(); } while (0)
#define ASTAMPW(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_assoc_stamps[i] = wall_clock64(); } while (0)      // wave 0, no barrier
#else
#define ASTAMP(i)
#define ASTAMPW(i)
#endif

__device__ inline double cyl_distance(const double* model, int mlabel, const double* tgt, int tlabel) {
  // Cylinder::distance cylinder.cpp:187-224 (model = map object, tgt = detection)
  if (tlabel != mlabel) return 1000.0;
  double distance = 10000.0;
  const double heights[3] = {0.0, 3.0, 6.0};
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    const double src_t = (heights[h] - model[2]) / model[5];
    const double tgt_t = (heights[h] - tgt[2]) / tgt[5];
    const double dx = (model[0] + src_t * model[3]) - (tgt[0] + tgt_t * tgt[3]);
    const double dy = (model[1] + src_t * model[4]) - (tgt[1] + tgt_t * tgt[4]);
    const double dz = (model[2] + src_t * model[5]) - (tgt[2] + tgt_t * tgt[5]);
    const double dist = sqrt(dx * dx + dy * dy + dz * dz);
    if (dist < distance) distance = dist;
  }
  return distance;
}

struct AssocCore {
  const float *cx, *cy, *cz; const double* model; const int32_t* label; int n; int K;
  int gate;                  // 1: K-NN gate (getSubmap); 0: the submap is the map itself in the caller's order (pure matcher)
  int Kp;                    // sort buffer length: power of two >= min(K, n)   (gate only)
  int cached;                // the n distance words fit in LDS next to the sort buffer
  int staged;                // the K survivors' models + labels fit in LDS (same region as the distance words, which are dead by then)
  double thresh, best_init; int label_gate, is_cyl;
  const double* qpos;        // 3: robot position (double; narrowed to float like PointT)
  const double* det_world;   // cyl: 7 per ; box: xyz taken at stride `det_stride` offset `det_off`
  int det_stride, det_off;
  const int32_t* det_label; int n_det;
  int32_t* match_sub; int32_t* match_map; int32_t* submap; int32_t* n_sub;      // match_sub / submap / n_sub may be null
};

// dynamic LDS: [sel: Kp x 8 B][hist: 256 x 4 B][region: distance cache (n x 4 B, during the select) / the survivors' models
// (Ksub x stride doubles) + labels (Ksub x 4 B) for the matching]
extern __shared__ unsigned long long assoc_lds[];

__device__ __forceinline__ unsigned dist_bits(const AssocCore& C, int i, float qx, float qy, float qz) {
  const float dx = C.cx[i] - qx, dy = C.cy[i] - qy, dz = C.cz[i] - qz;
  float r = dx * dx;
  r += dy * dy;
  r += dz * dz;
  return __float_as_uint(r);       // r >= 0: the bit pattern orders like the value
}

// The K smallest (distance bits << 32 | index) keys of the cloud into sel[0 .. Ksub): ascending when `sorted` (the reference's
// nearest-first submap), else in the order the compaction happened to place them (the matching breaks ties by the keys themselves, so
// it needs the SET only: the bitonic sort is a quarter of a frame's time).  Returns Ksub.
__device__ inline int knn_select(const AssocCore& C, unsigned long long* sel, unsigned* hist, unsigned* dcache, bool sorted) {
  __shared__ unsigned long long s_prefix;
  __shared__ int s_krem, s_stop, s_cnt;
  __shared__ unsigned whist[16 * 256];          // per-wave digit histograms of the radix select
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, nw = nthr >> 6;
  const int n = C.n, Ksub = C.K < n ? C.K : n;
  const float qx = (float)C.qpos[0], qy = (float)C.qpos[1], qz = (float)C.qpos[2];
  const int n_up = (n + nthr - 1) / nthr * nthr;
  if (C.cached)
    for (int i = tid; i < n; i += nthr) dcache[i] = dist_bits(C, i, qx, qy, qz);
  if (tid == 0) { s_prefix = 0ull; s_krem = Ksub; s_stop = 0; s_cnt = 0; }
  __syncthreads();
  ASTAMP(1);
  auto key_at = [&](int i) -> unsigned long long {
    const unsigned b = C.cached ? dcache[i] : dist_bits(C, i, qx, qy, qz);
    return ((unsigned long long)b << 32) | (unsigned)i;
  };
  int shift = 64;                   // selected <=> (key >> shift) <= (prefix >> shift); 64 = everything (K >= n)
  if (Ksub < n) {
    for (int byte = 7; byte >= 0; --byte) {
      shift = 8 * byte;
      // one histogram per wave (no contention between waves; lanes of a wave that hit the same bin are serialised by the LDS
      // atomic unit, ~1 per clock — cheaper than any software aggregation), summed into hist[0 .. 255] afterwards
      for (int b = tid; b < 256 * nw; b += nthr) whist[b] = 0u;
      __syncthreads();
      const unsigned long long prefix = s_prefix;
      unsigned* mine = whist + 256 * (tid >> 6);
      for (int i = tid; i < n; i += nthr) {
        const unsigned long long key = key_at(i);
        if (byte == 7 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&mine[(unsigned)(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      for (int b = tid; b < 256; b += nthr) {
        unsigned t = 0;
        for (int w = 0; w < nw; ++w) t += whist[256 * w + b];
        hist[b] = t;
      }
      __syncthreads();
      if (tid < 64) {
        // bins 4 lane .. 4 lane + 3: the first bin at which the running count reaches the wanted rank
        const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
        const unsigned own = h0 + h1 + h2 + h3;
        unsigned inc = own;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned o = (unsigned)__shfl_up((int)inc, off);
          if (lane >= off) inc += o;
        }
        const unsigned krem = (unsigned)s_krem;
        const unsigned long long reach = __ballot(inc >= krem);
        const int first = __ffsll((long long)reach) - 1;        // exists: the matching keys number at least krem
        if (lane == first) {
          unsigned before = inc - own;
          unsigned d = 4 * lane, cnt = h0;
          if (before + h0 < krem) { before += h0; d += 1; cnt = h1;
            if (before + h1 < krem) { before += h1; d += 1; cnt = h2;
              if (before + h2 < krem) { before += h2; d += 1; cnt = h3; } } }
          s_prefix = prefix | ((unsigned long long)d << shift);
          s_krem = (int)(krem - before);
          s_stop = (before + cnt == krem) ? 1 : 0;      // the whole bin goes: nothing left to decide below this byte
        }
      }
      __syncthreads();
      if (s_stop) break;
    }
  }
  ASTAMP(2);
  // compaction of the selected keys (any order), then the sort that orders them
  {
    const unsigned long long lim = shift < 64 ? (s_prefix >> shift) : 0ull;
    for (int i = tid; i < n_up; i += nthr) {
      unsigned long long key = 0ull;
      bool take = false;
      if (i < n) {
        key = key_at(i);
        take = shift >= 64 || (key >> shift) <= lim;
      }
      const unsigned long long m = __ballot(take);
      if (m) {
        int base = 0;
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) base = atomicAdd(&s_cnt, __popcll(m));
        base = __shfl(base, leader);
        if (take) sel[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
      }
    }
  }
  for (int i = Ksub + tid; i < C.Kp; i += nthr) sel[i] = ~0ull;
  __syncthreads();
  ASTAMP(3);
  if (!sorted) {
    ASTAMP(4);
    return Ksub;
  }
  if (C.Kp <= nthr) {
    // one key per thread: compare-exchange steps inside a wave (partner distance < 64) are two 32-bit shuffles, only the steps
    // across waves go through LDS (10 of the 55 steps of a 1024-key sort)
    unsigned long long v = tid < C.Kp ? sel[tid] : ~0ull;
    for (int k = 2; k <= C.Kp; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        unsigned long long p;
        if (j >= 64) {
          if (tid < C.Kp) sel[tid] = v;
          __syncthreads();
          p = tid < C.Kp ? sel[tid ^ j] : ~0ull;
          __syncthreads();
        } else {
          const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, j), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), j);
          p = ((unsigned long long)hi << 32) | lo;
        }
        const bool keep_min = ((tid & k) == 0) == ((tid & j) == 0);
        v = keep_min ? (v < p ? v : p) : (v > p ? v : p);
      }
    }
    if (tid < C.Kp) sel[tid] = v;
    __syncthreads();
  } else {
    for (int k = 2; k <= C.Kp; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < C.Kp; i += nthr) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long a = sel[i], b = sel[ixj];
            const bool asc = (i & k) == 0;
            if ((a > b) == asc) { sel[i] = b; sel[ixj] = a; }
          }
        }
        __syncthreads();
      }
    }
  }
  ASTAMP(4);
  return Ksub;
}

__device__ inline void assoc_core(const AssocCore& C) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  ASTAMP(0);
  unsigned long long* sel = assoc_lds;
  unsigned* hist = reinterpret_cast<unsigned*>(assoc_lds + (C.gate ? C.Kp : 0));
  unsigned* dcache = hist + 256;
  int Ksub = C.n;
  // the submap ORDER (nearest first, ties by map index) matters to the caller only when it asks for the list or for positions in it;
  // the matching itself needs it for ties alone: "the first of equally distant candidates" = the one with the smallest key
  const bool need_sort = C.submap != nullptr || C.match_sub != nullptr;
  if (C.gate) {
    Ksub = C.n > 0 ? knn_select(C, sel, hist, dcache, need_sort) : 0;
    if (C.submap)
      for (int s = tid; s < Ksub; s += nthr) C.submap[s] = (int32_t)(sel[s] & 0xffffffffull);
  }
  if (tid == 0 && C.n_sub) *C.n_sub = Ksub;
  // the submap's models and labels into LDS, gathered by all threads at once: the matching below would otherwise walk them with
  // one dependent global gather per candidate and lane (latency-bound: ~16 round trips per detection)
  const int ms = C.is_cyl ? 7 : 3;
  double* cand = reinterpret_cast<double*>(dcache);
  int* cand_lab = reinterpret_cast<int*>(cand + (size_t)Ksub * ms);
  const bool staged = C.gate && C.staged;
  if (staged) {
    __syncthreads();                      // (the select's last reads of the distance words are done)
    for (int s = tid; s < Ksub; s += nthr) {
      const int mi = (int)(sel[s] & 0xffffffffull);
      for (int k = 0; k < ms; ++k) cand[(size_t)s * ms + k] = C.model[(size_t)ms * mi + k];
      cand_lab[s] = C.label[mi];
    }
    __syncthreads();
  }
  ASTAMP(5);
  // one wavefront per PAIR of detections (o, o + nwave): the candidates are read once for both.  Boxes: a lane scans its
  // candidates in ascending submap index and keeps the one with the smallest distance d = sqrt(d2), the first on ties.  The
  // correctly rounded f64 sqrt (a long instruction sequence) stays out of the loop: a later candidate replaces the lane's best iff
  // its d is strictly smaller; that is decided on the squared distances when they differ by more than 2^-48 relative (the rounded
  // roots are then distinct), and by the two roots themselves inside that band.  One sqrt per lane at the end feeds the
  // lexicographic (d, index) reduction and the threshold tests, which therefore see exactly the reference's numbers.
  const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  for (int o0 = wave; o0 < C.n_det; o0 += 2 * nwave) {
    const int o1 = o0 + nwave;
    const bool two = o1 < C.n_det;
    const double* dw0 = C.det_world + (size_t)o0 * C.det_stride + C.det_off;
    const double* dw1 = C.det_world + (size_t)(two ? o1 : o0) * C.det_stride + C.det_off;
    const int ol0 = C.det_label[o0], ol1 = C.det_label[two ? o1 : o0];
    double best[2] = {C.best_init, C.best_init};
    int bests[2] = {INT_MAX, INT_MAX};
    unsigned long long bkey[2] = {~0ull, ~0ull};       // order of the candidates: the select's key (gate) or the index itself
    ASTAMPW(8);
    if (C.is_cyl) {
      for (int s = lane; s < Ksub; s += 64) {
        const unsigned long long key = C.gate ? sel[s] : (unsigned long long)s;
        const int mi = C.gate ? (int)(key & 0xffffffffull) : s;
        const double* mm = staged ? cand + (size_t)s * ms : C.model + (size_t)ms * mi;
        const int ml = staged ? cand_lab[s] : C.label[mi];
        const double d0 = cyl_distance(mm, ml, dw0 - C.det_off, ol0);
        if (d0 < C.best_init && (d0 < best[0] || (d0 == best[0] && key < bkey[0]))) { best[0] = d0; bests[0] = s; bkey[0] = key; }
        if (two) {
          const double d1 = cyl_distance(mm, ml, dw1 - C.det_off, ol1);
          if (d1 < C.best_init && (d1 < best[1] || (d1 == best[1] && key < bkey[1]))) { best[1] = d1; bests[1] = s; bkey[1] = key; }
        }
      }
    } else {
      double q0[3], q1[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { q0[k] = dw0[k]; q1[k] = dw1[k]; }
#ifdef SLIDE_STAMPS
      if (q0[0] + q1[0] == 1.2345e300) best[0] = 0.0;      // (forces the loads to complete before the stamp)
#endif
      ASTAMPW(9);
      double b2[2] = {-1.0, -1.0};          // squared distance of the lane's best (none yet: < 0)
      constexpr double BAND = 1.0 - 0x1p-48;
      for (int s = lane; s < Ksub; s += 64) {
        const unsigned long long key = C.gate ? sel[s] : (unsigned long long)s;
        const int mi = C.gate ? (int)(key & 0xffffffffull) : s;
        const double* mm = staged ? cand + (size_t)s * ms : C.model + (size_t)ms * mi;
        const int ml = staged ? cand_lab[s] : C.label[mi];
        const double m0 = mm[0], m1 = mm[1], m2 = mm[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (h == 1 && !two) continue;
          if (C.label_gate == 1 && ml != (h ? ol1 : ol0)) continue;
          const double* q = h ? q1 : q0;
          const double dx = q[0] - m0, dy = q[1] - m1, dz = q[2] - m2;
          const double d2 = dx * dx + dy * dy + dz * dz;
          bool take = b2[h] < 0.0 || d2 < b2[h] * BAND;
          if (!take && d2 * BAND <= b2[h]) {       // inside the band (either side): the rounded roots decide, equal roots the order
            const double rd = sqrt(d2), rb = sqrt(b2[h]);
            take = rd < rb || (rd == rb && key < bkey[h]);
          }
          if (take) { b2[h] = d2; bests[h] = s; bkey[h] = key; }
        }
      }
      ASTAMPW(10);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double d = b2[h] >= 0.0 ? sqrt(b2[h]) : C.best_init;
        if (d < C.best_init) best[h] = d; else { bests[h] = INT_MAX; bkey[h] = ~0ull; }      // "if (d < bestDist)" against the initial bestDist
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double b = best[h];
      int bs = bests[h];
      unsigned long long bk = bkey[h];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(b, off);
        const int os = __shfl_xor(bs, off);
        const unsigned klo = (unsigned)__shfl_xor((int)(unsigned)bk, off), khi = (unsigned)__shfl_xor((int)(unsigned)(bk >> 32), off);
        const unsigned long long ok_ = ((unsigned long long)khi << 32) | klo;
        if (ob < b || (ob == b && ok_ < bk)) { b = ob; bs = os; bk = ok_; }
      }
      const int o = h == 0 ? o0 : o1;
      if (h == 1) ASTAMPW(11);
      if (lane == 0 && (h == 0 || two)) {
        const bool ok = (bs != INT_MAX) && (b < C.thresh);
        if (C.match_sub) C.match_sub[o] = ok ? bs : -1;      // (position in the sorted submap: the list was sorted when this is asked for)
        C.match_map[o] = ok ? (int32_t)(bk & 0xffffffffull) : -1;
      }
    }
  }
  ASTAMP(6);
}
#ifdef SLIDE_STAMPS
extern "C" void slide_debug_assoc_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_assoc_stamps), sizeof(g_assoc_stamps)); }
#endif

// grid = 3 (cylinders, cubes, ellipsoids of ONE key frame)
__global__ __launch_bounds__(1024) void k_assoc_frame(const AssocFrameDev* __restrict__ cls3, const double* __restrict__ pose12) {
  const AssocFrameDev F = cls3[blockIdx.x];
  const int tid = threadIdx.x;
  const SE3 T = from12(pose12);
  // projectModels (sloam.cpp:205-217): body -> world
  for (int o = tid; o < F.n_det; o += blockDim.x) {
    if (F.is_cyl) {
      const double* c = F.det + 7 * (size_t)o;
      double* w = F.det_world + 7 * (size_t)o;
      const V3 root{c[0], c[1], c[2]}, ray{c[3], c[4], c[5]};
      const V3 other = root + ray;                 // Cylinder::project cylinder.cpp:236-242
      const V3 nr = transform_from(T, root), no = transform_from(T, other);
      const V3 nray = no - nr;
      w[0] = nr.x; w[1] = nr.y; w[2] = nr.z; w[3] = nray.x; w[4] = nray.y; w[5] = nray.z; w[6] = c[6];
    } else {
      const SE3 B = from12(F.det + 12 * (size_t)o);
      to12(compose(T, B), F.det_world + 12 * (size_t)o);   // Cube::project cube.cpp:31-36
    }
  }
  __syncthreads();
  AssocCore C;
  C.cx = F.cx; C.cy = F.cy; C.cz = F.cz; C.model = F.model; C.label = F.label; C.n = F.n; C.K = F.K;
  C.gate = F.gate; C.Kp = F.Kp; C.cached = F.cached; C.staged = F.staged;
  C.thresh = F.thresh; C.best_init = F.best_init; C.label_gate = F.label_gate; C.is_cyl = F.is_cyl;
  C.qpos = pose12 + 9;
  C.det_world = F.det_world;
  C.det_stride = F.is_cyl ? 7 : 12;
  C.det_off = F.is_cyl ? 0 : 9;
  C.det_label = F.det_label; C.n_det = F.n_det;
  C.match_sub = F.match_sub; C.match_map = F.match_map; C.submap = F.submap; C.n_sub = F.n_sub;
  assoc_core(C);
}

// Batched sweep: one workgroup per independent query frame against one resident map
// (label-gated points, i.e. the ellipsoid / point-landmark class of the headline graph).
__global__ __launch_bounds__(1024) void k_assoc_sweep(const float* __restrict__ cx, const float* __restrict__ cy, const float* __restrict__ cz,
                                                      const double* __restrict__ model_xyz, const int32_t* __restrict__ label, int n_map,
                                                      const double* __restrict__ query_pos, const double* __restrict__ obs_xyz,
                                                      const int32_t* __restrict__ obs_label, int n_obs, int K, int Kp, int cached,
                                                      int staged, double thresh, int32_t* __restrict__ out_map_idx) {
  const int q = blockIdx.x;
  AssocCore C;
  C.cx = cx; C.cy = cy; C.cz = cz; C.model = model_xyz; C.label = label; C.n = n_map; C.K = K;
  C.gate = 1; C.Kp = Kp; C.cached = cached; C.staged = staged;
  C.thresh = thresh; C.best_init = 1000.0; C.label_gate = 1; C.is_cyl = 0;
  C.qpos = query_pos + 3 * (size_t)q;
  C.det_world = obs_xyz + 3 * (size_t)q * n_obs;
  C.det_stride = 3; C.det_off = 0;
  C.det_label = obs_label + (size_t)q * n_obs; C.n_det = n_obs;
  C.match_sub = nullptr; C.submap = nullptr; C.n_sub = nullptr;
  C.match_map = out_map_idx + (size_t)q * n_obs;
  assoc_core(C);
}

// The same with 512-thread workgroups, two per CU (round 4): the kernel is bound by its in-CU select / match phases,
// every one of which ends at a workgroup barrier — a second resident workgroup fills the other's barrier stalls.  64 KB of LDS each
// (40 KB of distance words / staged models, 8 KB of keys, per-wave histograms), registers capped at 128 by the second bound.
__global__ __launch_bounds__(512, 2) void k_assoc_sweep_512(const float* __restrict__ cx, const float* __restrict__ cy, const float* __restrict__ cz,
                                                            const double* __restrict__ model_xyz, const int32_t* __restrict__ label, int n_map,
                                                            const double* __restrict__ query_pos, const double* __restrict__ obs_xyz,
                                                            const int32_t* __restrict__ obs_label, int n_obs, int K, int Kp, int cached,
                                                            int staged, double thresh, int32_t* __restrict__ out_map_idx) {
  const int q = blockIdx.x;
  AssocCore C;
  C.cx = cx; C.cy = cy; C.cz = cz; C.model = model_xyz; C.label = label; C.n = n_map; C.K = K;
  C.gate = 1; C.Kp = Kp; C.cached = cached; C.staged = staged;
  C.thresh = thresh; C.best_init = 1000.0; C.label_gate = 1; C.is_cyl = 0;
  C.qpos = query_pos + 3 * (size_t)q;
  C.det_world = obs_xyz + 3 * (size_t)q * n_obs;
  C.det_stride = 3; C.det_off = 0;
  C.det_label = obs_label + (size_t)q * n_obs; C.n_det = n_obs;
  C.match_sub = nullptr; C.submap = nullptr; C.n_sub = nullptr;
  C.match_map = out_map_idx + (size_t)q * n_obs;
  assoc_core(C);
}

// updateFactorGraphMap (graphWrapper.cpp:239-275): optimised landmarks -> map models
__global__ void k_map_refresh(double* cyl_model, int n_cyl, const int* cyl_lid, double* cube_xyz, int n_cube,
                              const int* cube_lid, double* ell_xyz, int n_ell, const int* ell_lid, const double* lm_est) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_cyl) {
    const double* e = lm_est + 15 * (size_t)cyl_lid[t];
    for (int k = 0; k < 7; ++k) cyl_model[7 * (size_t)t + k] = e[k];
  } else if (t < n_cyl + n_cube) {
    const int i = t - n_cyl;
    const double* e = lm_est + 15 * (size_t)cube_lid[i];
    for (int k = 0; k < 3; ++k) cube_xyz[3 * (size_t)i + k] = e[9 + k];
  } else if (t < n_cyl + n_cube + n_ell) {
    const int i = t - n_cyl - n_cube;
    const double* e = lm_est + 15 * (size_t)ell_lid[i];
    for (int k = 0; k < 3; ++k) ell_xyz[3 * (size_t)i + k] = e[k];
  }
}

static bool g_attr_set = false;
static void ensure_lds_attr() {
  if (g_attr_set) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_assoc_frame), hipFuncAttributeMaxDynamicSharedMemorySize, ASSOC_LDS_BUDGET);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_assoc_sweep), hipFuncAttributeMaxDynamicSharedMemorySize, ASSOC_LDS_BUDGET);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_assoc_sweep_512), hipFuncAttributeMaxDynamicSharedMemorySize, ASSOC_LDS_BUDGET);
  g_attr_set = true;
}

// LDS plan of one class: sort buffer length, whether the distance words are cached during the select, whether the K survivors'
// models are staged for the matching (the two share one region), bytes.  false: K exceeds the sort buffer.
bool assoc_plan(int n, int K, int gate, int model_stride, int* Kp, int* cached, int* staged, size_t* bytes) {
  *Kp = 0; *cached = 0; *staged = 0; *bytes = 0;
  if (!gate || n <= 0) return true;
  const int Ksub = K < n ? K : n;
  if (Ksub > ASSOC_MAX_K) return false;
  int p = 64;
  while (p < Ksub) p <<= 1;
  *Kp = p;
  const size_t fixed = (size_t)p * 8 + 256 * 4;
  const size_t dc = (size_t)n * 4, st = ((size_t)Ksub * (model_stride * 8 + 4) + 7) / 8 * 8;
  *cached = fixed + dc <= (size_t)ASSOC_LDS_BUDGET ? 1 : 0;
  *staged = fixed + st <= (size_t)ASSOC_LDS_BUDGET ? 1 : 0;
  *bytes = fixed + std::max(*cached ? dc : 0, *staged ? st : 0);
  return true;
}

void launch_assoc_frame(const AssocFrameDev* classes3, size_t lds_bytes, const double* pose12, hipStream_t s) {
  ensure_lds_attr();
  hipLaunchKernelGGL(k_assoc_frame, dim3(3), dim3(1024), lds_bytes, s, classes3, pose12);
}

int launch_assoc_sweep(const float* cx, const float* cy, const float* cz, const double* model_xyz, const int32_t* label, int n_map,
                       const double* query_pos, const double* obs_xyz, const int32_t* obs_label, int n_query, int n_obs,
                       int K, double thresh, int32_t* out_map_idx, hipStream_t s) {
  ensure_lds_attr();
  int Kp, cached, staged;
  size_t bytes;
  if (!assoc_plan(n_map, K, 1, 3, &Kp, &cached, &staged, &bytes)) return -1;
  // two 512-thread workgroups per CU when the plan's LDS lets two fit (the default since round 4: 0.870 -> 0.710 ms per launch of 8192
  // frames against a 10 k-landmark map, 1.40 -> 1.72 TB/s algorithmic; SLIDE_ASSOC_THREADS=1024: one 1024-thread workgroup per CU)
  static const int env_thr = getenv("SLIDE_ASSOC_THREADS") ? atoi(getenv("SLIDE_ASSOC_THREADS")) : 512;
  if (env_thr == 512 && bytes + 20 * 1024 <= 80 * 1024 && Kp <= 1024)
    hipLaunchKernelGGL(k_assoc_sweep_512, dim3(n_query), dim3(512), bytes, s, cx, cy, cz, model_xyz, label, n_map, query_pos, obs_xyz,
                       obs_label, n_obs, K, Kp, cached, staged, thresh, out_map_idx);
  else
    hipLaunchKernelGGL(k_assoc_sweep, dim3(n_query), dim3(1024), bytes, s, cx, cy, cz, model_xyz, label, n_map, query_pos, obs_xyz,
                       obs_label, n_obs, K, Kp, cached, staged, thresh, out_map_idx);
  return 0;
}

void launch_map_refresh(double* cyl_model, int n_cyl, const int* cyl_lid, double* cube_xyz, int n_cube, const int* cube_lid,
                        double* ell_xyz, int n_ell, const int* ell_lid, const double* lm_est, hipStream_t s) {
  const int n = n_cyl + n_cube + n_ell;
  if (n == 0) return;
  hipLaunchKernelGGL(k_map_refresh, dim3((n + 255) / 256), dim3(256), 0, s, cyl_model, n_cyl, cyl_lid, cube_xyz, n_cube,
                     cube_lid, ell_xyz, n_ell, ell_lid, lm_est);
}

}  // namespace sl
