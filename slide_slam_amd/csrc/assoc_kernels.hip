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
#ifndef SLIDE_STAMP_BLOCK
#define SLIDE_STAMP_BLOCK 0      // the workgroup that stamps (k_assoc_frame: 0 cylinders, 1 cubes, 2 points)
#endif
#define ASTAMP(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == SLIDE_STAMP_BLOCK) g_assoc_stamps[i] = wall_clock64(); } while (0)
#define ASTAMPW(i) do { if (threadIdx.x == 0 && blockIdx.x == SLIDE_STAMP_BLOCK) g_assoc_stamps[i] = wall_clock64(); } while (0)      // wave 0, no barrier
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
//
// Round 4: the K-th key is found by ONE histogram over bins that are LINEAR in the squared distance (bin = floor(r * 1023.5 / r_max):
// float multiplication by a positive constant and the conversion are monotone, so lower bin => smaller key), not by most-significant-
// digit passes over the float's bits — the leading byte of those bits is the exponent, nearly every landmark of a map shares it, and
// the 10 k LDS atomics of such a pass hit two or three addresses and serialise (stamps: 10.9 of a frame's 35 us in the select).  For
// landmarks spread over a plane r is close to uniformly distributed, the atomics scatter over the bins, and the bin that holds the
// K-th key holds a few dozen keys: those go to a candidate list and every candidate counts the candidates below it (its rank) — keys
// in lower bins are selected outright.  A bin with more than ASSOC_CAND_CAP keys (coincident landmarks, a map on a ring around the
// robot) falls back to the digit passes, restricted to that bin.  The compaction counts per wave first and claims its output range
// with one atomic per wave instead of one per 64 keys (each a dependent LDS round trip: 5.4 us of the 35).
constexpr int ASSOC_BINS = 1024, ASSOC_CAND_CAP = 1024;
__device__ inline int knn_select(const AssocCore& C, unsigned long long* sel, unsigned* hist, unsigned* dcache, bool sorted) {
  __shared__ unsigned long long s_prefix;
  __shared__ int s_krem, s_stop, s_cnt, s_ccnt, s_bin, s_before;
  __shared__ unsigned s_wred[16];
  __shared__ unsigned sbuf[16 * 256];           // bin counts (1024) + candidate keys (1024 x 8 B); the fallback's per-wave digit histograms
  unsigned* lhist = sbuf;
  unsigned long long* lcand = reinterpret_cast<unsigned long long*>(sbuf + ASSOC_BINS);
  unsigned* whist = sbuf;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, nw = nthr >> 6;
  const int n = C.n, Ksub = C.K < n ? C.K : n;
  const float qx = (float)C.qpos[0], qy = (float)C.qpos[1], qz = (float)C.qpos[2];
  const int n_up = (n + nthr - 1) / nthr * nthr;
  // distance words (kept in LDS when they fit) and their maximum; ten independent loads in flight per stream (the cloud is L2-resident
  // across the frames of a launch: the scan is a chain of L2 round trips otherwise)
  unsigned bmax = 0u;
  for (int i0 = tid; i0 < n; i0 += 10 * nthr) {       // ten loads in flight per stream (10 k landmarks on 512 threads: two rounds)
    unsigned b[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int i = i0 + u * nthr;
      b[u] = dist_bits(C, i < n ? i : i0, qx, qy, qz);
    }
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int i = i0 + u * nthr;
      if (i < n) {
        if (C.cached) dcache[i] = b[u];
        bmax = b[u] > bmax ? b[u] : bmax;
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)bmax, m); bmax = o > bmax ? o : bmax; }
  if (lane == 0) s_wred[tid >> 6] = bmax;
  for (int b = tid; b < ASSOC_BINS; b += nthr) lhist[b] = 0u;
  if (tid == 0) { s_prefix = 0ull; s_krem = Ksub; s_stop = 0; s_cnt = 0; s_ccnt = 0; s_bin = -1; s_before = 0; }
  __syncthreads();
  ASTAMP(1);
  auto key_at = [&](int i) -> unsigned long long {
    const unsigned b = C.cached ? dcache[i] : dist_bits(C, i, qx, qy, qz);
    return ((unsigned long long)b << 32) | (unsigned)i;
  };
  int shift = 64;                   // selected <=> (key >> shift) <= (prefix >> shift); 64 = everything (K >= n)
  int fbin = -1;                    // the bin of the K-th key; keys in lower bins are selected whatever the digit passes say
  float scale = 0.0f;
  auto bin_of = [&](unsigned long long key) -> int {
    const int b = (int)(__uint_as_float((unsigned)(key >> 32)) * scale);
    return b < ASSOC_BINS - 1 ? b : ASSOC_BINS - 1;
  };
  bool ranked = false;              // the candidates' ranks placed the last keys: nothing left for the compaction but the lower bins
  if (Ksub < n) {
    for (int w = 0; w < nw; ++w) bmax = s_wred[w] > bmax ? s_wred[w] : bmax;
    const float rmax = __uint_as_float(bmax);
    scale = rmax > 0.0f ? ((float)ASSOC_BINS - 0.5f) / rmax : 0.0f;      // (r = NaN or inf: a corrupt cloud — everything lands in one bin, the fallback sorts it out)
    if (!(scale == scale) || scale > 3.0e38f) scale = 0.0f;
    {
      int i = tid;
      for (; i + 3 * nthr < n; i += 4 * nthr) {
        int bb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bb[u] = bin_of(key_at(i + u * nthr));
#pragma unroll
        for (int u = 0; u < 4; ++u) atomicAdd(&lhist[bb[u]], 1u);
      }
      for (; i < n; i += nthr) atomicAdd(&lhist[bin_of(key_at(i))], 1u);
    }
    __syncthreads();
    ASTAMP(12);
    {
      // the bin at which the running count reaches K: bins `per` at a time per thread, wave scan, wave totals through LDS
      const int per = (ASSOC_BINS + nthr - 1) / nthr;
      unsigned own = 0;
      for (int k = 0; k < per; ++k) { const int bb = tid * per + k; if (bb < ASSOC_BINS) own += lhist[bb]; }
      unsigned inc = own;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = (unsigned)__shfl_up((int)inc, off);
        if (lane >= off) inc += o;
      }
      if (lane == 63) s_wred[tid >> 6] = inc;
      __syncthreads();
      unsigned base = 0;
      for (int w = 0; w < (tid >> 6); ++w) base += s_wred[w];
      const unsigned before_t = base + inc - own;
      if (before_t < (unsigned)Ksub && (unsigned)Ksub <= before_t + own) {      // exactly one thread: the counts sum to n >= K
        unsigned before = before_t;
        int bb = tid * per;
        while (before + lhist[bb] < (unsigned)Ksub) { before += lhist[bb]; ++bb; }
        s_bin = bb;
        s_before = (int)before;
        s_krem = Ksub - (int)before;
      }
      __syncthreads();
    }
    ASTAMP(13);
    fbin = s_bin;
    const int before = s_before, krem = s_krem;
    const int ncand = (int)lhist[fbin];
    if (ncand <= ASSOC_CAND_CAP) {
      // ONE pass places the keys of the lower bins in sel[0 .. before) and the K-th key's bin in the candidate list (any order): a
      // thread notes which of its keys go where (bit masks over up to 32 keys), a wave scan of the counts and one atomic per wave
      // and list give every thread its own output range — no dependent LDS round trip per 64 keys.  Then every candidate's rank among
      // the candidates: the krem lowest go to sel[before + rank].
      for (int c0 = 0; c0 < n; c0 += 32 * nthr) {
        unsigned tm = 0u, cm = 0u;
        const int its = min(32, (n - c0 + nthr - 1) / nthr);
#pragma unroll 4
        for (int it = 0; it < its; ++it) {
          const int i = c0 + it * nthr + tid;
          if (i < n) {
            const int b = bin_of(key_at(i));
            tm |= (b < fbin ? 1u : 0u) << it;
            cm |= (b == fbin ? 1u : 0u) << it;
          }
        }
        const int ct = __popc(tm), cc = __popc(cm);
        int st = ct, sc = cc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int o1 = __shfl_up(st, off), o2 = __shfl_up(sc, off);
          if (lane >= off) { st += o1; sc += o2; }
        }
        int bt = 0, bc = 0;
        if (lane == 63) {
          if (st > 0) bt = atomicAdd(&s_cnt, st);
          if (sc > 0) bc = atomicAdd(&s_ccnt, sc);
        }
        bt = __shfl(bt, 63) + st - ct;
        bc = __shfl(bc, 63) + sc - cc;
        while (tm) {
          const int it = __ffs((int)tm) - 1;
          tm &= tm - 1u;
          sel[bt++] = key_at(c0 + it * nthr + tid);
        }
        while (cm) {
          const int it = __ffs((int)cm) - 1;
          cm &= cm - 1u;
          lcand[bc++] = key_at(c0 + it * nthr + tid);
        }
      }
      __syncthreads();
      ASTAMP(14);
      for (int j = tid; j < ncand; j += nthr) {
        const unsigned long long kj = lcand[j];
        int rank = 0;
        for (int i = 0; i < ncand; ++i) rank += lcand[i] < kj ? 1 : 0;
        if (rank < krem) sel[before + rank] = kj;
      }
      ranked = true;
    } else {
      // digit passes over the keys of that bin alone (most significant byte first, as before round 4)
      __syncthreads();
      for (int byte = 7; byte >= 0; --byte) {
        shift = 8 * byte;
        for (int b = tid; b < 256 * nw; b += nthr) whist[b] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        unsigned* mine = whist + 256 * (tid >> 6);
        for (int i = tid; i < n; i += nthr) {
          const unsigned long long key = key_at(i);
          if (bin_of(key) != fbin) continue;
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
          const unsigned kr = (unsigned)s_krem;
          const unsigned long long reach = __ballot(inc >= kr);
          const int first = __ffsll((long long)reach) - 1;        // exists: the matching keys number at least kr
          if (lane == first) {
            unsigned bef = inc - own;
            unsigned d = 4 * lane, cnt = h0;
            if (bef + h0 < kr) { bef += h0; d += 1; cnt = h1;
              if (bef + h1 < kr) { bef += h1; d += 1; cnt = h2;
                if (bef + h2 < kr) { bef += h2; d += 1; cnt = h3; } } }
            s_prefix = prefix | ((unsigned long long)d << shift);
            s_krem = (int)(kr - bef);
            s_stop = (bef + cnt == kr) ? 1 : 0;      // the whole digit goes: nothing left to decide below this byte
          }
        }
        __syncthreads();
        if (s_stop) break;
      }
    }
  }
  ASTAMP(2);
  // what is left to place: everything when K >= n; after digit passes the keys of the lower bins and those of the K-th key's bin at or
  // below the prefix (every wave counts its keys first and claims its range of sel with one atomic)
  if (fbin < 0) {
    for (int i = tid; i < n; i += nthr) sel[i] = key_at(i);
  } else if (!ranked) {
    const unsigned long long lim = shift < 64 ? (s_prefix >> shift) : 0ull;
    auto taken = [&](unsigned long long key) -> bool {
      const int b = bin_of(key);
      if (b != fbin) return b < fbin;
      return (key >> shift) <= lim;
    };
    int mine = 0;
    for (int i = tid; i < n_up; i += nthr) {
      const bool take = i < n && taken(key_at(i));
      mine += __popcll(__ballot(take));                 // (wave-uniform)
    }
    int base = 0;
    if (lane == 0 && mine > 0) base = atomicAdd(&s_cnt, mine);
    base = __shfl(base, 0);
    for (int i = tid; i < n_up; i += nthr) {
      unsigned long long key = 0ull;
      bool take = false;
      if (i < n) {
        key = key_at(i);
        take = taken(key);
      }
      const unsigned long long m = __ballot(take);
      if (take) sel[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
      base += __popcll(m);
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

constexpr int ASSOC_GMAX = 64;     // detections per (frame, class) up to which the staged submap is grouped by label
__device__ inline void assoc_core(const AssocCore& C) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  ASTAMP(0);
  // label groups of the staged submap (see the staging below): a group is named by the FIRST detection that carries its label
  // s_dl: the detections' labels; s_dgrp: a detection's group; s_glab: a group's label; s_gcnt / s_goff: its survivors; s_ng: groups
  __shared__ int s_dl[ASSOC_GMAX], s_dgrp[ASSOC_GMAX], s_glab[ASSOC_GMAX], s_gcnt[ASSOC_GMAX], s_gcur[ASSOC_GMAX], s_goff[ASSOC_GMAX], s_ng;
  const bool grouped = C.gate && C.staged && !C.is_cyl && C.label_gate == 1 && C.n_det <= ASSOC_GMAX && (C.K < C.n ? C.K : C.n) <= 4 * nthr;
  if (grouped && tid < ASSOC_GMAX) { s_dl[tid] = tid < C.n_det ? C.det_label[tid] : 0; s_gcnt[tid] = 0; s_gcur[tid] = 0; }      // (visible after the select's barriers)
  unsigned long long* sel = assoc_lds;
  unsigned* hist = reinterpret_cast<unsigned*>(assoc_lds + (C.gate ? C.Kp : 0));
  unsigned* dcache = hist + 256;
  int Ksub = C.n;
  // the submap ORDER (nearest first, ties by map index) matters to the caller only when it asks for the list or for positions in it;
  // the matching itself needs it for ties alone: "the first of equally distant candidates" = the one with the smallest key
  // positions in the submap asked for WITHOUT the list itself (the per-frame path, round 5): a match's position is the RANK of its key
  // among the K selected ones — counted by its wavefront, nine reads per lane at 557 landmarks — and the bitonic sort (9.2 of that
  // frame kernel's 18.4 us) is not needed
  const bool rank_mode = C.gate && C.match_sub != nullptr && C.submap == nullptr;
  const bool need_sort = C.submap != nullptr;
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
  __shared__ unsigned s_cmax;
  if (tid == 0) s_cmax = 0u;
  if (staged) {
    __syncthreads();                      // (the select's last reads of the distance words are done)
    if (C.is_cyl) {
      for (int s = tid; s < Ksub; s += nthr) {
        const int mi = (int)(sel[s] & 0xffffffffull);
        for (int k = 0; k < ms; ++k) cand[(size_t)s * ms + k] = C.model[(size_t)ms * mi + k];
        cand_lab[s] = C.label[mi];
      }
    } else if (grouped) {
      // boxes / points, label-gated (round 4): the survivors GROUPED BY LABEL — x, y, z relative to the robot as three float arrays
      // and the survivor's position in sel, group after group, so that a detection scans the ~K / (number of labels) survivors that
      // carry its label instead of all K behind a label test (five in six lanes idle on the headline map).  Survivors whose label no
      // detection carries are dropped.  The groups (distinct detection labels, in order of first appearance) are found by wave 0
      // with lane-to-lane reads while everybody's gathers are in flight; counting and placing go group by group with ballots — a
      // lane per group keeps the wave's count, ONE LDS atomic instruction per wave and pass (a lane per group, distinct addresses).
      float* fx = reinterpret_cast<float*>(dcache);
      float* fy = fx + Ksub;
      float* fz = fx + 2 * (size_t)Ksub;
      int* sidx = reinterpret_cast<int*>(fx + 3 * (size_t)Ksub);
      double* dmx = reinterpret_cast<double*>(fx + 4 * (size_t)Ksub);
      const int lane_ = tid & 63;
      double mx[4], my[4], mz[4];
      int lab[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int s = tid + it * nthr;
        lab[it] = 0;
        mx[it] = my[it] = mz[it] = 0.0;
        if (s < Ksub) {
          const int mi = (int)(sel[s] & 0xffffffffull);
          const double* mm = C.model + 3 * (size_t)mi;
          mx[it] = mm[0]; my[it] = mm[1]; mz[it] = mm[2];
          lab[it] = C.label[mi];
        }
      }
      if (tid < 64) {
        // (lane-to-lane reads with a uniform lane number: v_readlane, no LDS round trip)
        const int mylab = s_dl[lane_];
        const bool in = lane_ < C.n_det;
        bool first = in;
        for (int p = 0; p < C.n_det; ++p) {
          const int lp = __builtin_amdgcn_readlane(mylab, p);
          if (p < lane_ && lp == mylab) first = false;
        }
        const unsigned long long fm = __ballot(first);
        int jj = __popcll(fm & ((1ull << lane_) - 1ull));      // a first detection's group: its rank among the firsts
        if (first) s_glab[jj] = mylab;
        const int jf = jj;
        for (int p = 0; p < C.n_det; ++p) {
          const int lp = __builtin_amdgcn_readlane(mylab, p), jp = __builtin_amdgcn_readlane(jf, p);
          if (((fm >> p) & 1ull) && lp == mylab) jj = jp;
        }
        if (in) s_dgrp[lane_] = jj;
        if (lane_ == 0) s_ng = __popcll(fm);
      }
      __syncthreads();
      ASTAMP(7);
      const int ng = s_ng;
      const int glab_l = lane_ < ng ? s_glab[lane_] : 0;       // lane j: group j's label
      float x[4], y[4], z[4], cm = 0.0f;
      int g[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        g[it] = -1;
        x[it] = (float)(mx[it] - C.qpos[0]); y[it] = (float)(my[it] - C.qpos[1]); z[it] = (float)(mz[it] - C.qpos[2]);
      }
      for (int j = 0; j < ng; ++j) {
        const int gl = __builtin_amdgcn_readlane(glab_l, j);
#pragma unroll
        for (int it = 0; it < 4; ++it)
          if (tid + it * nthr < Ksub && lab[it] == gl) g[it] = j;
      }
#pragma unroll
      for (int it = 0; it < 4; ++it)
        if (g[it] >= 0) cm = fmaxf(cm, fmaxf(fabsf(x[it]), fmaxf(fabsf(y[it]), fabsf(z[it]))));
      // counts: lane j of every wave holds the wave's number of survivors of group j
      int wc[4] = {0, 0, 0, 0};
      for (int j = 0; j < ng; ++j) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if (it * nthr >= Ksub) continue;               // (uniform)
          const int c = __popcll(__ballot(g[it] == j));
          if (lane_ == j) wc[it] = c;
        }
      }
      const int wtot = wc[0] + wc[1] + wc[2] + wc[3];
      if (lane_ < ng && wtot > 0) atomicAdd(&s_gcnt[lane_], wtot);
      __syncthreads();
      ASTAMP(15);
      {
        // every wave scans the group sizes itself (lane j: group j); wave 0 leaves the offsets for the matching
        const int v = lane_ < ng ? s_gcnt[lane_] : 0;
        int inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int o = __shfl_up(inc, off);
          if (lane_ >= off) inc += o;
        }
        if (tid < ng) s_goff[tid] = inc - v;
        int base = inc - v;                             // lane j: where this wave's survivors of group j start
        if (lane_ < ng && wtot > 0) base += atomicAdd(&s_gcur[lane_], wtot);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if (it * nthr >= Ksub) continue;
          int rank = 0;
          for (int j = 0; j < ng; ++j) {
            const unsigned long long m = __ballot(g[it] == j);
            if (g[it] == j) rank = __popcll(m & ((1ull << lane_) - 1ull));
          }
          const int gb = __shfl(base, g[it] >= 0 ? g[it] : 0);
          if (g[it] >= 0) {
            const int pos = gb + rank;
            fx[pos] = x[it]; fy[pos] = y[it]; fz[pos] = z[it];
            sidx[pos] = tid + it * nthr;
            dmx[pos] = mx[it]; dmx[Ksub + pos] = my[it]; dmx[2 * (size_t)Ksub + pos] = mz[it];      // the model itself, for the exact rule
          }
          base += wc[it];                               // (lane j: the next iteration's survivors of group j follow this one's)
        }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) cm = fmaxf(cm, __shfl_xor(cm, m));
      if (lane_ == 0) atomicMax(&s_cmax, __float_as_uint(cm));
    } else {
      // boxes / points: x, y, z RELATIVE TO THE ROBOT as three float arrays, for the screening pass of the matching (a lane reads
      // candidate s = lane + 64 u: consecutive words, no bank conflicts), and the largest coordinate magnitude for its error bound
      float* fx = reinterpret_cast<float*>(dcache);
      float cm = 0.0f;
      for (int s = tid; s < Ksub; s += nthr) {
        const int mi = (int)(sel[s] & 0xffffffffull);
        const double* mm = C.model + 3 * (size_t)mi;
        const float x = (float)(mm[0] - C.qpos[0]), y = (float)(mm[1] - C.qpos[1]), z = (float)(mm[2] - C.qpos[2]);
        fx[s] = x; fx[Ksub + s] = y; fx[2 * (size_t)Ksub + s] = z;
        reinterpret_cast<int*>(fx + 3 * (size_t)Ksub)[s] = C.label[mi];
        cm = fmaxf(cm, fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))));
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) cm = fmaxf(cm, __shfl_xor(cm, m));
      if ((tid & 63) == 0) atomicMax(&s_cmax, __float_as_uint(cm));      // (cm >= 0 or NaN: the bit patterns order like the values)
    }
    __syncthreads();
  }
  ASTAMP(5);
  const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  // lexicographic (distance, key) minimum over the wave = the reference's strict-'<' first-wins rule, then the threshold test
  auto reduce_write = [&](int o, double b, int bs, unsigned long long bk) {
    // (a lane without a candidate holds (best_init, INT_MAX, ~0) and loses to every lane with one: when at most one lane has a
    // candidate — the rule after the screening — there is nothing to reduce)
    const unsigned long long have = __ballot(bs != INT_MAX);
    int w = 0;
    if (__popcll(have) <= 1) {
      w = have ? __ffsll((long long)have) - 1 : 0;
    } else {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(b, off);
        const int os = __shfl_xor(bs, off);
        const unsigned klo = (unsigned)__shfl_xor((int)(unsigned)bk, off), khi = (unsigned)__shfl_xor((int)(unsigned)(bk >> 32), off);
        const unsigned long long ok_ = ((unsigned long long)khi << 32) | klo;
        if (ob < b || (ob == b && ok_ < bk)) { b = ob; bs = os; bk = ok_; }
      }
    }
    if (rank_mode) {
      // lane w's result to every lane; the position in the (never formed) nearest-first list = the keys below the match's
      const double bw = __shfl(b, w);
      const int bsw = __shfl(bs, w);
      const unsigned klo = (unsigned)__shfl((int)(unsigned)bk, w), khi = (unsigned)__shfl((int)(unsigned)(bk >> 32), w);
      const unsigned long long bkw = ((unsigned long long)khi << 32) | klo;
      const bool ok = (bsw != INT_MAX) && (bw < C.thresh);
      int rank = 0;
      if (ok) {
        for (int s = lane; s < Ksub; s += 64) rank += sel[s] < bkw ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) rank += __shfl_xor(rank, off);
      }
      if (lane == 0) {
        C.match_sub[o] = ok ? rank : -1;
        C.match_map[o] = ok ? (int32_t)(bkw & 0xffffffffull) : -1;
      }
      return;
    }
    if (lane == w) {
      const bool ok = (bs != INT_MAX) && (b < C.thresh);
      if (C.match_sub) C.match_sub[o] = ok ? bs : -1;      // (position in the sorted submap: the list was sorted when this is asked for; pure matchers: the index itself)
      C.match_map[o] = ok ? (int32_t)(bk & 0xffffffffull) : -1;
    }
  };
  if (C.is_cyl) {
    // one wavefront per PAIR of detections (o, o + nwave): the candidates are read once for both
    for (int o0 = wave; o0 < C.n_det; o0 += 2 * nwave) {
      const int o1 = o0 + nwave;
      const bool two = o1 < C.n_det;
      const double* dw0 = C.det_world + (size_t)o0 * C.det_stride + C.det_off;
      const double* dw1 = C.det_world + (size_t)(two ? o1 : o0) * C.det_stride + C.det_off;
      const int ol0 = C.det_label[o0], ol1 = C.det_label[two ? o1 : o0];
      double best[2] = {C.best_init, C.best_init};
      int bests[2] = {INT_MAX, INT_MAX};
      unsigned long long bkey[2] = {~0ull, ~0ull};       // order of the candidates: the select's key (gate) or the index itself
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
      reduce_write(o0, best[0], bests[0], bkey[0]);
      if (two) reduce_write(o1, best[1], bests[1], bkey[1]);
    }
  } else {
    // Boxes / points: one wavefront per up to THREE detections (o, o + nwave, o + 2 nwave) — the candidates are read once for all of
    // them, and 20 detections on 8 waves are one scan per wave instead of two.  The reference's rule: the candidate with the smallest
    // distance d = sqrt(d2) in double precision, the first in submap order on ties, label-gated.
    //
    // Staged submap (round 4): the rule is applied to a SCREENED set.  Pass 1 computes every candidate's squared distance in float
    // from coordinates relative to the robot and takes the minimum m over the wave; pass 2 recomputes them (bit-identical) and hands a
    // candidate to the exact double-precision rule iff its float distance is within the error bound of m:
    //   with u = 2^-24, candidates / detection rounded to float with |error| <= u |coordinate|, the float result is
    //   d_f = (|q^ - c^|)(1 + t), |t| <= 3 u, and | |q^ - c^| - |q - c| | <= sqrt(3) u (|q|_inf + |c|_inf) =: E, so every candidate
    //   whose true distance is not above the true minimum (the arg-min and all its ties) has d_f <= (m + 2 E)(1 + 8 u).
    // The test uses twice that E.  The exact rule then sees one or two candidates per detection instead of a sixth of the submap, on
    // the original double-precision models: the decisions (arg-min, tie order, threshold) are the reference's, bit for bit, and the
    // double-precision work — which bounded this phase: 64-bit vector ALU, five in six lanes idle behind the label gate — is gone.
    constexpr int HM = 3;
    constexpr double BAND = 1.0 - 0x1p-48;
    const float* fx = reinterpret_cast<const float*>(dcache);
    const float* fy = fx + Ksub;
    const float* fz = fx + 2 * (size_t)Ksub;
    const int* flab = reinterpret_cast<const int*>(fx + 3 * (size_t)Ksub);
    if (grouped) {
      // label groups (see the staging): a wavefront per detection (HG > 1 interleaves several: spills at the 128 registers of this kernel;
      // interleaved), each over the survivors of ITS label only — float distances of up to 256 of them stay in registers between the
      // two passes
      const int* sidx = flab;
      const double* dmx = reinterpret_cast<const double*>(fx + 4 * (size_t)Ksub);
      constexpr float U = 0x1p-24f;
      const float INF = __uint_as_float(0x7f800000u);
      const float cmax = __uint_as_float(s_cmax);
      constexpr int HG = 1;
      for (int o0 = wave; o0 < C.n_det; o0 += HG * nwave) {
        double q[HG][3];
        float qf[HG][3], d2r[HG][4], mf[HG], T2[HG];
        int lo[HG], cnt[HG];
#pragma unroll
        for (int h = 0; h < HG; ++h) {
          const int o = o0 + h * nwave;
          const bool in = o < C.n_det;
          const double* dw = C.det_world + (size_t)(in ? o : o0) * C.det_stride + C.det_off;
#pragma unroll
          for (int k = 0; k < 3; ++k) { q[h][k] = dw[k]; qf[h][k] = (float)(q[h][k] - C.qpos[k]); }
          const int g = s_dgrp[in ? o : o0];
          lo[h] = s_goff[g];
          cnt[h] = in ? s_gcnt[g] : 0;
          mf[h] = INF;
        }
        ASTAMPW(8);
        ASTAMPW(9);
        auto fd2 = [&](int h, int t) -> float {
          const float dx = qf[h][0] - fx[lo[h] + t], dy = qf[h][1] - fy[lo[h] + t], dz = qf[h][2] - fz[lo[h] + t];
          return dx * dx + dy * dy + dz * dz;
        };
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int h = 0; h < HG; ++h) {
            const int t = lane + 64 * u;
            d2r[h][u] = t < cnt[h] ? fd2(h, t) : INF;
            mf[h] = fminf(mf[h], d2r[h][u]);
          }
#pragma unroll
        for (int h = 0; h < HG; ++h)
          for (int t = lane + 256; t < cnt[h]; t += 64) mf[h] = fminf(mf[h], fd2(h, t));
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
          for (int h = 0; h < HG; ++h) mf[h] = fminf(mf[h], __shfl_xor(mf[h], m));
        double b2[HG];                        // squared distance of the lane's best (none yet: < 0)
        int bests[HG];
        unsigned long long bkey[HG];
#pragma unroll
        for (int h = 0; h < HG; ++h) {
          const float qm = fmaxf(fabsf(qf[h][0]), fmaxf(fabsf(qf[h][1]), fabsf(qf[h][2])));
          const float E = 4.0f * U * (qm + cmax);
          const float T = (sqrtf(mf[h]) + 2.0f * E) * (1.0f + 8.0f * U);
          T2[h] = mf[h] < INF ? T * T * (1.0f + 4.0f * U) : -1.0f;      // (no survivor of that label: nothing passes)
          b2[h] = -1.0; bests[h] = INT_MAX; bkey[h] = ~0ull;
        }
        auto exact1 = [&](int h, int t) {     // the exact rule (see below) on survivor lo + t
          const int s = sidx[lo[h] + t];
          const unsigned long long key = sel[s];
          const double dx = q[h][0] - dmx[lo[h] + t], dy = q[h][1] - dmx[Ksub + lo[h] + t], dz = q[h][2] - dmx[2 * (size_t)Ksub + lo[h] + t];
          const double d2 = dx * dx + dy * dy + dz * dz;
          bool take = b2[h] < 0.0 || d2 < b2[h] * BAND;
          if (!take && d2 * BAND <= b2[h]) {
            const double rd = sqrt(d2), rb = sqrt(b2[h]);
            take = rd < rb || (rd == rb && key < bkey[h]);
          }
          if (take) { b2[h] = d2; bests[h] = s; bkey[h] = key; }
        };
#pragma unroll
        for (int h = 0; h < HG; ++h) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (d2r[h][u] <= T2[h]) exact1(h, lane + 64 * u);
          for (int t = lane + 256; t < cnt[h]; t += 64)
            if (fd2(h, t) <= T2[h]) exact1(h, t);
        }
        ASTAMPW(10);
#pragma unroll
        for (int h = 0; h < HG; ++h) {
          const int o = o0 + h * nwave;
          if (o >= C.n_det) continue;
          double best = C.best_init;
          const double d = b2[h] >= 0.0 ? sqrt(b2[h]) : C.best_init;
          if (d < C.best_init) best = d; else { bests[h] = INT_MAX; bkey[h] = ~0ull; }      // "if (d < bestDist)" against the initial bestDist
          ASTAMPW(11);
          reduce_write(o, best, bests[h], bkey[h]);
        }
      }
    } else
    for (int o0 = wave; o0 < C.n_det; o0 += HM * nwave) {
      int nh = 0, ol[HM];
      double q[HM][3];
#pragma unroll
      for (int h = 0; h < HM; ++h) {
        const int o = o0 + h * nwave;
        const bool in = o < C.n_det;
        if (in) nh = h + 1;
        const double* dw = C.det_world + (size_t)(in ? o : o0) * C.det_stride + C.det_off;
        ol[h] = C.det_label[in ? o : o0];
#pragma unroll
        for (int k = 0; k < 3; ++k) q[h][k] = dw[k];
      }
      ASTAMPW(8);
#ifdef SLIDE_STAMPS
      if (q[0][0] + q[1][0] == 1.2345e300) nh = 0;      // (forces the loads to complete before the stamp)
#endif
      ASTAMPW(9);
      double b2[HM];                        // squared distance of the lane's best (none yet: < 0)
      int bests[HM];
      unsigned long long bkey[HM];          // order of the candidates: the select's key (gate) or the index itself
#pragma unroll
      for (int h = 0; h < HM; ++h) { b2[h] = -1.0; bests[h] = INT_MAX; bkey[h] = ~0ull; }
      // the exact rule on one candidate: a later candidate replaces the lane's best iff its d is strictly smaller — decided on the
      // squared distances when they differ by more than 2^-48 relative (the correctly rounded roots are then distinct), by the two
      // roots themselves inside that band (equal roots: the order).  The f64 sqrt stays out of the common path.
      auto exact = [&](int h, int s, unsigned long long key, double m0, double m1, double m2) {
        const double dx = q[h][0] - m0, dy = q[h][1] - m1, dz = q[h][2] - m2;
        const double d2 = dx * dx + dy * dy + dz * dz;
        bool take = b2[h] < 0.0 || d2 < b2[h] * BAND;
        if (!take && d2 * BAND <= b2[h]) {
          const double rd = sqrt(d2), rb = sqrt(b2[h]);
          take = rd < rb || (rd == rb && key < bkey[h]);
        }
        if (take) { b2[h] = d2; bests[h] = s; bkey[h] = key; }
      };
      if (staged) {
        constexpr float U = 0x1p-24f;
        const float INF = __uint_as_float(0x7f800000u);
        float qf[HM][3], mf[HM], T2[HM];
#pragma unroll
        for (int h = 0; h < HM; ++h) {
#pragma unroll
          for (int k = 0; k < 3; ++k) qf[h][k] = (float)(q[h][k] - C.qpos[k]);
          mf[h] = INF;
        }
        auto screen = [&](int h, int lab, float x, float y, float z) -> float {
          const float dx = qf[h][0] - x, dy = qf[h][1] - y, dz = qf[h][2] - z;
          const float d2 = dx * dx + dy * dy + dz * dz;
          return (C.label_gate == 1 && lab != ol[h]) ? INF : d2;
        };
        for (int s0 = lane; s0 < Ksub; s0 += 256) {
          float x[4], y[4], z[4];
          int lab[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int s = s0 + 64 * u;
            const int sc = s < Ksub ? s : s0;
            x[u] = fx[sc]; y[u] = fy[sc]; z[u] = fz[sc]; lab[u] = flab[sc];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (s0 + 64 * u >= Ksub) continue;
#pragma unroll
            for (int h = 0; h < HM; ++h) mf[h] = fminf(mf[h], screen(h, lab[u], x[u], y[u], z[u]));
          }
        }
        const float cmax = __uint_as_float(s_cmax);
#pragma unroll
        for (int h = 0; h < HM; ++h) {
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) mf[h] = fminf(mf[h], __shfl_xor(mf[h], m));
          const float qm = fmaxf(fabsf(qf[h][0]), fmaxf(fabsf(qf[h][1]), fabsf(qf[h][2])));
          const float E = 4.0f * U * (qm + cmax);
          const float T = (sqrtf(mf[h]) + 2.0f * E) * (1.0f + 8.0f * U);
          T2[h] = mf[h] < INF ? T * T * (1.0f + 4.0f * U) : -1.0f;      // (no candidate of that label: nothing passes)
          if (h >= nh) T2[h] = -1.0f;
        }
        for (int s0 = lane; s0 < Ksub; s0 += 256) {
          float x[4], y[4], z[4];
          int lab[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int s = s0 + 64 * u;
            const int sc = s < Ksub ? s : s0;
            x[u] = fx[sc]; y[u] = fy[sc]; z[u] = fz[sc]; lab[u] = flab[sc];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int s = s0 + 64 * u;
            if (s >= Ksub) continue;
            bool pass[HM], any = false;
#pragma unroll
            for (int h = 0; h < HM; ++h) { pass[h] = screen(h, lab[u], x[u], y[u], z[u]) <= T2[h]; any = any || pass[h]; }
            if (any) {
              const unsigned long long key = sel[s];
              const double* mm = C.model + 3 * (size_t)(key & 0xffffffffull);
              const double m0 = mm[0], m1 = mm[1], m2 = mm[2];
#pragma unroll
              for (int h = 0; h < HM; ++h)
                if (pass[h]) exact(h, s, key, m0, m1, m2);
            }
          }
        }
      } else {
        // the submap as it lies in global memory (the pure matchers: no gate, the map itself in the caller's order); four candidates'
        // words are loaded before the first is looked at
        for (int s0 = lane; s0 < Ksub; s0 += 256) {
          unsigned long long key[4];
          int ml[4];
          double m0[4], m1[4], m2[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int s = s0 + 64 * u;
            const int sc = s < Ksub ? s : s0;
            key[u] = C.gate ? sel[sc] : (unsigned long long)sc;
            const int mi = C.gate ? (int)(key[u] & 0xffffffffull) : sc;
            const double* mm = C.model + 3 * (size_t)mi;
            m0[u] = mm[0]; m1[u] = mm[1]; m2[u] = mm[2];
            ml[u] = C.label[mi];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int s = s0 + 64 * u;
            if (s >= Ksub) continue;
#pragma unroll
            for (int h = 0; h < HM; ++h) {
              if (h >= nh) continue;
              if (C.label_gate == 1 && ml[u] != ol[h]) continue;
              exact(h, s, key[u], m0[u], m1[u], m2[u]);
            }
          }
        }
      }
      ASTAMPW(10);
#pragma unroll
      for (int h = 0; h < HM; ++h) {
        if (h >= nh) continue;
        double best = C.best_init;
        const double d = b2[h] >= 0.0 ? sqrt(b2[h]) : C.best_init;
        if (d < C.best_init) best = d; else { bests[h] = INT_MAX; bkey[h] = ~0ull; }      // "if (d < bestDist)" against the initial bestDist
        if (h == nh - 1) ASTAMPW(11);
        reduce_write(o0 + h * nwave, best, bests[h], bkey[h]);
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
__global__ __launch_bounds__(512, 4) void k_assoc_sweep_512(const float* __restrict__ cx, const float* __restrict__ cy, const float* __restrict__ cz,
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

// ---- round 5: the sweep with the distance words in REGISTERS -----------------------------------------------------------------------
// k_assoc_sweep_512 keeps a frame's n distance words in LDS (40 KB for the 10 k-landmark map) next to 16 KB of select tables, 8 KB of
// keys and — during the matching — the survivors' double-precision models: 66 KB, two workgroups per CU, and every phase of a frame
// ends at a workgroup barrier (a frame alone on a CU takes 18.3 us, two side by side 21.5 us each: the phases are latency, not issue).
// Here a thread keeps ITS keys (map index tid + j * 512, j < NK) in NK registers from the distance scan to the placement of the selected
// keys — the select's passes never revisit anybody else's — and the exact rule of the matching reads the one or two models it needs
// from memory instead of LDS copies of all K: 38 KB of LDS per workgroup, THREE workgroups per CU.  Same select (one histogram over bins
// linear in the squared distance, ranks inside the K-th key's bin, digit passes when that bin overflows), same label groups, same
// screening bound, same exact rule: the matches are the same ids (tests/test_gpu_kernels.py sweep tests run through this kernel).
// For: label-gated boxes / points, no submap list asked for, n <= NK * 512, K <= 1024, at most 64 detections per frame.
// Wave-wide reductions and scans on DPP row operations: __shfl_xor / __shfl_up go through ds_bpermute, whose six lane-address registers per
// shuffle pattern the compiler keeps for the whole kernel — a dozen of the eighty vector registers of k_assoc_sweep_r.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {      // result uniform (scalar)
  v = max(v, (unsigned)dpp_i32<0xB1>((int)v));      // quad_perm [1,0,3,2]
  v = max(v, (unsigned)dpp_i32<0x4E>((int)v));      // quad_perm [2,3,0,1]
  v = max(v, (unsigned)dpp_i32<0x141>((int)v));     // row_half_mirror
  v = max(v, (unsigned)dpp_i32<0x140>((int)v));     // row_mirror: every lane of a row of sixteen holds the row's maximum
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16),
                 c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}
__device__ __forceinline__ float wave_min_f32(float v) {            // result uniform
  v = fminf(v, __int_as_float(dpp_i32<0xB1>(__float_as_int(v))));
  v = fminf(v, __int_as_float(dpp_i32<0x4E>(__float_as_int(v))));
  v = fminf(v, __int_as_float(dpp_i32<0x141>(__float_as_int(v))));
  v = fminf(v, __int_as_float(dpp_i32<0x140>(__float_as_int(v))));
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)),
              c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fminf(fminf(a, b), fminf(c, d));
}
__device__ __forceinline__ float wave_max_f32(float v) {            // v >= 0: the bit patterns order like the values
  return __uint_as_float(wave_max_u32(__float_as_uint(v)));
}
__device__ __forceinline__ int wave_incl_scan_i32(int v, int lane) {
  v += dpp_i32<0x111>(v);      // row_shr:1 (lanes without a source inside their row of sixteen add 0)
  v += dpp_i32<0x112>(v);
  v += dpp_i32<0x114>(v);
  v += dpp_i32<0x118>(v);
  const int s0 = __builtin_amdgcn_readlane(v, 15), s1 = __builtin_amdgcn_readlane(v, 31), s2 = __builtin_amdgcn_readlane(v, 47);
  return v + (lane >= 16 ? s0 : 0) + (lane >= 32 ? s1 : 0) + (lane >= 48 ? s2 : 0);
}

template <int NK, int NTHR, int NREG>
__device__ __forceinline__ void sweep_core_r(const AssocCore& C) {
  __shared__ unsigned long long sbuf64[1536];      // 12 KB: bin counts (1024 x 4 B) + candidate keys (1024 x 8 B); the fallback's per-wave digit histograms (8 x 256 x 4 B) + their sums
  __shared__ int s_dl[ASSOC_GMAX], s_dgrp[ASSOC_GMAX], s_glab[ASSOC_GMAX], s_gcnt[ASSOC_GMAX], s_gcur[ASSOC_GMAX], s_goff[ASSOC_GMAX], s_ng;
  __shared__ unsigned long long s_prefix;
  __shared__ int s_krem, s_stop, s_cnt, s_ccnt, s_bin, s_before;
  __shared__ unsigned s_wred[16];
  __shared__ unsigned s_cmax;
  unsigned* lhist = reinterpret_cast<unsigned*>(sbuf64);
  unsigned long long* lcand = sbuf64 + ASSOC_BINS / 2;
  unsigned* whist = reinterpret_cast<unsigned*>(sbuf64);
  unsigned* hist = whist + 2048;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int nthr = NTHR, nw = NTHR >> 6;
  const int n = C.n, Ksub = C.K < n ? C.K : n;
  unsigned long long* sel = assoc_lds;
  float* fx = reinterpret_cast<float*>(assoc_lds + C.Kp);
  float* fy = fx + Ksub;
  float* fz = fx + 2 * (size_t)Ksub;
  int* sidx = reinterpret_cast<int*>(fx + 3 * (size_t)Ksub);
  ASTAMP(0);
  if (tid < ASSOC_GMAX) { s_dl[tid] = tid < C.n_det ? C.det_label[tid] : 0; s_gcnt[tid] = 0; s_gcur[tid] = 0; }
  for (int b = tid; b < ASSOC_BINS; b += nthr) lhist[b] = 0u;
  if (tid == 0) { s_prefix = 0ull; s_krem = Ksub; s_stop = 0; s_cnt = 0; s_ccnt = 0; s_bin = -1; s_before = 0; s_cmax = 0u; }
  // (uniform values the vector ALU produced go back to scalar registers: the kernel has 80 vector registers and needs every one)
  auto uni = [](float v) -> float { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
  const float qx = uni((float)C.qpos[0]), qy = uni((float)C.qpos[1]), qz = uni((float)C.qpos[2]);
  // ---- the own keys' distance words: LD loads in flight per stream, straight-line (indices clamped, not branched around) ----
  // Round j of a thread's keys is map index tid + j * nthr.  Which rounds exist is decided by ONE lane predicate (tid < rem, the partial
  // round) and scalar compares — twenty per-round index registers and twenty compare masks, kept alive from here to the placement, were
  // what made the first version of this kernel spill.
  // rounds NREG .. NK - 1 of the words live in LDS (in the region the staging of the survivors takes over after the placement): with all
  // twenty in registers five dwords per lane spilled
  unsigned r[NREG];
  unsigned* rl = reinterpret_cast<unsigned*>(fx);      // [(j - NREG) * nthr + tid]
  unsigned bmax = 0u;
  constexpr int LD = 2;
  const int nlast = n - 1;
  const int jfull = n / nthr, rem = n - jfull * nthr;
  const bool in_rem = tid < rem;
  auto valid = [&](int j) -> bool { return j < jfull || (j == jfull && in_rem); };
  auto fresh_tid = [&]() -> int { int t = tid; asm volatile("" : "+v"(t)); return t; };      // (a new value: indices derived from it are not kept across phases)
  const int t1 = fresh_tid();
#pragma unroll
  for (int j0 = 0; j0 < NK; j0 += LD) {
    float x[LD], y[LD], z[LD];
#pragma unroll
    for (int u = 0; u < LD; ++u) {
      if (j0 + u < NK) {
        const int i = t1 + (j0 + u) * nthr;
        const int ic = i < nlast ? i : nlast;
        x[u] = C.cx[ic]; y[u] = C.cy[ic]; z[u] = C.cz[ic];
      }
    }
#pragma unroll
    for (int u = 0; u < LD; ++u) {
      if (j0 + u < NK) {
        const float dx = x[u] - qx, dy = y[u] - qy, dz = z[u] - qz;
        float d = dx * dx;
        d += dy * dy;
        d += dz * dz;
        const unsigned b = __float_as_uint(d);       // (the arithmetic of dist_bits)
        if (valid(j0 + u)) bmax = b > bmax ? b : bmax;
        // (consumed here: without the pin the scheduler issues all NK x 3 loads first and their sixty results spill)
        if (j0 + u < NREG) {
          r[j0 + u] = b;
          asm volatile("" : "+v"(r[j0 + u]) : : "memory");
        } else {
          rl[(j0 + u - NREG) * nthr + tid] = b;
          asm volatile("" : : : "memory");
        }
      }
    }
  }
  bmax = wave_max_u32(bmax);
  if (lane == 0) s_wred[wave] = bmax;
  __syncthreads();
  ASTAMP(1);
  // (the overflow path below recomputes its keys from the cloud: the register copies are not kept alive for it)
  auto key_at_g = [&](int i) -> unsigned long long { return ((unsigned long long)dist_bits(C, i, qx, qy, qz) << 32) | (unsigned)i; };
  auto rw = [&](int j) -> unsigned { return j < NREG ? r[j < NREG ? j : 0] : rl[(j - NREG) * nthr + tid]; };      // (j: compile-time after unrolling; own words only: no barrier needed)
  int t3 = 0;      // (a fresh copy of the thread index, taken where the keys are written)
  // a key = (distance word << 32) | map index, written as its two halves (one ds_write2_b32): formed as 64-bit values the twenty keys
  // want twenty aligned register pairs
  auto put_key = [&](unsigned long long* dst, int j) {
    unsigned* p2 = reinterpret_cast<unsigned*>(dst);
    p2[0] = (unsigned)(t3 + j * nthr);
    p2[1] = rw(j);
  };
  float scale = 0.0f;
  auto bin_of = [&](unsigned bits) -> int {
    const int b = (int)(__uint_as_float(bits) * scale);
    return b < ASSOC_BINS - 1 ? b : ASSOC_BINS - 1;
  };
  int fbin = -1, shift = 64;
  bool ranked = false;
  if (n > 0 && Ksub < n) {
    for (int w = 0; w < nw; ++w) bmax = s_wred[w] > bmax ? s_wred[w] : bmax;
    const float rmax = __uint_as_float(bmax);
    scale = rmax > 0.0f ? ((float)ASSOC_BINS - 0.5f) / rmax : 0.0f;
    if (!(scale == scale) || scale > 3.0e38f) scale = 0.0f;
    scale = uni(scale);
#pragma unroll
    for (int j = 0; j < NK; ++j)
    {
      if (j < NREG) asm volatile("" : "+v"(r[j < NREG ? j : 0]));      // (one bin address at a time: twenty computed ahead of their atomics are twenty more registers)
      if (valid(j)) atomicAdd(&lhist[bin_of(rw(j))], 1u);
    }
    __syncthreads();
    ASTAMP(12);
    {
      const int per = (ASSOC_BINS + nthr - 1) / nthr;
      unsigned own = 0;
      for (int k = 0; k < per; ++k) { const int bb = tid * per + k; if (bb < ASSOC_BINS) own += lhist[bb]; }
      const unsigned inc = (unsigned)wave_incl_scan_i32((int)own, lane);
      if (lane == 63) s_wred[wave] = inc;
      __syncthreads();
      unsigned base = 0;
      for (int w = 0; w < wave; ++w) base += s_wred[w];
      const unsigned before_t = base + inc - own;
      if (before_t < (unsigned)Ksub && (unsigned)Ksub <= before_t + own) {      // exactly one thread: the counts sum to n >= K
        unsigned before = before_t;
        int bb = tid * per;
        while (before + lhist[bb] < (unsigned)Ksub) { before += lhist[bb]; ++bb; }
        s_bin = bb;
        s_before = (int)before;
        s_krem = Ksub - (int)before;
      }
      __syncthreads();
    }
    ASTAMP(13);
    fbin = s_bin;
    const int before = s_before, krem = s_krem;
    const int ncand = (int)lhist[fbin];
    if (ncand <= ASSOC_CAND_CAP) {
      unsigned tm = 0u, cm = 0u;
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        if (j < NREG) asm volatile("" : "+v"(r[j < NREG ? j : 0]));
        if (valid(j)) {
          const int b = bin_of(rw(j));
          tm |= (b < fbin ? 1u : 0u) << j;
          cm |= (b == fbin ? 1u : 0u) << j;
        }
      }
      t3 = fresh_tid();
      const int ct = __popc(tm), cc = __popc(cm);
      const int st = wave_incl_scan_i32(ct, lane), sc = wave_incl_scan_i32(cc, lane);
      int bt = 0, bc = 0;
      if (lane == 63) {
        if (st > 0) bt = atomicAdd(&s_cnt, st);
        if (sc > 0) bc = atomicAdd(&s_ccnt, sc);
      }
      bt = __builtin_amdgcn_readlane(bt, 63) + st - ct;
      bc = __builtin_amdgcn_readlane(bc, 63) + sc - cc;
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        if ((tm >> j) & 1u) put_key(&sel[bt++], j);
        if ((cm >> j) & 1u) put_key(&lcand[bc++], j);
      }
      __syncthreads();
      ASTAMP(14);
      for (int jj = tid; jj < ncand; jj += nthr) {
        const unsigned long long kj = lcand[jj];
        int rank = 0;
        for (int i = 0; i < ncand; ++i) rank += lcand[i] < kj ? 1 : 0;
        if (rank < krem) sel[before + rank] = kj;
      }
      ranked = true;
    } else {
      // the K-th key's bin overflows the candidate list: most-significant-digit passes over the keys of that bin (as knn_select)
      __syncthreads();
      for (int byte = 7; byte >= 0; --byte) {
        shift = 8 * byte;
        for (int b = tid; b < 256 * nw; b += nthr) whist[b] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        unsigned* mine = whist + 256 * wave;
        for (int i = tid; i < n; i += nthr) {
          const unsigned long long key = key_at_g(i);
          if (bin_of((unsigned)(key >> 32)) != fbin) continue;
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
          const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
          const unsigned own = h0 + h1 + h2 + h3;
          unsigned inc = own;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = (unsigned)__shfl_up((int)inc, off);
            if (lane >= off) inc += o;
          }
          const unsigned kr = (unsigned)s_krem;
          const unsigned long long reach = __ballot(inc >= kr);
          const int first = __ffsll((long long)reach) - 1;
          if (lane == first) {
            unsigned bef = inc - own;
            unsigned d = 4 * lane, cnt = h0;
            if (bef + h0 < kr) { bef += h0; d += 1; cnt = h1;
              if (bef + h1 < kr) { bef += h1; d += 1; cnt = h2;
                if (bef + h2 < kr) { bef += h2; d += 1; cnt = h3; } } }
            s_prefix = prefix | ((unsigned long long)d << shift);
            s_krem = (int)(kr - bef);
            s_stop = (bef + cnt == kr) ? 1 : 0;
          }
        }
        __syncthreads();
        if (s_stop) break;
      }
    }
  }
  ASTAMP(2);
  if (fbin < 0) {
    t3 = fresh_tid();
#pragma unroll
    for (int j = 0; j < NK; ++j)
      if (valid(j)) put_key(&sel[t3 + j * nthr], j);
  } else if (!ranked) {
    const unsigned long long lim = shift < 64 ? (s_prefix >> shift) : 0ull;
    auto taken = [&](unsigned long long key) -> bool {
      const int b = bin_of((unsigned)(key >> 32));
      if (b != fbin) return b < fbin;
      return (key >> shift) <= lim;
    };
    const int n_up = (n + nthr - 1) / nthr * nthr;
    int mine = 0;
    for (int i = tid; i < n_up; i += nthr) mine += __popcll(__ballot(i < n && taken(key_at_g(i < n ? i : 0))));      // (wave-uniform)
    int base = 0;
    if (lane == 0 && mine > 0) base = atomicAdd(&s_cnt, mine);
    base = __shfl(base, 0);
    for (int i = tid; i < n_up; i += nthr) {
      const unsigned long long key = key_at_g(i < n ? i : 0);
      const bool take = i < n && taken(key);
      const unsigned long long m = __ballot(take);
      if (take) sel[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
      base += __popcll(m);
    }
  }
  __syncthreads();
  ASTAMP(3);
  ASTAMP(4);
  // (the thread index as a NEW value: what the phases below derive from it — addresses, lane masks, predicates — is otherwise computed at
  // the kernel's entry and kept in registers across the select, whose twenty key registers then spill)
  int tidb = tid;
  asm volatile("" : "+v"(tidb));
  const int laneb = tidb & 63;
  // ---- the survivors grouped by label (assoc_core's staging; the double-precision models stay in memory) ----
  float cm = 0.0f;
  {
    double mx[4], my[4], mz[4];
    int lab[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int s = tidb + it * nthr;
      lab[it] = 0;
      mx[it] = my[it] = mz[it] = 0.0;
      if (s < Ksub) {
        const int mi = (int)(sel[s] & 0xffffffffull);
        const double* mm = C.model + 3 * (size_t)mi;
        mx[it] = mm[0]; my[it] = mm[1]; mz[it] = mm[2];
        lab[it] = C.label[mi];
      }
    }
    if (tidb < 64) {
      const int mylab = s_dl[laneb];
      const bool in = laneb < C.n_det;
      bool first = in;
      for (int p = 0; p < C.n_det; ++p) {
        const int lp = __builtin_amdgcn_readlane(mylab, p);
        if (p < laneb && lp == mylab) first = false;
      }
      const unsigned long long fm = __ballot(first);
      int jj = __popcll(fm & ((1ull << laneb) - 1ull));
      if (first) s_glab[jj] = mylab;
      const int jf = jj;
      for (int p = 0; p < C.n_det; ++p) {
        const int lp = __builtin_amdgcn_readlane(mylab, p), jp = __builtin_amdgcn_readlane(jf, p);
        if (((fm >> p) & 1ull) && lp == mylab) jj = jp;
      }
      if (in) s_dgrp[laneb] = jj;
      if (laneb == 0) s_ng = __popcll(fm);
    }
    __syncthreads();
    ASTAMP(7);
    const int ng = s_ng;
    const int glab_l = laneb < ng ? s_glab[laneb] : 0;
    float x[4], y[4], z[4];
    int g[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      g[it] = -1;
      x[it] = (float)(mx[it] - C.qpos[0]); y[it] = (float)(my[it] - C.qpos[1]); z[it] = (float)(mz[it] - C.qpos[2]);
    }
    for (int j = 0; j < ng; ++j) {
      const int gl = __builtin_amdgcn_readlane(glab_l, j);
#pragma unroll
      for (int it = 0; it < 4; ++it)
        if (tidb + it * nthr < Ksub && lab[it] == gl) g[it] = j;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it)
      if (g[it] >= 0) cm = fmaxf(cm, fmaxf(fabsf(x[it]), fmaxf(fabsf(y[it]), fabsf(z[it]))));
    int wc[4] = {0, 0, 0, 0};
    for (int j = 0; j < ng; ++j) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        if (it * nthr >= Ksub) continue;               // (uniform)
        const int c = __popcll(__ballot(g[it] == j));
        if (laneb == j) wc[it] = c;
      }
    }
    const int wtot = wc[0] + wc[1] + wc[2] + wc[3];
    if (laneb < ng && wtot > 0) atomicAdd(&s_gcnt[laneb], wtot);
    __syncthreads();
    ASTAMP(15);
    {
      const int v = laneb < ng ? s_gcnt[laneb] : 0;
      const int inc = wave_incl_scan_i32(v, laneb);
      if (tidb < ng) s_goff[tidb] = inc - v;
      int base = inc - v;
      if (laneb < ng && wtot > 0) base += atomicAdd(&s_gcur[laneb], wtot);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        if (it * nthr >= Ksub) continue;
        int rank = 0;
        for (int j = 0; j < ng; ++j) {
          const unsigned long long m = __ballot(g[it] == j);
          if (g[it] == j) rank = __popcll(m & ((1ull << laneb) - 1ull));
        }
        const int gb = __shfl(base, g[it] >= 0 ? g[it] : 0);
        if (g[it] >= 0) {
          const int pos = gb + rank;
          fx[pos] = x[it]; fy[pos] = y[it]; fz[pos] = z[it];
          sidx[pos] = tidb + it * nthr;
        }
        base += wc[it];
      }
    }
    cm = wave_max_f32(cm);
    if (laneb == 0) atomicMax(&s_cmax, __float_as_uint(cm));
  }
  __syncthreads();
  ASTAMP(5);
  // ---- matching: a wavefront per detection over the survivors of its label (assoc_core's grouped branch) ----
  constexpr double BAND = 1.0 - 0x1p-48;
  constexpr float U = 0x1p-24f;
  const float INF = __uint_as_float(0x7f800000u);
  const float cmax = __uint_as_float(s_cmax);
  for (int o = wave; o < C.n_det; o += nw) {
    const double* dw = C.det_world + (size_t)o * C.det_stride + C.det_off;
    double q[3];
    float qf[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { q[k] = dw[k]; qf[k] = (float)(q[k] - C.qpos[k]); }
    const int gg = __builtin_amdgcn_readfirstlane(s_dgrp[o]);
    const int lo = __builtin_amdgcn_readfirstlane(s_goff[gg]), cnt = __builtin_amdgcn_readfirstlane(s_gcnt[gg]);
    ASTAMPW(8);
    ASTAMPW(9);
    auto fd2 = [&](int t) -> float {
      const float dx = qf[0] - fx[lo + t], dy = qf[1] - fy[lo + t], dz = qf[2] - fz[lo + t];
      return dx * dx + dy * dy + dz * dz;
    };
    // pass 1: the float distances' minimum over the group; pass 2 recomputes them (three LDS words and eight flops per survivor, ~3 per
    // laneb) instead of carrying them in registers: the kernel lives in 80 registers
    float mf = INF;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (int t = laneb; t < cnt; t += 64) mf = fminf(mf, fd2(t));
    mf = wave_min_f32(mf);
    const float qm = fmaxf(fabsf(qf[0]), fmaxf(fabsf(qf[1]), fabsf(qf[2])));
    const float E = 4.0f * U * (qm + cmax);
    const float T = (sqrtf(mf) + 2.0f * E) * (1.0f + 8.0f * U);
    const float T2 = mf < INF ? T * T * (1.0f + 4.0f * U) : -1.0f;      // (no survivor of that label: nothing passes)
    double b2 = -1.0;
    int bests = INT_MAX;
    unsigned long long bkey = ~0ull;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (int t = laneb; t < cnt; t += 64) {
      if (fd2(t) <= T2) {                   // (one or two lanes of the wave, once: their models come from memory)
        const int s = sidx[lo + t];
        const unsigned long long key = sel[s];
        const double* mm = C.model + 3 * (size_t)(key & 0xffffffffull);
        const double dx = q[0] - mm[0], dy = q[1] - mm[1], dz = q[2] - mm[2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        // the exact rule (assoc_core): a later candidate replaces the laneb's best iff its d is strictly smaller — decided on the squared
        // distances when they differ by more than 2^-48 relative, by the two correctly rounded roots inside that band (equal: the order)
        bool take = b2 < 0.0 || d2 < b2 * BAND;
        if (!take && d2 * BAND <= b2) {
          const double rd = sqrt(d2), rb = sqrt(b2);
          take = rd < rb || (rd == rb && key < bkey);
        }
        if (take) { b2 = d2; bests = s; bkey = key; }
      }
    }
    ASTAMPW(10);
    double best = C.best_init;
    const double d = b2 >= 0.0 ? sqrt(b2) : C.best_init;
    if (d < C.best_init) best = d; else { bests = INT_MAX; bkey = ~0ull; }      // "if (d < bestDist)" against the initial bestDist
    ASTAMPW(11);
    // lexicographic (distance, key) minimum over the wave = the reference's strict-'<' first-wins rule, then the threshold test
    const unsigned long long have = __ballot(bests != INT_MAX);
    if (__popcll(have) <= 1) {
      const int w = have ? __ffsll((long long)have) - 1 : 0;
      if (laneb == w) {
        const bool ok = (bests != INT_MAX) && (best < C.thresh);
        C.match_map[o] = ok ? (int32_t)(bkey & 0xffffffffull) : -1;
      }
    } else {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const int os = __shfl_xor(bests, off);
        const unsigned klo = (unsigned)__shfl_xor((int)(unsigned)bkey, off), khi = (unsigned)__shfl_xor((int)(unsigned)(bkey >> 32), off);
        const unsigned long long ok_ = ((unsigned long long)khi << 32) | klo;
        if (ob < best || (ob == best && ok_ < bkey)) { best = ob; bests = os; bkey = ok_; }
      }
      if (laneb == 0) {
        const bool ok = (bests != INT_MAX) && (best < C.thresh);
        C.match_map[o] = ok ? (int32_t)(bkey & 0xffffffffull) : -1;
      }
    }
  }
  ASTAMP(6);
}
constexpr int SWEEP_R_NK = 20, SWEEP_R_NREG = 16;
__global__ __launch_bounds__(512, 6) void k_assoc_sweep_r(const float* __restrict__ cx, const float* __restrict__ cy, const float* __restrict__ cz,
                                                          const double* __restrict__ model_xyz, const int32_t* __restrict__ label, int n_map,
                                                          const double* __restrict__ query_pos, const double* __restrict__ obs_xyz,
                                                          const int32_t* __restrict__ obs_label, int n_obs, int K, int Kp, double thresh,
                                                          int32_t* __restrict__ out_map_idx) {
  const int q = blockIdx.x;
  AssocCore C;
  C.cx = cx; C.cy = cy; C.cz = cz; C.model = model_xyz; C.label = label; C.n = n_map; C.K = K;
  C.gate = 1; C.Kp = Kp; C.cached = 0; C.staged = 1;
  C.thresh = thresh; C.best_init = 1000.0; C.label_gate = 1; C.is_cyl = 0;
  C.qpos = query_pos + 3 * (size_t)q;
  C.det_world = obs_xyz + 3 * (size_t)q * n_obs;
  C.det_stride = 3; C.det_off = 0;
  C.det_label = obs_label + (size_t)q * n_obs; C.n_det = n_obs;
  C.match_sub = nullptr; C.submap = nullptr; C.n_sub = nullptr;
  C.match_map = out_map_idx + (size_t)q * n_obs;
  sweep_core_r<SWEEP_R_NK, 512, SWEEP_R_NREG>(C);
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
  // staged survivors: cylinders 7 doubles + label; boxes / points 3 floats + an int (label, or position in sel) + the 3 doubles of the model
  const size_t dc = (size_t)n * 4, st = ((size_t)Ksub * (model_stride == 3 ? 40 : model_stride * 8 + 4) + 7) / 8 * 8;
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
  // round 5: the register-resident sweep, three workgroups per CU (SLIDE_ASSOC_REG=0: the kernels below)
  static const int env_reg = getenv("SLIDE_ASSOC_REG") ? atoi(getenv("SLIDE_ASSOC_REG")) : 1;
  if (env_reg && env_thr == 512 && n_map > 0 && n_map <= SWEEP_R_NK * 512 && Kp <= 1024 && n_obs <= ASSOC_GMAX) {
    const int Ksub = K < n_map ? K : n_map;
    // keys + the staged survivors (16 B each); the same region first holds the last four rounds of distance words (4 x 512 x 4 B)
    const size_t lds = (size_t)Kp * 8 + std::max<size_t>((((size_t)Ksub * 16 + 7) / 8) * 8, (size_t)(SWEEP_R_NK - SWEEP_R_NREG) * 512 * 4);
    hipLaunchKernelGGL(k_assoc_sweep_r, dim3(n_query), dim3(512), lds, s, cx, cy, cz, model_xyz, label, n_map, query_pos, obs_xyz, obs_label,
                       n_obs, K, Kp, thresh, out_map_idx);
    return 0;
  }
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
