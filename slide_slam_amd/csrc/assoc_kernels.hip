// Per-frame semantic data association on gfx950 — the "association sweep" of BASELINE.json.
// Reference: *MapManager::getSubmap (backend/sloam/src/core/cubeMapManager.cpp:36-75,
// cylinderMapManager.cpp:213-243, ellipsoidMapManager.cpp:40-80), sloam::projectModels + match*Models
// (src/core/sloam.cpp:73-217), object distances (src/objects/cube.cpp:22-24, ellipsoid.cpp:24-26,
// cylinder.cpp:187-224).
//
// One workgroup per (frame, class).  The K-NN gate is an exact float32 brute-force scan of the
// first-seen cloud (coalesced SoA stream from HBM/L2) followed by an in-LDS bitonic sort of
// (distance bits, map index) keys, which reproduces the reference's nearest-first submap order; the
// nearest-neighbour match is one wavefront per detection with a lexicographic (distance, submap index)
// shuffle reduction, which reproduces the reference's strict-'<' first-index-wins rule.
// This file is compiled with -ffp-contract=off: distances are compared against thresholds, so the
// arithmetic must round exactly like the reference's un-fused x86-64 build.
#include <hip/hip_runtime.h>
#include <limits.h>

#include "kernels.hpp"
#include "sl_math.hpp"

namespace sl {

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
  const float* cloud; const double* model; const int32_t* label; int n; int K;
  double thresh, best_init; int label_gate, is_cyl;
  const double* qpos;        // 3: robot position (double; narrowed to float like PointT)
  const double* det_world;   // cyl: 7 per ; box: xyz taken at stride `det_stride` offset `det_off`
  int det_stride, det_off;
  const int32_t* det_label; int n_det;
  int32_t* match_sub; int32_t* match_map; int32_t* submap; int32_t* n_sub;
};

// keys: dynamic LDS, capacity Np (power of two >= n)
__device__ inline void assoc_core(const AssocCore& C, unsigned long long* keys, int Np) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int Ksub = C.K < C.n ? C.K : C.n;
  if (C.n > 0) {
    const float qx = (float)C.qpos[0], qy = (float)C.qpos[1], qz = (float)C.qpos[2];
    for (int i = tid; i < Np; i += nthr) {
      unsigned long long key = ~0ull;
      if (i < C.n) {
        const float dx = C.cloud[3 * i] - qx, dy = C.cloud[3 * i + 1] - qy, dz = C.cloud[3 * i + 2] - qz;
        float r = dx * dx;
        r += dy * dy;
        r += dz * dz;
        key = ((unsigned long long)__float_as_uint(r) << 32) | (unsigned)i;
      }
      keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= Np; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < Np; i += nthr) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long a = keys[i], b = keys[ixj];
            const bool asc = (i & k) == 0;
            if ((a > b) == asc) { keys[i] = b; keys[ixj] = a; }
          }
        }
        __syncthreads();
      }
    }
    for (int s = tid; s < Ksub; s += nthr) C.submap[s] = (int32_t)(keys[s] & 0xffffffffull);
  }
  if (tid == 0) *C.n_sub = Ksub;
  __syncthreads();
  // one wavefront per detection
  const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  for (int o = wave; o < C.n_det; o += nwave) {
    double best = C.best_init;
    int bests = INT_MAX;
    const int ol = C.det_label[o];
    const double* dw = C.det_world + (size_t)o * C.det_stride + C.det_off;
    for (int s = lane; s < Ksub; s += 64) {
      const int mi = (int)(keys[s] & 0xffffffffull);
      double d;
      bool consider = true;
      if (C.is_cyl) {
        d = cyl_distance(C.model + 7 * (size_t)mi, C.label[mi], dw - C.det_off, ol);
      } else {
        if (C.label_gate == 1 && C.label[mi] != ol) consider = false;
        const double dx = dw[0] - C.model[3 * (size_t)mi], dy = dw[1] - C.model[3 * (size_t)mi + 1],
                     dz = dw[2] - C.model[3 * (size_t)mi + 2];
        d = sqrt(dx * dx + dy * dy + dz * dz);
      }
      // sequential rule: "if (d < bestDist)" scanning s upward -> lexicographic min over (d, s) of the
      // candidates that beat the initial bestDist
      if (consider && d < C.best_init && (d < best || (d == best && s < bests))) { best = d; bests = s; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off);
      const int os = __shfl_xor(bests, off);
      if (ob < best || (ob == best && os < bests)) { best = ob; bests = os; }
    }
    if (lane == 0) {
      const bool ok = (bests != INT_MAX) && (best < C.thresh);
      C.match_sub[o] = ok ? bests : -1;
      C.match_map[o] = ok ? (int32_t)(keys[bests] & 0xffffffffull) : -1;
    }
  }
}

extern __shared__ unsigned long long dyn_keys[];

__device__ inline int pow2_at_least(int n) {
  int p = 64;
  while (p < n) p <<= 1;
  return p;
}

// grid = 3 (cylinders, cubes, ellipsoids of ONE key frame)
__global__ __launch_bounds__(1024) void k_assoc_frame(const AssocFrameDev* __restrict__ cls3, const double* __restrict__ pose12,
                                                      int* status) {
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
  if (F.n > ASSOC_MAX_N) {
    if (tid == 0) atomicOr(status, 1);
    return;
  }
  AssocCore C;
  C.cloud = F.cloud; C.model = F.model; C.label = F.label; C.n = F.n; C.K = F.K;
  C.thresh = F.thresh; C.best_init = F.best_init; C.label_gate = F.label_gate; C.is_cyl = F.is_cyl;
  C.qpos = pose12 + 9;
  C.det_world = F.det_world;
  C.det_stride = F.is_cyl ? 7 : 12;
  C.det_off = F.is_cyl ? 0 : 9;
  C.det_label = F.det_label; C.n_det = F.n_det;
  C.match_sub = F.match_sub; C.match_map = F.match_map; C.submap = F.submap; C.n_sub = F.n_sub;
  assoc_core(C, dyn_keys, pow2_at_least(F.n));
}

// Batched sweep: one workgroup per independent query frame against one resident map
// (label-gated points, i.e. the ellipsoid / point-landmark class of the headline graph).
__global__ __launch_bounds__(1024) void k_assoc_sweep(const float* __restrict__ cloud, const double* __restrict__ model_xyz,
                                                      const int32_t* __restrict__ label, int n_map,
                                                      const double* __restrict__ query_pos, const double* __restrict__ obs_xyz,
                                                      const int32_t* __restrict__ obs_label, int n_obs, int K, double thresh,
                                                      int32_t* __restrict__ out_map_idx, int32_t* __restrict__ scratch) {
  const int q = blockIdx.x;
  AssocCore C;
  C.cloud = cloud; C.model = model_xyz; C.label = label; C.n = n_map; C.K = K;
  C.thresh = thresh; C.best_init = 1000.0; C.label_gate = 1; C.is_cyl = 0;
  C.qpos = query_pos + 3 * (size_t)q;
  C.det_world = obs_xyz + 3 * (size_t)q * n_obs;
  C.det_stride = 3; C.det_off = 0;
  C.det_label = obs_label + (size_t)q * n_obs; C.n_det = n_obs;
  const int Ksub = K < n_map ? K : n_map;
  // per-query scratch: [match_sub (n_obs) | submap (Ksub) | n_sub (1)]
  int32_t* sc = scratch + (size_t)q * (n_obs + Ksub + 1);
  C.match_sub = sc; C.submap = sc + n_obs; C.n_sub = sc + n_obs + Ksub;
  C.match_map = out_map_idx + (size_t)q * n_obs;
  assoc_core(C, dyn_keys, pow2_at_least(n_map));
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
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_assoc_frame), hipFuncAttributeMaxDynamicSharedMemorySize,
                      ASSOC_MAX_N * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_assoc_sweep), hipFuncAttributeMaxDynamicSharedMemorySize,
                      ASSOC_MAX_N * 8);
  g_attr_set = true;
}
static int host_pow2(int n) {
  int p = 64;
  while (p < n) p <<= 1;
  return p;
}

void launch_assoc_frame(const AssocFrameDev* classes3, const double* pose12, int* status, hipStream_t s) {
  ensure_lds_attr();
  // LDS sized for the largest class; the host guarantees n <= ASSOC_MAX_N before calling
  hipLaunchKernelGGL(k_assoc_frame, dim3(3), dim3(1024), ASSOC_MAX_N * 8, s, classes3, pose12, status);
}

static int32_t* g_sweep_scratch = nullptr;
static size_t g_sweep_scratch_n = 0;

void launch_assoc_sweep(const float* cloud, const double* model_xyz, const int32_t* label, int n_map,
                        const double* query_pos, const double* obs_xyz, const int32_t* obs_label, int n_query, int n_obs,
                        int K, double thresh, int32_t* out_map_idx, hipStream_t s) {
  ensure_lds_attr();
  const int Ksub = K < n_map ? K : n_map;
  const size_t need = (size_t)n_query * (n_obs + Ksub + 1);
  if (need > g_sweep_scratch_n) {
    if (g_sweep_scratch) hipFree(g_sweep_scratch);
    hipMalloc(&g_sweep_scratch, need * sizeof(int32_t));
    g_sweep_scratch_n = need;
  }
  hipLaunchKernelGGL(k_assoc_sweep, dim3(n_query), dim3(1024), (size_t)host_pow2(n_map) * 8, s, cloud, model_xyz, label, n_map,
                     query_pos, obs_xyz, obs_label, n_obs, K, thresh, out_map_idx, g_sweep_scratch);
}

void launch_map_refresh(double* cyl_model, int n_cyl, const int* cyl_lid, double* cube_xyz, int n_cube, const int* cube_lid,
                        double* ell_xyz, int n_ell, const int* ell_lid, const double* lm_est, hipStream_t s) {
  const int n = n_cyl + n_cube + n_ell;
  if (n == 0) return;
  hipLaunchKernelGGL(k_map_refresh, dim3((n + 255) / 256), dim3(256), 0, s, cyl_model, n_cyl, cyl_lid, cube_xyz, n_cube,
                     cube_lid, ell_xyz, n_ell, ell_lid, lm_est);
}

}  // namespace sl
