// Inter-robot map-to-map association kernels.
//   SlideMatch sweep: PlaceRecognition::MatchMaps (backend/sloam/src/core/place_recognition.cpp:98-387) —
//   the exhaustive (x, y, yaw) lattice of label-gated inlier counting.  One wavefront per candidate
//   transform; lanes own query objects and scan the reference map (staged in LDS, every lane reads the
//   same entry -> LDS broadcast) in order with the reference's first-hit break; the lattice is
//   produced on the host by the reference's own repeated-addition loops so every candidate value is
//   bit-identical.  FP64 vector ALU / LDS bound: the maps are a few KB, HBM traffic is negligible.
//   CLIPPER affinity: CLIPPER::scorePairwiseConsistency (clipper_semantic_object/src/clipper.cpp:21-65).
// Compiled with -ffp-contract=off (threshold comparisons must round like the reference's x86-64 build).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace sl {

// LDS image: ref objects as 6 doubles [label, x, y, d1, d2, d3]
__global__ __launch_bounds__(256) void k_place_sweep(PlaceDev P, const double* __restrict__ cosv, const double* __restrict__ sinv) {
  extern __shared__ double ref[];
  for (int e = threadIdx.x; e < P.nr * 6; e += blockDim.x) {
    const int k = e / 6, f = e % 6;
    const int src = f == 0 ? 0 : (f <= 2 ? f : f + 1);   // label, x, y, (skip z), d1, d2, d3
    ref[e] = P.ref7[7 * (size_t)k + src];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long ncand = P.n_cells * P.n_yaw;
  for (long long cand = wave; cand < ncand; cand += nwaves) {
    const long long cell = cand / P.n_yaw;
    const int iy = (int)(cand % P.n_yaw);
    const double x = P.xs[P.cell_x[cell]], y = P.ys[P.cell_y[cell]];
    const double c = cosv[iy], s = sinv[iy];
    int inl = 0;
    for (int j = lane; j < P.nq; j += 64) {
      const double* q = P.qry7 + 7 * (size_t)j;
      const double ql = q[0];
      double tx = c * q[1] + (-s) * q[2] + x * 1.0;
      double ty = s * q[1] + c * q[2] + y * 1.0;
      const double tw = 0.0 * q[1] + 0.0 * q[2] + 1.0 * 1.0;
      tx = tx / tw; ty = ty / tw;
      for (int k = 0; k < P.nr; ++k) {
        const double* m = ref + 6 * k;
        if (m[0] != ql) continue;
        const double xd = m[1] - tx, yd = m[2] - ty;
        double avg = 0;
        if (m[4] == 0 && m[5] == 0) {
          avg = fabs(m[3] - q[4]);
        } else {
          avg += fabs(m[3] - q[4]);
          avg += fabs(m[4] - q[5]);
          avg += fabs(m[5] - q[6]);
          avg /= 3;
        }
        const bool dist_ok = sqrt(xd * xd + yd * yd) < P.thr_pos;
        const bool dim_ok = P.ignore_dim ? true : (avg < P.thr_dim);
        if (dist_ok && dim_ok) { ++inl; break; }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) inl += __shfl_xor(inl, off);
    if (lane == 0) P.inliers[cand] = inl;
  }
}

// Round 5: the same count with both maps BUCKETED BY LABEL.  The reference scans, for every query object, the reference objects in
// order, skips those of another label and stops at the first hit (place_recognition.cpp:281-357): a stable bucketing keeps "the first
// hit among the objects of my label" what it was, and the inlier COUNT does not depend on the order of the query objects — so the host
// sorts both maps by label (stably), and a wavefront scans, per chunk of 64 query objects, only the bucket(s) its lanes ask for: no label
// test, a third of the pairs at three labels, LDS broadcast reads ((x, y) as one 16-byte entry), and the distance test
// sqrt(dx^2 + dy^2) < t as dx^2 + dy^2 < v_crit with v_crit = the smallest double whose correctly rounded root is >= t (the same
// decision bit for bit: sqrt is monotone and correctly rounded on both sides) — the f64 square root was most of a pair test's cost.
// Query objects live in LDS too (lane-contiguous), every candidate re-uses them.
__global__ __launch_bounds__(256) void k_place_sweep_b(PlaceDev P, const double* __restrict__ cosv, const double* __restrict__ sinv) {
  extern __shared__ __align__(16) double lds[];
  double* rxy = lds;                                   // 2 nr
  double* rdim = rxy + 2 * (size_t)P.nr;               // 3 nr (only when !ignore_dim)
  double* qxy = rdim + (P.ignore_dim ? 0 : 3 * (size_t)P.nr);      // 2 nq
  double* qdim = qxy + 2 * (size_t)P.nq;               // 3 nq (only when !ignore_dim)
  int* qrange = reinterpret_cast<int*>(qdim + (P.ignore_dim ? 0 : 3 * (size_t)P.nq));   // 2 nq
  for (int e = threadIdx.x; e < 2 * P.nr; e += blockDim.x) rxy[e] = P.rxy[e];
  for (int e = threadIdx.x; e < 2 * P.nq; e += blockDim.x) { qxy[e] = P.qxy[e]; qrange[e] = P.qrange[e]; }
  if (!P.ignore_dim) {
    for (int e = threadIdx.x; e < 3 * P.nr; e += blockDim.x) rdim[e] = P.rdim[e];
    for (int e = threadIdx.x; e < 3 * P.nq; e += blockDim.x) qdim[e] = P.qdim[e];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long ncand = P.n_cells * P.n_yaw;
  const double v_crit = P.v_crit, thr_dim = P.thr_dim;
  const bool use_dim = !P.ignore_dim;
  for (long long cand = wave; cand < ncand; cand += nwaves) {
    const long long cell = cand / P.n_yaw;
    const int iy = (int)(cand % P.n_yaw);
    const double x = P.xs[P.cell_x[cell]], y = P.ys[P.cell_y[cell]];
    const double c = cosv[iy], s = sinv[iy];
    int inl = 0;
    for (int j0 = 0; j0 < P.nq; j0 += 64) {
      const int j = j0 + lane;
      const bool active = j < P.nq;
      double tx = 0.0, ty = 0.0, q4 = 0.0, q5 = 0.0, q6 = 0.0;
      int lo = 0, hi = 0;
      if (active) {
        const double q1 = qxy[2 * j], q2 = qxy[2 * j + 1];
        tx = c * q1 + (-s) * q2 + x * 1.0;
        ty = s * q1 + c * q2 + y * 1.0;
        const double tw = 0.0 * q1 + 0.0 * q2 + 1.0 * 1.0;
        tx = tx / tw; ty = ty / tw;
        lo = qrange[2 * j]; hi = qrange[2 * j + 1];
        if (use_dim) { q4 = qdim[3 * j]; q5 = qdim[3 * j + 1]; q6 = qdim[3 * j + 2]; }
      }
      bool hit = false;
      unsigned long long todo = __ballot(active && hi > lo);
      while (todo) {      // one bucket per chunk (the query objects are sorted by label), two or three where labels meet
        const int first = __ffsll((long long)todo) - 1;
        const int blo = __shfl(lo, first), bhi = __shfl(hi, first);
        const bool mine = active && lo == blo && hi == bhi;
        bool open_ = mine;                                  // still looking for its first hit
        int k = blo;
        if (!use_dim) {
          // four reference objects per round: whether a query object has A hit does not depend on the order inside the bucket, and
          // four independent distance tests hide each other's latencies (one test is a chain of five dependent f64 operations
          // behind an LDS read: 128 cycles per object and wavefront when taken one by one)
          for (; k + 4 <= bhi; k += 4) {
            if (((k - blo) & 15) == 0 && __ballot(open_) == 0) { k = bhi; break; }
            const double x0 = rxy[2 * k] - tx, y0 = rxy[2 * k + 1] - ty, x1 = rxy[2 * k + 2] - tx, y1 = rxy[2 * k + 3] - ty;
            const double x2 = rxy[2 * k + 4] - tx, y2 = rxy[2 * k + 5] - ty, x3 = rxy[2 * k + 6] - tx, y3 = rxy[2 * k + 7] - ty;
            const bool ok = (x0 * x0 + y0 * y0 < v_crit) | (x1 * x1 + y1 * y1 < v_crit) | (x2 * x2 + y2 * y2 < v_crit) | (x3 * x3 + y3 * y3 < v_crit);
            if (open_ && ok) { hit = true; open_ = false; }
          }
        }
        for (; k < bhi; ++k) {
          if (((k - blo) & 15) == 0 && __ballot(open_) == 0) break;
          const double xd = rxy[2 * k] - tx, yd = rxy[2 * k + 1] - ty;
          bool ok = xd * xd + yd * yd < v_crit;
          if (use_dim) {
            const double m3 = rdim[3 * k], m4 = rdim[3 * k + 1], m5 = rdim[3 * k + 2];
            double avg = 0;
            if (m4 == 0 && m5 == 0) {
              avg = fabs(m3 - q4);
            } else {
              avg += fabs(m3 - q4);
              avg += fabs(m4 - q5);
              avg += fabs(m5 - q6);
              avg /= 3;
            }
            ok = ok && (avg < thr_dim);
          }
          if (open_ && ok) { hit = true; open_ = false; }
        }
        todo &= ~__ballot(mine);
      }
      inl += hit ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) inl += __shfl_xor(inl, off);
    if (lane == 0) P.inliers[cand] = inl;
  }
}

// first index of the maximum (the reference keeps a candidate only when it has STRICTLY more inliers)
__global__ __launch_bounds__(256) void k_place_argmax(const int32_t* __restrict__ v, long long n, long long* best_idx,
                                                      int32_t* best_val) {
  __shared__ long long sidx[256];
  __shared__ int sval[256];
  int bv = INT32_MIN;
  long long bi = -1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int x = v[i];
    if (x > bv) { bv = x; bi = i; }
  }
  sval[threadIdx.x] = bv;
  sidx[threadIdx.x] = bi;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) {
      const int ov = sval[threadIdx.x + st];
      const long long oi = sidx[threadIdx.x + st];
      if (oi >= 0 && (ov > sval[threadIdx.x] || (ov == sval[threadIdx.x] && (sidx[threadIdx.x] < 0 || oi < sidx[threadIdx.x])))) {
        sval[threadIdx.x] = ov;
        sidx[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { best_idx[blockIdx.x] = sidx[0]; best_val[blockIdx.x] = sval[0]; }
}

__global__ void k_clipper_affinity(const double* __restrict__ D1, const double* __restrict__ D2, int dim, const int32_t* __restrict__ A,
                                   int m, double sigma, double eps, double mindist, double affinityeps, double* __restrict__ M) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= m || i >= m) return;
  double out = 0.0;
  if (j > i && A[2 * i] != A[2 * j] && A[2 * i + 1] != A[2 * j + 1]) {
    const double* ai = D1 + (size_t)A[2 * i] * dim;
    const double* aj = D1 + (size_t)A[2 * j] * dim;
    const double* bi = D2 + (size_t)A[2 * i + 1] * dim;
    const double* bj = D2 + (size_t)A[2 * j + 1] * dim;
    double s1 = 0, s2 = 0;
    for (int k = 0; k < dim; ++k) {
      s1 += (ai[k] - aj[k]) * (ai[k] - aj[k]);
      s2 += (bi[k] - bj[k]) * (bi[k] - bj[k]);
    }
    const double l1 = sqrt(s1), l2 = sqrt(s2);
    if (!(mindist > 0 && (l1 < mindist || l2 < mindist))) {
      const double c = fabs(l1 - l2);
      const double scr = (c < eps) ? exp(-0.5 * c * c / (sigma * sigma)) : 0.0;
      if (scr > affinityeps) out = scr;
    }
  }
  M[(size_t)i * m + j] = out;
}

// ---- SlideGraph triangle matching (semantic_clipper.cpp:49-118) ------------------------------------------------------
// Per triangle: centroid, the three vertex-centroid distances, stable ascending argsort (the reference's std::sort on
// three indices is an insertion sort), vertices re-ordered accordingly.
__global__ void k_tri_prepare(const double* __restrict__ tri, int n, double* __restrict__ sdist, double* __restrict__ sxy) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const double* v = tri + 6 * (size_t)t;
  const double x0 = v[0], y0 = v[1], x1 = v[2], y1 = v[3], x2 = v[4], y2 = v[5];
  const double cx = (x0 + x1 + x2) / 3.0, cy = (y0 + y1 + y2) / 3.0;
  double d0 = sqrt((x0 - cx) * (x0 - cx) + (y0 - cy) * (y0 - cy));
  double d1 = sqrt((x1 - cx) * (x1 - cx) + (y1 - cy) * (y1 - cy));
  double d2 = sqrt((x2 - cx) * (x2 - cx) + (y2 - cy) * (y2 - cy));
  double ax = x0, ay = y0, bx = x1, by = y1, gx = x2, gy = y2;
  // insertion sort of (d0, d1, d2) with "<" (stable)
  if (d1 < d0) { double t0 = d0; d0 = d1; d1 = t0; t0 = ax; ax = bx; bx = t0; t0 = ay; ay = by; by = t0; }
  if (d2 < d1) {
    double t0 = d1; d1 = d2; d2 = t0; t0 = bx; bx = gx; gx = t0; t0 = by; by = gy; gy = t0;
    if (d1 < d0) { t0 = d0; d0 = d1; d1 = t0; t0 = ax; ax = bx; bx = t0; t0 = ay; ay = by; by = t0; }
  }
  sdist[3 * (size_t)t] = d0; sdist[3 * (size_t)t + 1] = d1; sdist[3 * (size_t)t + 2] = d2;
  double* o = sxy + 6 * (size_t)t;
  o[0] = ax; o[1] = ay; o[2] = bx; o[3] = by; o[4] = gx; o[5] = gy;
}
// One wave per model triangle, lanes over the data triangles.  EMIT = false: counts[i] = matches of row i.
// EMIT = true: rows are written at offs[i] in data order (ballot ranks keep the reference's loop order).
template <bool EMIT>
__global__ __launch_bounds__(256) void k_tri_match(const double* __restrict__ dm, const double* __restrict__ xm, int ntm,
                                                   const double* __restrict__ dd, const double* __restrict__ xd, int ntd, double thr,
                                                   int* __restrict__ counts, const long long* __restrict__ offs,
                                                   double* __restrict__ pts, double* __restrict__ diffs) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= ntm) return;
  const double m0 = dm[3 * (size_t)i], m1 = dm[3 * (size_t)i + 1], m2 = dm[3 * (size_t)i + 2];
  long long base = EMIT ? offs[i] : 0;
  int cnt = 0;
  for (int j0 = 0; j0 < ntd; j0 += 64) {
    const int j = j0 + lane;
    bool hit = false;
    double diff = 0.0;
    if (j < ntd) {
      const double e0 = m0 - dd[3 * (size_t)j], e1 = m1 - dd[3 * (size_t)j + 1], e2 = m2 - dd[3 * (size_t)j + 2];
      double sacc = 0.0;
      sacc += e0 * e0; sacc += e1 * e1; sacc += e2 * e2;
      diff = sqrt(sacc);
      hit = diff < thr;
    }
    const unsigned long long mask = __ballot(hit);
    if (EMIT && hit) {
      const long long row = base + __popcll(mask & ((1ull << lane) - 1ull));
      diffs[row] = diff;
      double* o = pts + 12 * row;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        o[4 * k] = xm[6 * (size_t)i + 2 * k]; o[4 * k + 1] = xm[6 * (size_t)i + 2 * k + 1];
        o[4 * k + 2] = xd[6 * (size_t)j + 2 * k]; o[4 * k + 3] = xd[6 * (size_t)j + 2 * k + 1];
      }
    }
    base += __popcll(mask);
    cnt += __popcll(mask);
  }
  if (!EMIT && lane == 0) counts[i] = cnt;
}

void launch_tri_prepare(const double* tri, int n, double* sdist, double* sxy, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_tri_prepare, dim3((n + 255) / 256), dim3(256), 0, s, tri, n, sdist, sxy);
}
void launch_tri_match(bool emit, const double* dm, const double* xm, int ntm, const double* dd, const double* xd, int ntd, double thr,
                      int* counts, const long long* offs, double* pts, double* diffs, hipStream_t s) {
  if (ntm <= 0) return;
  if (emit) hipLaunchKernelGGL(k_tri_match<true>, dim3((ntm + 3) / 4), dim3(256), 0, s, dm, xm, ntm, dd, xd, ntd, thr, counts, offs, pts, diffs);
  else hipLaunchKernelGGL(k_tri_match<false>, dim3((ntm + 3) / 4), dim3(256), 0, s, dm, xm, ntm, dd, xd, ntd, thr, counts, offs, pts, diffs);
}

void launch_place_sweep(const PlaceDev& P, hipStream_t s) {
  // cos/sin tables ride behind the yaw table: yaws[n_yaw .. 3 n_yaw)
  const double* cosv = P.yaws + P.n_yaw;
  const double* sinv = P.yaws + 2 * (size_t)P.n_yaw;
  const long long ncand = P.n_cells * P.n_yaw;
  long long blocks = (ncand + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (blocks < 1) blocks = 1;
  if (P.rxy) {
    static const bool attr = (hipFuncSetAttribute(reinterpret_cast<const void*>(k_place_sweep_b), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), true);
    (void)attr;
    const size_t lds = ((size_t)(P.ignore_dim ? 2 : 5) * (P.nr + P.nq)) * sizeof(double) + 2 * (size_t)P.nq * sizeof(int) + 16;
    hipLaunchKernelGGL(k_place_sweep_b, dim3((unsigned)blocks), dim3(256), lds, s, P, cosv, sinv);
    return;
  }
  hipLaunchKernelGGL(k_place_sweep, dim3((unsigned)blocks), dim3(256), (size_t)P.nr * 6 * sizeof(double), s, P, cosv, sinv);
}
void launch_place_argmax(const int32_t* inliers, long long n, long long* best_idx, int32_t* best_val, hipStream_t s) {
  hipLaunchKernelGGL(k_place_argmax, dim3(256), dim3(256), 0, s, inliers, n, best_idx, best_val);
}
void launch_clipper_affinity(const double* D1, const double* D2, int dim, const int32_t* A, int m, double sigma, double eps,
                             double mindist, double affinityeps, double* M, hipStream_t s) {
  hipLaunchKernelGGL(k_clipper_affinity, dim3((m + 127) / 128, m), dim3(128), 0, s, D1, D2, dim, A, m, sigma, eps, mindist,
                     affinityeps, M);
}

}  // namespace sl
