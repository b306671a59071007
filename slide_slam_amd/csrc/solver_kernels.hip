// Hand-written gfx950 kernels for the factor-graph update of the SlideSLAM backend
// (reference: SemanticFactorGraph::solve backend/sloam/src/factorgraph/graph.cpp:260-272 ->
// gtsam::ISAM2::update; factor definitions graph.cpp:24-258, cubeFactor.cpp:17-53,
// cylinderFactor.cpp:20-51).  These kernels are HBM/L2-bound gather + small-block arithmetic in
// FP64; the matrix-core work lives in chol_kernels.hip.  All reductions are gathers in a fixed
// order (no floating-point atomics) so results are bit-stable run to run.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>

#include "graph_dev.hpp"
#include "kernels.hpp"
#include "sl_math.hpp"

namespace sl {

struct BufPtrs { double* p[8]; };      // per-robot buffers of a batched launch (blockIdx.z = robot)

// ------------------------------------------------------------------------------------------------
// variable (+) tangent, per variable type
// ------------------------------------------------------------------------------------------------
__device__ inline void pose_retract12(const double* in12, const double* d6, int chart, double* out12) {
  to12(retract(from12(in12), d6, chart), out12);
}
__device__ inline void lm_retract(int type, const double* in, const double* d, int chart, double* out) {
  if (type == VT_POINT) {
    out[0] = in[0] + d[0]; out[1] = in[1] + d[1]; out[2] = in[2] + d[2];
  } else if (type == VT_CUBE) {  // CubeMeasurement::retract cubeFactor.h:95-114
    pose_retract12(in, d, chart, out);
    out[12] = in[12] + d[6]; out[13] = in[13] + d[7]; out[14] = in[14] + d[8];
  } else {  // CylinderMeasurement::retract cylinderFactor.h:59-64: tangent [ray, root, radius]
    out[3] = in[3] + d[0]; out[4] = in[4] + d[1]; out[5] = in[5] + d[2];
    out[0] = in[0] + d[3]; out[1] = in[1] + d[4]; out[2] = in[2] + d[5];
    out[6] = in[6] + d[6];
  }
}

// The lowest pose whose blocks of the reduced system change when variable (pose p | landmark l) moves its linearisation point: the
// pose itself, its relative-pose partners, every pose that observes one of its landmarks (a landmark's H_ll enters the Schur terms of
// all its observers) — reported as the maximum of P - pose in status[6] (incremental re-factorisation, HostGraph::run_update).
__device__ __forceinline__ int mark_dirty_pose(const GraphDev& G, int p) {
  if (!G.lm_first) return 0;
  int cand = p;
  for (int q = G.pose_bt_ptr[p]; q < G.pose_bt_ptr[p + 1]; ++q) {
    const int ent = G.pose_bt[q], b = ent >> 1;
    cand = min(cand, (ent & 1) ? G.bt_i[b] : G.bt_j[b]);
  }
  for (int q = G.pose_ptr[p]; q < G.pose_ptr[p + 1]; ++q) cand = min(cand, G.lm_first[G.pose_lms[q]]);
  return G.P - cand;
}
__device__ __forceinline__ int mark_dirty_lm(const GraphDev& G, int l) {
  return G.lm_first ? G.P - min(G.lm_first[l], G.P - 1) : 0;
}
// [GTSAM ISAM2 relinearisation] theta <- theta (+) delta where |delta|_inf >= threshold.
// Every private array below is indexed by fully unrolled loops only, so the kernel needs no scratch.
__device__ __forceinline__ void k_relin_body(const GraphDev& G) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  int did = 0, dirty = 0;      // per wavefront ONE atomic each at the end: with a zero threshold (Gauss-Newton passes) every variable moves, and
                               // thousands of atomics on two words serialise in L2 (39 us for 1443 variables x 8 robots)
  if (t < G.P) {
    double d[6];
    double mx = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { d[k] = G.pose_delta[6 * (size_t)t + k]; mx = fmax(mx, fabs(d[k])); }
    if (mx >= G.relin_thr) {
      double* v = G.pose_val + 12 * (size_t)t;
      double o[12];
      to12(retract(from12(v), d, G.chart), o);
#pragma unroll
      for (int k = 0; k < 12; ++k) v[k] = o[k];
      did = 1;
      dirty = mark_dirty_pose(G, t);
    }
  } else if (t < G.P + G.L) {
    const int l = t - G.P;
    const int type = G.lm_type[l];
    double* v = G.lm_val + 15 * (size_t)l;
    const double* dl = G.lm_delta + 9 * (size_t)l;
    if (type == VT_POINT) {
      const double d0 = dl[0], d1 = dl[1], d2 = dl[2];
      if (fmax(fmax(fabs(d0), fabs(d1)), fabs(d2)) >= G.relin_thr) {
        v[0] += d0; v[1] += d1; v[2] += d2;
        did = 1;
        dirty = mark_dirty_lm(G, l);
      }
    } else if (type == VT_CUBE) {
      double d[9];
      double mx = 0.0;
#pragma unroll
      for (int k = 0; k < 9; ++k) { d[k] = dl[k]; mx = fmax(mx, fabs(d[k])); }
      if (mx >= G.relin_thr) {
        double o[12];
        to12(retract(from12(v), d, G.chart), o);
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = o[k];
        v[12] += d[6]; v[13] += d[7]; v[14] += d[8];
        did = 1;
        dirty = mark_dirty_lm(G, l);
      }
    } else {
      double d[7];
      double mx = 0.0;
#pragma unroll
      for (int k = 0; k < 7; ++k) { d[k] = dl[k]; mx = fmax(mx, fabs(d[k])); }
      if (mx >= G.relin_thr) {   // tangent order [ray, root, radius]
        v[3] += d[0]; v[4] += d[1]; v[5] += d[2];
        v[0] += d[3]; v[1] += d[4]; v[2] += d[5];
        v[6] += d[6];
        did = 1;
        dirty = mark_dirty_lm(G, l);
      }
    }
  }
  const unsigned long long m = __ballot(did);
  if (m) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dirty = max(dirty, __shfl_xor(dirty, off));
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
      atomicAdd(&G.status[2], __popcll(m));
      if (dirty > 0) atomicMax(&G.status[6], dirty);
    }
  }
}
__global__ void k_relin(GraphDev G) { k_relin_body(G); }
__global__ void k_relin_b(const GraphDev* __restrict__ Gs) { k_relin_body(Gs[blockIdx.z]); }

// ------------------------------------------------------------------------------------------------
// prior / between factors   [GTSAM PriorFactor / BetweenFactor<Pose3>]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void k_lin_pose_factors_body(const GraphDev& G, int t) {
  if (t < G.n_prior) {
    if (G.pr_pose[t] < G.pose0) return;      // (incremental update: nothing this factor depends on moved)
    // r = -Local(x, prior), J = I
    const SE3 X = from12(G.pose_val + 12 * (size_t)G.pr_pose[t]);
    const SE3 Z = from12(G.pr_z + 12 * (size_t)t);
    double l[6];
    local(X, Z, l, G.chart);
#pragma unroll
    for (int k = 0; k < 6; ++k) G.pr_r[6 * t + k] = -l[k] / G.pr_sigma[6 * t + k];
  } else if (t < G.n_prior + G.n_between) {
    const int b = t - G.n_prior;
    if (min(G.bt_i[b], G.bt_j[b]) < G.pose0) return;      // (a relinearised pose dirties its partners: an untouched partner means an untouched factor)
    const SE3 X1 = from12(G.pose_val + 12 * (size_t)G.bt_i[b]);
    const SE3 X2 = from12(G.pose_val + 12 * (size_t)G.bt_j[b]);
    const SE3 Z = from12(G.bt_z + 12 * (size_t)b);
    const SE3 h = between(X1, X2);
    double e[6];
    local(Z, h, e, G.chart);          // r = Local(measured, x1^-1 x2)
    double Ad[36];
    adjoint(between(X2, X1), Ad);     // H1 = -Ad(h^-1), H2 = I
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double w = 1.0 / G.bt_sigma[6 * b + r];
      G.bt_r[6 * b + r] = e[r] * w;
#pragma unroll
      for (int c = 0; c < 6; ++c) G.bt_J0[36 * (size_t)b + 6 * r + c] = -Ad[6 * r + c] * w;
    }
  } else if (t < G.n_prior + G.n_between + G.n_ghost) {
    // Between factor whose other pose is a ghost (constant of this pass): only the local side is linearised
    const int q = t - G.n_prior - G.n_between;
    const SE3 XL = from12(G.pose_val + 12 * (size_t)G.gh_pose[q]);
    const SE3 XO = from12(G.ghost_val + 12 * (size_t)G.gh_slot[q]);
    const SE3 Z = from12(G.gh_z + 12 * (size_t)q);
    const bool first = G.gh_first[q] != 0;
    const SE3 h = first ? between(XL, XO) : between(XO, XL);
    double e[6];
    local(Z, h, e, G.chart);
    double Ad[36];
    adjoint(between(XO, XL), Ad);        // used when the local pose is the first key: H1 = -Ad(h^-1), h^-1 = XO^-1 XL
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double w = 1.0 / G.gh_sigma[6 * q + r];
      G.gh_r[6 * q + r] = e[r] * w;
#pragma unroll
      for (int c = 0; c < 6; ++c) G.gh_J[36 * (size_t)q + 6 * r + c] = first ? -Ad[6 * r + c] * w : (r == c ? w : 0.0);
    }
  }
}
__global__ __launch_bounds__(128) void k_lin_pose_factors(GraphDev G) { k_lin_pose_factors_body(G, blockIdx.x * blockDim.x + threadIdx.x); }
__global__ __launch_bounds__(128) void k_lin_pose_factors_b(const GraphDev* __restrict__ Gs) { k_lin_pose_factors_body(Gs[blockIdx.z], blockIdx.x * blockDim.x + threadIdx.x); }

// ------------------------------------------------------------------------------------------------
// landmark factors
// ------------------------------------------------------------------------------------------------
// CubeFactor::evaluateError cubeFactor.cpp:35:  [Logmap(C^-1 (X M)); s_meas - s_C]
__device__ inline void cube_err(const SE3& X, const SE3& C, const double* cs, const double* z, double* e) {
  const SE3 M = from12(z);
  const SE3 err = compose(inverse(C), compose(X, M));
  se3_log(err, e);
  e[6] = z[12] - cs[0]; e[7] = z[13] - cs[1]; e[8] = z[14] - cs[2];
}
// CylinderFactor::evaluateError cylinderFactor.cpp:35: [q.ray - R ray ; q.root - X root ; radius_meas - q.radius]
__device__ inline void cyl_err(const SE3& X, const double* q, const double* z, double* e) {
  const V3 root = transform_from(X, V3{z[0], z[1], z[2]});
  const V3 ray = mul(X.R, V3{z[3], z[4], z[5]});
  e[0] = q[3] - ray.x; e[1] = q[4] - ray.y; e[2] = q[5] - ray.z;
  e[3] = q[0] - root.x; e[4] = q[1] - root.y; e[5] = q[2] - root.z;
  e[6] = z[6] - q[6];
}

// THIRTY-TWO lanes per factor: the cube and cylinder factors differentiate numerically (central differences with
// delta = 1e-6 through the variables' own retract, as the reference's numericalDerivative21/22 do), i.e. 2 x 15 or 2 x 13
// evaluations of the error function per factor — one per lane (lane = 2 column + sign; lanes 30/31 evaluate the
// unperturbed error), paired by a lane swap.  Every lane runs the same code on its own perturbation (a zero tangent
// retracts to the value itself, exactly), so a thread no longer walks through thirty evaluations one after the other.
// Thread map (round 4): the first nb1 workgroups give every factor ONE thread and linearise the bearing-range factors (analytic
// Jacobians: one lane's work — with 32 lanes per factor a wavefront held two of them, 2 of 64 lanes active, and bearing-range
// factors are most of a SLAM graph); the workgroups behind them give every factor 32 lanes and linearise cubes / cylinders by the
// reference's central differences (one error evaluation per lane) — wavefronts whose two factors are bearing-range leave at once.
__device__ __forceinline__ void k_lin_lf_body(const GraphDev& G, int nb1, int bid) {
  const bool br_region = bid < nb1;
  const int f = br_region ? (int)(bid * 256 + threadIdx.x) : (int)(((bid - nb1) * 256 + threadIdx.x) >> 5);
  const int j = br_region ? 0 : (threadIdx.x & 31);
  if (f >= G.n_lf) return;
  const int type = G.lf_type[f];
  if ((type == FT_BR) != br_region) return;
  const int p = G.lf_pose[f], l = G.lf_lm[f], slot = G.lf_slot[f];
  if (p < G.pose0) return;      // (incremental update: neither this pose nor the landmark moved — a moved landmark dirties its first observer)
  double* out = G.jbuf + G.lf_joff[f];
  const SE3 X = from12(G.pose_val + 12 * (size_t)p);
  const double* lv = G.lm_val + 15 * (size_t)l;
  const int col = j >> 1;                                // 0..14 perturbed tangent component, 15: none
  const double dl = G.numdiff_delta, fac = 1.0 / (2.0 * dl);
  const double d = (j & 1) ? -dl : dl;
  if (type == FT_BR) {
    // [GTSAM BearingRangeFactor<Pose3,Point3>] r = [sphere-local(z_b, b) ; rho - z_rho] / sigma
    const double* z = G.br_z + 4 * (size_t)slot;
    const double w = 1.0 / G.bearing_sigma;
    const V3 q = transform_to(X, V3{lv[0], lv[1], lv[2]});
    const double rho = norm(q);
    const V3 b = (1.0 / rho) * q;
    double e2[2];
    sphere_local(V3{z[0], z[1], z[2]}, b, e2);
    out[0] = e2[0] * w; out[1] = e2[1] * w; out[2] = (rho - z[3]) * w;
    // D_q_pose = [hat(q), -I], D_q_point = R^T ; D_b_q = B(b)^T (I - b b^T)/rho ; D_rho_q = b^T
    V3 b1, b2;
    sphere_basis(b, b1, b2);
    double Dm[9];
    const double bb[3] = {b.x, b.y, b.z}, c1[3] = {b1.x, b1.y, b1.z}, c2[3] = {b2.x, b2.y, b2.z};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double s1 = 0, s2 = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double dn = ((i == j ? 1.0 : 0.0) - bb[i] * bb[j]) / rho;
        s1 += c1[i] * dn; s2 += c2[i] * dn;
      }
      Dm[j] = s1; Dm[3 + j] = s2; Dm[6 + j] = bb[j];
    }
    const M3 Q = hat(q);
    double* Jp = out + 3;
    double* Jl = out + 21;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double s = 0, u = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { s += Dm[3 * r + k] * Q.a[3 * k + j]; u += Dm[3 * r + k] * X.R.a[3 * j + k]; }
        Jp[6 * r + j] = s * w;
        Jp[6 * r + 3 + j] = -Dm[3 * r + j] * w;
        Jl[3 * r + j] = u * w;
      }
  } else if (type == FT_CUBE) {
    // cubeFactor.cpp:41-50; tangent: pose (6), cube pose (6), cube scale (3)
    const double* z = G.cu_z + 15 * (size_t)slot;
    const double* sg = G.cu_sigma + 9 * (size_t)slot;
    const SE3 C = from12(lv);
    double dX[6], dC[6], cs[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) { dX[k] = (col == k) ? d : 0.0; dC[k] = (col == 6 + k) ? d : 0.0; }
#pragma unroll
    for (int k = 0; k < 3; ++k) cs[k] = lv[12 + k] + ((col == 12 + k) ? d : 0.0);
    double e[9];
    cube_err(retract(X, dX, G.chart), retract(C, dC, G.chart), cs, z, e);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const double hx = __shfl(e[i], 30, 32), eo = __shfl_xor(e[i], 1, 32), w = 1.0 / sg[i];
      if (j == 30) out[i] = hx * w;
      if ((j & 1) == 0 && col < 15) {
        const double v = ((e[i] - hx) - (eo - hx)) * fac * w;
        if (col < 6) out[9 + 6 * i + col] = v;           // Jp
        else out[63 + 9 * i + col - 6] = v;               // Jl
      }
    }
  } else {  // FT_CYL
    // cylinderFactor.cpp; tangent: pose (6), CylinderMeasurement [ray(3), root(3), radius] onto value [root, ray, radius]
    const double* z = G.cy_z + 7 * (size_t)slot;
    const double w = 1.0 / G.cyl_sigma;
    const int jl = col - 6;
    const int vi = jl < 0 ? -1 : (jl < 3 ? jl + 3 : (jl < 6 ? jl - 3 : (jl == 6 ? 6 : -1)));
    double dX[6], q[7];
#pragma unroll
    for (int k = 0; k < 6; ++k) dX[k] = (col == k) ? d : 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) q[k] = lv[k] + ((vi == k) ? d : 0.0);
    double e[7];
    cyl_err(retract(X, dX, G.chart), q, z, e);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const double hx = __shfl(e[i], 30, 32), eo = __shfl_xor(e[i], 1, 32);
      if (j == 30) out[i] = hx * w;
      if ((j & 1) == 0 && col < 13) {
        const double v = ((e[i] - hx) - (eo - hx)) * fac * w;
        if (col < 6) out[7 + 6 * i + col] = v;           // Jp
        else out[49 + 7 * i + jl] = v;                    // Jl
      }
    }
  }
}
// nb0 leading workgroups linearise the prior / between / ghost factors (round 5: one launch instead of two — the two kinds of factors do
// not depend on each other, and a launch costs a streaming update ~8 us of its ~280)
__global__ __launch_bounds__(256) void k_lin_lf(GraphDev G, int nb0, int nb1) {
  if ((int)blockIdx.x < nb0) k_lin_pose_factors_body(G, blockIdx.x * 256 + threadIdx.x);
  else k_lin_lf_body(G, nb1, (int)blockIdx.x - nb0);
}
__global__ __launch_bounds__(256) void k_lin_lf_b(const GraphDev* __restrict__ Gs, int nb0, int nb1) {
  if ((int)blockIdx.x < nb0) k_lin_pose_factors_body(Gs[blockIdx.z], blockIdx.x * 256 + threadIdx.x);
  else k_lin_lf_body(Gs[blockIdx.z], nb1, (int)blockIdx.x - nb0);
}

// ------------------------------------------------------------------------------------------------
// landmark reduce: H_ll = sum Jl^T Jl, g_l = sum Jl^T r, H_ll^-1, and per factor
// E = Jp^T Jl, F = E H_ll^-1, u = F g_l.   One WAVEFRONT per landmark: lanes own factors, the D(D+1)/2 + D
// partial sums meet in a fixed xor-shuffle tree (deterministic), every lane inverts H_ll redundantly in
// registers, then each lane finishes its own factors.  (M == D for all three landmark factor kinds.)
// MODE 0: everything (single GPU).  MODE 1: accumulate only, partial sums -> lm_Hacc (54 per landmark:
// packed lower H then g) for the cross-robot all-reduce.  MODE 2: start from the (all-reduced) sums in lm_Hacc.
// MODE 3: everything in one launch as MODE 0, the sums also left in lm_Hacc and separator landmarks treated as in MODE 2 (exact joint
// pass: nothing is exchanged between the two halves, k_border_fill reads the robot's own H_ll from lm_Hacc).
__shared__ double lm_hs[4][96];   // per wave of k_landmark: H_ll^-1 (81) + g_l (9)
constexpr int LM_RED_MAX = 24;    // k_landmark<3>: landmarks with at most that many factors reduce their partial sums through LDS
__shared__ double lm_red[4][54 * LM_RED_MAX];

template <int D, int MODE>
__device__ inline void landmark_wave(const GraphDev& G, int l, int lane) {
  constexpr int NH = D * (D + 1) / 2;
  double h[NH], g[D];
#pragma unroll
  for (int i = 0; i < NH; ++i) h[i] = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) g[i] = 0.0;
  const int f0 = G.lm_ptr[l], nf = G.lm_ptr[l + 1] - f0;
  double* acc = G.lm_Hacc + 54 * (size_t)l;
  if (MODE == 2) {
#pragma unroll
    for (int i = 0; i < NH; ++i) h[i] = acc[i];
#pragma unroll
    for (int i = 0; i < D; ++i) g[i] = acc[45 + i];
  }
  for (int q = lane; MODE != 2 && q < nf; q += 64) {
    const int f = G.lm_fids[f0 + q];
    const double* rec = G.jbuf + G.lf_joff[f];
    const double* Jl = rec + D + 6 * D;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      double row[D];
#pragma unroll
      for (int a = 0; a < D; ++a) row[a] = Jl[D * k + a];
      const double rk = rec[k];
#pragma unroll
      for (int a = 0; a < D; ++a) {
        g[a] += row[a] * rk;
#pragma unroll
        for (int c = 0; c <= a; ++c) h[a * (a + 1) / 2 + c] += row[a] * row[c];
      }
    }
  }
  if (MODE != 2) {
    if (MODE == 3) {
      // the partial sums meet through LDS instead of a six-level shuffle tree over all 64 lanes: in rounds of LM_RED_MAX lanes (one round
      // for the usual landmark with a dozen factors) lane f (a factor) writes its NH + D partials, lane e adds entry e of the round's
      // lanes in lane order, and at the end every lane reads the totals (~130 LDS operations per lane against 6 x 2 x (NH + D) = 648
      // cross-lane ones)
      double* red = lm_red[threadIdx.x >> 6];
      const int nl = nf < 64 ? nf : 64;      // lanes that hold partial sums
      double tot = 0.0;
      for (int base = 0; base < nl; base += LM_RED_MAX) {
        const int cnt = nl - base < LM_RED_MAX ? nl - base : LM_RED_MAX;
        if (lane >= base && lane < base + cnt) {
#pragma unroll
          for (int i = 0; i < NH; ++i) red[i * LM_RED_MAX + lane - base] = h[i];
#pragma unroll
          for (int i = 0; i < D; ++i) red[(NH + i) * LM_RED_MAX + lane - base] = g[i];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (lane < NH + D)
          for (int q = 0; q < cnt; ++q) tot += red[lane * LM_RED_MAX + q];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
      }
      if (lane < NH + D) red[lane] = tot;
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
      for (int i = 0; i < NH; ++i) h[i] = red[i];
#pragma unroll
      for (int i = 0; i < D; ++i) g[i] = red[NH + i];
    } else {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int i = 0; i < NH; ++i) h[i] += __shfl_xor(h[i], off);
#pragma unroll
        for (int i = 0; i < D; ++i) g[i] += __shfl_xor(g[i], off);
      }
    }
  }
  if (MODE == 1 || MODE == 3) {
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NH; ++i) acc[i] = h[i];
#pragma unroll
      for (int i = 0; i < D; ++i) acc[45 + i] = g[i];
    }
    if (MODE == 1) return;
  }
  double* Hinv = G.lm_Hinv + 81 * (size_t)l;
  double* gout = G.lm_g + 9 * (size_t)l;
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < D; ++a) gout[a] = g[a];
  }
  // exact joint step: a separator landmark is not eliminated here — H_ll^-1 = 0 makes F and u vanish (nothing of it enters the pose
  // system), E = Jp^T Jl is still written: it is the landmark's coupling row in the border (k_border_fill)
  const bool sep = (MODE == 2 || MODE == 3) && G.arrow && G.lm_slot && G.lm_slot[l] >= 0;
  if (nf == 0 && (MODE == 0 || MODE == 3)) {
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < D * D; ++i) Hinv[i] = 0.0;
    }
    return;
  }
  // in-register Cholesky H = C C^T, then H^-1 = C^-T C^-1 (all lanes, identical)
  double Cm[D][D];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double s = h[j * (j + 1) / 2 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= Cm[j][k] * Cm[j][k];
    if (!(s > 0.0)) { ok = false; s = 1.0; }
    const double dj = sqrt(s);
    Cm[j][j] = dj;
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      double t = h[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= Cm[i][k] * Cm[j][k];
      Cm[i][j] = t / dj;
    }
  }
  if (!ok && !sep && lane == 0) atomicOr(&G.status[0], 1);
  double Ci[D][D];
#pragma unroll
  for (int c = 0; c < D; ++c) {
    Ci[c][c] = 1.0 / Cm[c][c];
#pragma unroll
    for (int i = c + 1; i < D; ++i) {
      double s = 0.0;
#pragma unroll
      for (int k = c; k < i; ++k) s -= Cm[i][k] * Ci[k][c];
      Ci[i][c] = s / Cm[i][i];
    }
  }
  double Hi[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c <= a; ++c) {
      double s = 0.0;
#pragma unroll
      for (int k = a; k < D; ++k) s += Ci[k][a] * Ci[k][c];
      if (sep) s = 0.0;
      Hi[a][c] = s;
      Hi[c][a] = s;
    }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int c = 0; c < D; ++c) Hinv[a * D + c] = Hi[a][c];
  }
  // E = Jp^T Jl, F = E H^-1, u = F g per factor: one lane per (factor, pose row a) pair — 6 nf items over the 64 lanes instead of
  // one lane per factor with the other ~50 idle; H^-1 and g go through LDS (every lane holds the same copy)
  double* hs = lm_hs[threadIdx.x >> 6];
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < D; ++a) {
#pragma unroll
      for (int c = 0; c < D; ++c) hs[a * D + c] = Hi[a][c];
      hs[81 + a] = g[a];
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the LDS copy is written before any lane reads it
  for (int it = lane; it < 6 * nf; it += 64) {
    const int q = it / 6, a = it - 6 * q;
    const int f = G.lm_fids[f0 + q];
    const double* rec = G.jbuf + G.lf_joff[f];
    const double* Jp = rec + D;
    const double* Jl = rec + D + 6 * D;
    double* E = G.ebuf + G.lf_eoff[f];
    double* F = E + 6 * D;
    double* u = F + 6 * D;
    double Ea[D];
#pragma unroll
    for (int c = 0; c < D; ++c) Ea[c] = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const double jp = Jp[6 * k + a];
#pragma unroll
      for (int c = 0; c < D; ++c) Ea[c] += jp * Jl[D * k + c];
    }
#pragma unroll
    for (int c = 0; c < D; ++c) E[a * D + c] = Ea[c];
    double ua = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double sF = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) sF += Ea[k] * hs[k * D + c];
      F[a * D + c] = sF;
      ua += sF * hs[81 + c];
    }
    u[a] = ua;
  }
}

template <int MODE>
__device__ __forceinline__ void k_landmark_body(const GraphDev& G) {
  const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (l >= G.L) return;
  const int lane = threadIdx.x & 63;
  if (G.pose0 > 0) {
    // incremental update: a landmark no pose >= pose0 observes keeps its sums, inverse and Schur records (a new or moved factor, or
    // the landmark's own relinearisation, would have put its FIRST observer at or above pose0)
    bool any = false;
    for (int q = G.lm_ptr[l] + lane; q < G.lm_ptr[l + 1]; q += 64) any = any || G.lf_pose[G.lm_fids[q]] >= G.pose0;
    if (!__ballot(any)) return;
  }
  const int type = G.lm_type[l];
  if (type == VT_POINT) landmark_wave<3, MODE>(G, l, lane);
  else if (type == VT_CUBE) landmark_wave<9, MODE>(G, l, lane);
  else landmark_wave<7, MODE>(G, l, lane);
}
template <int MODE>
__global__ __launch_bounds__(256) void k_landmark(GraphDev G) { k_landmark_body<MODE>(G); }
template <int MODE>
__global__ __launch_bounds__(256) void k_landmark_b(const GraphDev* __restrict__ Gs) {
  const GraphDev G = Gs[blockIdx.z];
  k_landmark_body<MODE>(G);
}

// ------------------------------------------------------------------------------------------------
// pose reduce: H_pp (6x6) and the already-reduced gradient g_p - sum_f F_f g_l
// ------------------------------------------------------------------------------------------------
// One WAVE per pose, one lane per incident factor (the few priors / odometry factors ride on the first lanes): every lane
// forms the 21 + 6 numbers of its factor's J^T J and J^T r - u in registers, the 64 partial sums meet in an LDS transpose
// and lane e < 42 adds column e and writes entry e of (H_pp | g_p) — so the dependent index -> record -> Jacobian loads of
// all factors of a pose are in flight at once instead of one after the other.
__device__ __forceinline__ void k_pose_body(const GraphDev& G) {
  __shared__ double part[4][27][65];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + wave;
  if (p >= G.P || p < G.pose0) return;      // (incremental update: an earlier pose's H_pp and g_p are the last solve's)
  double H[21], g[6];       // lower triangle, index a (a + 1) / 2 + c, c <= a
#pragma unroll
  for (int i = 0; i < 21; ++i) H[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) g[i] = 0.0;
  const int nbt = G.pose_bt_ptr[p + 1] - G.pose_bt_ptr[p];
  const int nlf = G.pose_ptr[p + 1] - G.pose_ptr[p];
  const int nun = G.n_prior + G.n_ghost;       // unary pose factors (priors, ghost betweens): few, every wave scans them all
  for (int e = lane; e < nun + nbt + nlf; e += 64) {
    if (e < G.n_prior) {
      if (G.pr_pose[e] != p) continue;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const double w = 1.0 / G.pr_sigma[6 * e + k];
        H[k * (k + 1) / 2 + k] += w * w;
        g[k] += w * G.pr_r[6 * e + k];
      }
    } else if (e < nun) {
      const int q = e - G.n_prior;
      if (G.gh_pose[q] != p || (G.arrow && G.gh_bord)) continue;      // (exact joint step: the factor enters through its lambda rows)
      const double* J = G.gh_J + 36 * (size_t)q;
      const double* r = G.gh_r + 6 * (size_t)q;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        double jr[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) jr[a] = J[6 * k + a];
        const double rk = r[k];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          g[a] += jr[a] * rk;
#pragma unroll
          for (int c = 0; c <= a; ++c) H[a * (a + 1) / 2 + c] += jr[a] * jr[c];
        }
      }
    } else if (e < nun + nbt) {
      const int ent = G.pose_bt[G.pose_bt_ptr[p] + e - nun];
      const int b = ent >> 1, role = ent & 1;
      const double* r = G.bt_r + 6 * (size_t)b;
      if (role == 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const double w = 1.0 / G.bt_sigma[6 * b + k];
          H[k * (k + 1) / 2 + k] += w * w;
          g[k] += w * r[k];
        }
      } else {
        const double* J = G.bt_J0 + 36 * (size_t)b;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double jr[6];
#pragma unroll
          for (int a = 0; a < 6; ++a) jr[a] = J[6 * k + a];
          const double rk = r[k];
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            g[a] += jr[a] * rk;
#pragma unroll
            for (int c = 0; c <= a; ++c) H[a * (a + 1) / 2 + c] += jr[a] * jr[c];
          }
        }
      }
    } else {
      const int f = G.pose_fids[G.pose_ptr[p] + e - nun - nbt];
      const int M = lf_rows(G.lf_type[f]);   // square landmark blocks: m == d for all three factor kinds
      const double* rec = G.jbuf + G.lf_joff[f];
      const double* Jp = rec + M;
      for (int k = 0; k < M; ++k) {
        const double rk = rec[k];
        double jr[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) jr[a] = Jp[6 * k + a];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          g[a] += jr[a] * rk;
#pragma unroll
          for (int c = 0; c <= a; ++c) H[a * (a + 1) / 2 + c] += jr[a] * jr[c];
        }
      }
      const double* u = G.ebuf + G.lf_eoff[f] + 12 * M;
#pragma unroll
      for (int a = 0; a < 6; ++a) g[a] -= u[a];
    }
  }
#pragma unroll
  for (int i = 0; i < 21; ++i) part[wave][i][lane] = H[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) part[wave][21 + i][lane] = g[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < 42) {
    // entry e: H[a][c] (e = 6a + c) or g[e - 36]
    int src;
    if (lane < 36) {
      const int a = lane / 6, c = lane % 6;
      src = a >= c ? a * (a + 1) / 2 + c : c * (c + 1) / 2 + a;
    } else {
      src = 21 + lane - 36;
    }
    double sum = 0.0;
#pragma unroll 8
    for (int q = 0; q < 64; ++q) sum += part[wave][src][q];
    if (lane < 36) G.pose_H[36 * (size_t)p + lane] = sum;
    else G.pose_g[6 * (size_t)p + lane - 36] = sum;
  }
}
__global__ __launch_bounds__(256) void k_pose(GraphDev G) { k_pose_body(G); }
__global__ __launch_bounds__(256) void k_pose_b(const GraphDev* __restrict__ Gs) {
  const GraphDev G = Gs[blockIdx.z];
  k_pose_body(G);
}

// ------------------------------------------------------------------------------------------------
// Schur assemble: one WORKGROUP per (pose column j, chunk of 32 row poses i >= j), EIGHT lanes per lower block:
//   S_ij = [i == j] H_pp,i + sum_between J_i^T J_j - sum_{l seen by i and j} F_fa(i) E_fb(j)^T
// The column pose's landmark list is published as an LDS lookup (landmark id -> first list position); the eight lanes
// of a block share out pose i's list (one LDS read per entry, no merge, no pair list is ever materialised), so the
// dependent index -> record loads of the co-observed landmarks run eight abreast, and a three-step butterfly adds the
// partial 6x6 blocks.  Lanes 0..5 of a group then write one column of the block each: consecutive groups write
// consecutive 48-byte runs of the same S column, and every block is written exactly once (S needs no memset, no
// atomics -> bit-stable).  The finished blocks of a workgroup leave through an LDS tile as 1536-byte runs per S column.
extern __shared__ short schur_slot[];

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

constexpr int SCHUR_PJ_CAP = 256;   // entries of the column pose's factor list kept in LDS (longer lists fall back to global reads)

// Pose adjacency for the Schur assembly: bit p of row j = pose p >= j observes one of pose j's landmarks, or shares a relative-pose
// factor with it, or is j itself.  Topology only: rebuilt when the graph changes (HostGraph::upload_new).
__global__ __launch_bounds__(256) void k_pose_adj(GraphDev G) {
  extern __shared__ unsigned adj_lds[];
  const int pj = blockIdx.x, tid = threadIdx.x;
  for (int t = tid; t < G.adj_words; t += 256) adj_lds[t] = 0u;
  __syncthreads();
  const int b0 = G.pose_ptr[pj], nb = G.pose_ptr[pj + 1] - b0;
  // eight lanes per list entry, strided over that landmark's factors
  for (int q = tid >> 3; q < nb; q += 32) {
    const int l = G.pose_lms[b0 + q];
    if (q > 0 && G.pose_lms[b0 + q - 1] == l) continue;
    const int f1 = G.lm_ptr[l + 1];
    for (int f = G.lm_ptr[l] + (tid & 7); f < f1; f += 8) {
      const int p = G.lf_pose[G.lm_fids[f]];
      if (p >= pj) atomicOr(&adj_lds[p >> 5], 1u << (p & 31));
    }
  }
  for (int q = G.pose_bt_ptr[pj] + tid; q < G.pose_bt_ptr[pj + 1]; q += 256) {
    const int ent = G.pose_bt[q];
    const int b = ent >> 1;
    const int other = (ent & 1) ? G.bt_i[b] : G.bt_j[b];
    if (other >= pj) atomicOr(&adj_lds[other >> 5], 1u << (other & 31));
  }
  if (tid == 0) atomicOr(&adj_lds[pj >> 5], 1u << (pj & 31));
  __syncthreads();
  for (int t = tid; t < G.adj_words; t += 256) G.pose_adj[(size_t)pj * G.adj_words + t] = adj_lds[t];
}
void launch_pose_adj(const GraphDev& G, hipStream_t s) {
  if (G.P > 0) hipLaunchKernelGGL(k_pose_adj, dim3(G.P), dim3(256), (size_t)G.adj_words * sizeof(unsigned), s, G);
}

template <bool LISTED>
__device__ __forceinline__ void k_schur_body(const GraphDev& G, int pj, int yb) {
  __shared__ double schur_tile[6][192];
  __shared__ long long pj_ed[SCHUR_PJ_CAP];
  __shared__ int pj_lm[SCHUR_PJ_CAP];
  if (pj >= G.P) return;            // (a batched launch covers the largest graph)
  if (6 * pj + 5 < G.col0) return;  // incremental re-factorisation: this pose's columns hold the factor of the last solve
  const int tid = threadIdx.x;
  // dynamic LDS: landmark -> first entry of pose j's list (shorts), then the bitmap of the poses >= j that share a landmark or a
  // relative-pose factor with pose j — only those blocks of the strip are non-zero, all others are written as zeros unseen
  unsigned* adj = reinterpret_cast<unsigned*>(schur_slot + (G.L + 7) / 8 * 8);
  const int adj_words = (G.P + 31) / 32 + 1;
  constexpr bool listed = LISTED;               // pair lists (HostGraph::build_schur_pairs): nothing to look up
  if (!listed) {
    for (int t = tid; t < G.L; t += 256) schur_slot[t] = -1;
    __syncthreads();
  }
  // (the bitmap depends on the topology only: k_pose_adj builds it once per change of the graph, not in every pass)
  for (int t = tid; t < adj_words; t += 256) adj[t] = G.pose_adj[(size_t)pj * G.adj_words + t];
  const int b0 = G.pose_ptr[pj], nb = G.pose_ptr[pj + 1] - b0;
  if (!listed)
    for (int q = tid; q < nb; q += 256) {
      const int l = G.pose_lms[b0 + q];
      if (q == 0 || G.pose_lms[b0 + q - 1] != l) schur_slot[l] = (short)q;
      if (q < SCHUR_PJ_CAP) { pj_lm[q] = l; pj_ed[q] = G.pose_ed[b0 + q]; }
    }
  __syncthreads();
  const int sub = tid & 7;
  // rows below the profile of this column's tile are structurally zero and stay untouched (zero since the last change of profile)
  int p_end = G.P;
  if (G.prof) {
    const int rows = (G.prof[(6 * pj + 5) / NB] + 1) * NB;      // (the profile is monotone: the later of the column's two tiles)
    p_end = min(G.P, (rows + 5) / 6);
  }
  // the chunks of the strip are dealt out to the gridDim.y workgroups of this pose column (each builds the two tables itself)
  for (int pi0 = pj + 32 * yb; pi0 < p_end; pi0 += 32 * (int)gridDim.y) {
    const int nval = 6 * min(32, G.P - pi0);
    double* Sb = G.S + (size_t)(6 * pj) * G.ld + 6 * (size_t)pi0;
    double* S0b = G.S0 + (size_t)(6 * pj) * G.ld + 6 * (size_t)pi0;      // second copy for the joint solve (save_S0)
    const int w = pi0 >> 5, sh = pi0 & 31;
    const unsigned m = sh ? ((adj[w] >> sh) | (adj[w + 1] << (32 - sh))) : adj[w];      // poses pi0 .. pi0 + 31
    if (m == 0u) {
      for (int e = tid; e < 6 * 192; e += 256) {
        const int c = e / 192, r = e % 192;
        if (r < nval && 6 * pj + c >= G.col0) {
          Sb[(size_t)c * G.ld + r] = 0.0;
          if (G.save_S0) S0b[(size_t)c * G.ld + r] = 0.0;
        }
      }
      continue;
    }
  const int pi = pi0 + (tid >> 3);
  const bool live = pi < G.P && ((m >> (tid >> 3)) & 1u);
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.0;
  if (live) {
    if (sub == 0) {
      if (pi == pj) {
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] = G.pose_H[36 * (size_t)pi + k];
      } else {
        for (int q = G.pose_bt_ptr[pi]; q < G.pose_bt_ptr[pi + 1]; ++q) {
          const int ent = G.pose_bt[q];
          const int b = ent >> 1, role = ent & 1;
          const int other = role ? G.bt_i[b] : G.bt_j[b];
          if (other != pj) continue;
          const double* J = G.bt_J0 + 36 * (size_t)b;
          if (role == 1) {   // pose i is the second key: J_i = diag(w), J_j = J0
#pragma unroll
            for (int a = 0; a < 6; ++a) {
              const double w = 1.0 / G.bt_sigma[6 * b + a];
#pragma unroll
              for (int c = 0; c < 6; ++c) acc[6 * a + c] += w * J[6 * a + c];
            }
          } else {           // pose i is the first key: J_i = J0, J_j = diag(w)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
              const double w = 1.0 / G.bt_sigma[6 * b + c];
#pragma unroll
              for (int a = 0; a < 6; ++a) acc[6 * a + c] += J[6 * c + a] * w;
            }
          }
        }
      }
    }
    if (listed) {
      // own share of the block's pair list, two pairs at a time: F_x E_y^T
      const int q0 = G.sp_idx[2 * ((size_t)pj * G.sp_w + (pi - pj))], q1 = q0 + G.sp_idx[2 * ((size_t)pj * G.sp_w + (pi - pj)) + 1];
      for (int x = q0 + sub; x < q1; x += 8) {
        const long long ex = G.sp_pairs[2 * (size_t)x], ey = G.sp_pairs[2 * (size_t)x + 1];
        const int D = (int)(ex & 15);
        const double* F = G.ebuf + (ex >> 4) + 6 * D;
        const double* E = G.ebuf + (ey >> 4);
        {
          for (int k = 0; k < D; ++k) {
            double fk[6], ek[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) { fk[a] = F[a * D + k]; ek[a] = E[a * D + k]; }
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
              for (int c = 0; c < 6; ++c) acc[6 * a + c] -= fk[a] * ek[c];
          }
        }
      }
    }
    // own share of pose i's list, three entries at a time: first all index loads and LDS look-ups, then the records
    const int a0 = listed ? 0 : G.pose_ptr[pi], a1 = listed ? 0 : G.pose_ptr[pi + 1];
    for (int x0 = a0 + sub; x0 < a1; x0 += 24) {
      int sl[3];
      long long ed[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int x = x0 + 8 * u;
        sl[u] = -1;
        ed[u] = 0;
        if (x < a1) {
          const int l = G.pose_lms[x];
          ed[u] = G.pose_ed[x];
          sl[u] = schur_slot[l];
        }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (sl[u] < 0) continue;
        const int D = (int)(ed[u] & 15);
        const double* F = G.ebuf + (ed[u] >> 4) + 6 * D;
        const int l = (sl[u] < SCHUR_PJ_CAP) ? pj_lm[sl[u]] : G.pose_lms[b0 + sl[u]];
        for (int y = sl[u]; y < nb; ++y) {
          const int ly = (y < SCHUR_PJ_CAP) ? pj_lm[y] : G.pose_lms[b0 + y];
          if (ly != l) break;
          const long long edy = (y < SCHUR_PJ_CAP) ? pj_ed[y] : G.pose_ed[b0 + y];
          const double* E = G.ebuf + (edy >> 4);
          if (D == 3) {
            double e[18];
#pragma unroll
            for (int k = 0; k < 18; ++k) e[k] = E[k];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
              const double f0 = F[3 * a], f1 = F[3 * a + 1], f2 = F[3 * a + 2];
#pragma unroll
              for (int c = 0; c < 6; ++c) acc[6 * a + c] -= f0 * e[3 * c] + f1 * e[3 * c + 1] + f2 * e[3 * c + 2];
            }
          } else {
            for (int k = 0; k < D; ++k) {
              double fk[6], ek[6];
#pragma unroll
              for (int a = 0; a < 6; ++a) { fk[a] = F[a * D + k]; ek[a] = E[a * D + k]; }
#pragma unroll
              for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int c = 0; c < 6; ++c) acc[6 * a + c] -= fk[a] * ek[c];
            }
          }
        }
      }
    }
  }
  // sum over the eight lanes of the group on the vector ALU (DPP: quad butterflies, then the mirrored half row)
#pragma unroll
  for (int k = 0; k < 36; ++k) {
    acc[k] += dpp_f64<0xB1>(acc[k]);     // quad_perm [1,0,3,2]
    acc[k] += dpp_f64<0x4E>(acc[k]);     // quad_perm [2,3,0,1]
    acc[k] += dpp_f64<0x141>(acc[k]);    // row_half_mirror: lane i <-> 7 - i of each eight
  }
  // through an LDS tile to full-line stores: 192 consecutive doubles per S column and workgroup
#pragma unroll
  for (int c = 0; c < 6; ++c)
    if (sub == c) {
#pragma unroll
      for (int a = 0; a < 6; ++a) schur_tile[c][6 * (tid >> 3) + a] = acc[6 * a + c];
    }
  __syncthreads();
  for (int e = tid; e < 6 * 192; e += 256) {
    const int c = e / 192, r = e % 192;
    if (r < nval && 6 * pj + c >= G.col0) {      // (a pose straddling the first dirty tile column: only its columns inside it)
      Sb[(size_t)c * G.ld + r] = schur_tile[c][r];
      if (G.save_S0) S0b[(size_t)c * G.ld + r] = schur_tile[c][r];
    }
  }
  __syncthreads();      // the tile is reused by the next chunk
  }
}
__device__ __forceinline__ void k_pad_rhs_body(const GraphDev& G, int bid);
// workgroups x >= P (y = 0) of the launch write the right-hand-side row and the padding of the last tile (k_pad_rhs_body: they depend on
// k_pose only, like the Schur blocks — round 5: one launch instead of two on the streaming path)
// p_first: the first pose column the launch covers (an incremental update re-assembles from G.col0 on: the pose columns left of it
// would leave at once — 575 of 625 workgroups on a streaming frame — and are not launched)
__global__ __launch_bounds__(256) void k_schur(GraphDev G, int p_first) {
  const int x = (int)blockIdx.x + p_first;
  if (x >= G.P) {
    if (blockIdx.y == 0) k_pad_rhs_body(G, x - G.P);
    return;
  }
  k_schur_body<false>(G, x, blockIdx.y);
}
__global__ __launch_bounds__(256) void k_schur_b(const GraphDev* __restrict__ Gs) {
  const GraphDev G = Gs[blockIdx.z];
  k_schur_body<false>(G, blockIdx.x, blockIdx.y);
}
// the same from pair lists (every graph of the launch has them: launch_phase3_arrow_batched) — a kernel of its own: the walk's tables and
// its three-deep staging would set this one's register budget too
// xcd != 0: the workgroups are renumbered so that the ones the hardware deals to one XCD (linear id mod 8) cover a CONTIGUOUS range of
// (robot, chunk, pose column) — with eight robots, one robot per XCD: a landmark's E / F records are read by every pose column that
// observes it, and with consecutive columns dealt round-robin every XCD's L2 fetched every record (PMC: 182 MB fetched per launch for
// 35 MB of records).  Experiment (SLIDE_SCHUR_XCD=1), measured on C4: assembly 0.507 - 0.510 ms either way — the kernel is bound by
// the latency of its dependent loads per workgroup, not by where they are served from; off by default
__global__ __launch_bounds__(256) void k_schur_lb(const GraphDev* __restrict__ Gs, int xcd) {
  int x = blockIdx.x, y = blockIdx.y, z = blockIdx.z;
  if (xcd) {
    const long long total = (long long)gridDim.x * gridDim.y * gridDim.z;
    const long long lin = x + (long long)gridDim.x * (y + (long long)gridDim.y * z);
    const int r = (int)(lin % 8);
    long long l2 = lin / 8;                        // XCD r holds the ids r, r + 8, ...: (total - r + 7) / 8 of them, a contiguous range
    for (int i = 0; i < r; ++i) l2 += (total - i + 7) / 8;
    x = (int)(l2 % gridDim.x);
    y = (int)((l2 / gridDim.x) % gridDim.y);
    z = (int)(l2 / ((long long)gridDim.x * gridDim.y));
  }
  const GraphDev G = Gs[z];
  k_schur_body<true>(G, x, y);
}

// padding (identity) between 6P and T*NB, and the RHS row (-g) at row T*NB
// One staged host -> device upload scattered to its destinations: descriptor i = (dst pointer, byte offset in the staging
// buffer, byte count), all multiples of 4 bytes (host_graph.hpp UploadBatch)
struct ScatterSeg { unsigned long long dst; unsigned off, bytes; };
__global__ __launch_bounds__(256) void k_scatter(const unsigned char* __restrict__ stage, unsigned desc_off) {
  const ScatterSeg sg = reinterpret_cast<const ScatterSeg*>(stage + desc_off)[blockIdx.x];
  const unsigned* src = reinterpret_cast<const unsigned*>(stage + sg.off);
  unsigned* dst = reinterpret_cast<unsigned*>(sg.dst);
  const unsigned n = sg.bytes >> 2;
  for (unsigned i = blockIdx.y * 256 + threadIdx.x; i < n; i += 256 * gridDim.y) dst[i] = src[i];
}
// all-reduce(sum) between the exchange buffers of the robots that share this GPU: every buffer ends up with the sum
struct SumBcastArgs { int n; double* buf[8]; };
__global__ __launch_bounds__(256) void k_sum_bcast(SumBcastArgs A, int count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  double s = 0.0;
  for (int r = 0; r < A.n; ++r) s += A.buf[r][i];        // fixed order: the result does not depend on arrival order
  for (int r = 0; r < A.n; ++r) A.buf[r][i] = s;
}
void launch_sum_bcast(double* const* bufs, int n, int count, hipStream_t s) {
  if (count <= 0 || n <= 0) return;
  SumBcastArgs A{};
  A.n = n;
  for (int r = 0; r < n; ++r) A.buf[r] = bufs[r];
  hipLaunchKernelGGL(k_sum_bcast, dim3((count + 255) / 256), dim3(256), 0, s, A, count);
}

// status words of the joined graphs: cleared by the first node of a batched pass, gathered into one array by its last (one device ->
// host copy per pass instead of one per robot)
// x0 / x1 (or null): two more status blocks of eight words (the separator's, the lambdas') handled by workgroup 0 in the same launch:
// cleared with the graphs' — or, at the end of a pass, OR-ed into graph 0's flags before the gather (words 4 / 5 are ticket counters)
__global__ void k_status_clear(const GraphDev* __restrict__ Gs, int* x0, int* x1) {
  if (threadIdx.x < 8) {
    Gs[blockIdx.x].status[threadIdx.x] = 0;
    if (blockIdx.x == 0) {
      if (x0) x0[threadIdx.x] = 0;
      if (x1) x1[threadIdx.x] = 0;
    }
  }
}
__global__ void k_status_gather(const GraphDev* __restrict__ Gs, int* __restrict__ out, const int* x0, const int* x1) {
  if (threadIdx.x < 8) {
    int v = Gs[blockIdx.x].status[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x != 4 && threadIdx.x != 5) {
      if (x0) v |= x0[threadIdx.x];
      if (x1) v |= x1[threadIdx.x];
      Gs[0].status[threadIdx.x] = v;
    }
    out[8 * blockIdx.x + threadIdx.x] = v;
  }
}
__global__ void k_ints_clear(int* p, int n) { if ((int)threadIdx.x < n) p[threadIdx.x] = 0; }
__global__ void k_status_or(int* dst, const int* src, int n) { if ((int)threadIdx.x < n && threadIdx.x != 4 && threadIdx.x != 5) dst[threadIdx.x] |= src[threadIdx.x]; }   // (words 4 / 5 are ticket counters)
void launch_ints_clear(int* p, int n, hipStream_t s) { hipLaunchKernelGGL(k_ints_clear, dim3(1), dim3(64), 0, s, p, n); }
void launch_status_or(int* dst, const int* src, int n, hipStream_t s) { hipLaunchKernelGGL(k_status_or, dim3(1), dim3(64), 0, s, dst, src, n); }
void launch_status_clear(const GraphDev* d, int n, hipStream_t s, int* x0, int* x1) {
  if (n > 0) hipLaunchKernelGGL(k_status_clear, dim3(n), dim3(64), 0, s, d, x0, x1);
}
void launch_status_gather(const GraphDev* d, int n, int* out, hipStream_t s, const int* x0, const int* x1) {
  if (n > 0) hipLaunchKernelGGL(k_status_gather, dim3(n), dim3(64), 0, s, d, out, x0, x1);
}

// buf[0] -> buf[1 .. n-1]: hands the result of a cross-GPU all-reduce (done on buf[0]) to the other robots of this GPU
__global__ __launch_bounds__(256) void k_bcast(SumBcastArgs A, int count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const double v = A.buf[0][i];
  for (int r = 1; r < A.n; ++r) A.buf[r][i] = v;
}
void launch_bcast(double* const* bufs, int n, int count, hipStream_t s) {
  if (count <= 0 || n <= 1) return;
  SumBcastArgs A{};
  A.n = n;
  for (int r = 0; r < n; ++r) A.buf[r] = bufs[r];
  hipLaunchKernelGGL(k_bcast, dim3((count + 255) / 256), dim3(256), 0, s, A, count);
}

// n independent small copies dst[i][0 .. count[i]) = src[i][.] in one launch (blockIdx.y = i)
struct CopyPairsArgs { int n; const double* src[8]; double* dst[8]; int count[8]; };
__global__ __launch_bounds__(256) void k_copy_pairs(CopyPairsArgs A) {
  const int r = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i < A.count[r]) A.dst[r][i] = A.src[r][i];
}
void launch_copy_pairs(const double* const* src, double* const* dst, const int* count, int n, hipStream_t s) {
  CopyPairsArgs A{};
  A.n = n;
  int mx = 0;
  for (int r = 0; r < n && r < 8; ++r) { A.src[r] = src[r]; A.dst[r] = dst[r]; A.count[r] = count[r]; mx = std::max(mx, count[r]); }
  if (mx > 0) hipLaunchKernelGGL(k_copy_pairs, dim3((mx + 255) / 256, n), dim3(256), 0, s, A);
}

// the reverse: device arrays gathered into one staging buffer for ONE device -> host copy (DownloadBatch); `dst` of a
// descriptor is the source pointer here
__global__ __launch_bounds__(256) void k_gather(unsigned char* __restrict__ stage, unsigned desc_off) {
  const ScatterSeg sg = reinterpret_cast<const ScatterSeg*>(stage + desc_off)[blockIdx.x];
  unsigned* dst = reinterpret_cast<unsigned*>(stage + sg.off);
  const unsigned* src = reinterpret_cast<const unsigned*>(sg.dst);
  const unsigned n = sg.bytes >> 2;
  for (unsigned i = blockIdx.y * 256 + threadIdx.x; i < n; i += 256 * gridDim.y) dst[i] = src[i];
}
// the same with the descriptors as kernel arguments (up to 16 segments: the per-frame read-backs) — no descriptor copy in front
struct GatherArgs { ScatterSeg seg[16]; };
__global__ __launch_bounds__(256) void k_gather_args(unsigned char* __restrict__ stage, GatherArgs A) {
  const ScatterSeg sg = A.seg[blockIdx.x];
  unsigned* dst = reinterpret_cast<unsigned*>(stage + sg.off);
  const unsigned* src = reinterpret_cast<const unsigned*>(sg.dst);
  const unsigned n = sg.bytes >> 2;
  for (unsigned i = blockIdx.y * 256 + threadIdx.x; i < n; i += 256 * gridDim.y) dst[i] = src[i];
}
void launch_gather_args(void* stage, const void* segs, int nseg, hipStream_t s) {
  GatherArgs A{};
  std::memcpy(A.seg, segs, (size_t)nseg * sizeof(ScatterSeg));
  if (nseg > 0) hipLaunchKernelGGL(k_gather_args, dim3(nseg, 4), dim3(256), 0, s, static_cast<unsigned char*>(stage), A);
}
void launch_gather(void* stage, unsigned desc_off, int nseg, hipStream_t s) {
  if (nseg > 0) hipLaunchKernelGGL(k_gather, dim3(nseg, 4), dim3(256), 0, s, static_cast<unsigned char*>(stage), desc_off);
}
void launch_scatter(const void* stage, unsigned desc_off, int nseg, hipStream_t s) {
  if (nseg > 0) hipLaunchKernelGGL(k_scatter, dim3(nseg, 8), dim3(256), 0, s, static_cast<const unsigned char*>(stage), desc_off);
}

__device__ __forceinline__ void k_pad_rhs_body(const GraphDev& G, int bid) {
  const int n = 6 * G.P, NT = G.T * NB;
  const long long t = (long long)bid * 256 + threadIdx.x;
  if (t < NT) {
    const int c = (int)t;
    // (the right-hand-side row lies below the border rows; columns left of col0 keep the forward-substituted entries of the last solve)
    G.S[(size_t)c * G.ld + NT + (size_t)G.nbr * NB] = c < G.col0 ? G.yv[c] : ((c < n) ? -G.pose_g[c] : 0.0);
    return;                               // (the right-hand-side row is not part of S0: the joint solve takes b from pose_g)
  }
  const long long u = t - NT;
  const int npad = NT - n;
  if (u >= (long long)npad * NT) return;
  const int r = n + (int)(u / NT), c = (int)(u % NT);
  if (c > r || c < G.col0) return;
  if (G.first && c < G.first[G.T - 1] * NB) return;      // left of the profile: zero already
  G.S[(size_t)c * G.ld + r] = (c == r) ? 1.0 : 0.0;
  if (G.save_S0) G.S0[(size_t)c * G.ld + r] = (c == r) ? 1.0 : 0.0;
}
__global__ __launch_bounds__(256) void k_pad_rhs(GraphDev G) { k_pad_rhs_body(G, blockIdx.x); }
__global__ __launch_bounds__(256) void k_pad_rhs_b(const GraphDev* __restrict__ Gs) {
  const GraphDev G = Gs[blockIdx.z];
  k_pad_rhs_body(G, blockIdx.x);
}

// landmark back-substitution  delta_l = -H_ll^-1 (g_l + sum_f E_f^T delta_p), and delta_p = dp.
// MODE 0: everything.  MODE 1: only t_l = sum_f E_f^T delta_p -> lm_t (for the cross-robot all-reduce).
// MODE 2: delta_l from the (all-reduced) t_l.
// One wave per landmark (lanes over its factors, butterfly sum of the <= 9 numbers), four landmarks per workgroup; the
// pose part is a plain copy by the first blocks' threads.
template <int MODE>
__device__ __forceinline__ void k_backsub_body(const GraphDev& G) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (MODE != 2) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < 6 * G.P) G.pose_delta[t] = G.dp[t];
  }
  for (int l = blockIdx.x * 4 + wave; l < G.L; l += gridDim.x * 4) {
    const int D = lm_dim(G.lm_type[l]);
    double rhs[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) rhs[k] = 0.0;
    if (MODE == 2) {
#pragma unroll
      for (int k = 0; k < 9; ++k) if (k < D) rhs[k] = G.lm_t[9 * (size_t)l + k];
    } else {
      for (int q = G.lm_ptr[l] + lane; q < G.lm_ptr[l + 1]; q += 64) {
        const int f = G.lm_fids[q];
        const double* E = G.ebuf + G.lf_eoff[f];
        const double* d = G.dp + 6 * (size_t)G.lf_pose[f];
        double dd[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) dd[a] = d[a];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          if (k < D) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) s += E[a * D + k] * dd[a];
            rhs[k] += s;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) rhs[k] += __shfl_xor(rhs[k], m);
      }
    }
    if (MODE == 1) {
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) G.lm_t[9 * (size_t)l + k] = rhs[k];
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) if (k < D) rhs[k] += G.lm_g[9 * (size_t)l + k];
    if (lane < D) {
      const double* Hi = G.lm_Hinv + 81 * (size_t)l + lane * D;
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 9; ++c) if (c < D) s += Hi[c] * rhs[c];
      G.lm_delta[9 * (size_t)l + lane] = -s;
    }
  }
}
template <int MODE>
__global__ __launch_bounds__(256) void k_backsub(GraphDev G) { k_backsub_body<MODE>(G); }
template <int MODE>
__global__ __launch_bounds__(256) void k_backsub_b(const GraphDev* __restrict__ Gs) { k_backsub_body<MODE>(Gs[blockIdx.z]); }

// ---- cross-robot exchange of shared landmarks (one robot per GPU) --------------------------
// what 0: normal-equation partial sums (54 per slot: packed lower H_ll, g_l), what 1: t_l (9), what 2: the
// landmark VALUE from its owner rank (15; other ranks contribute zeros so that an all-reduce(sum) broadcasts it).
__device__ __forceinline__ void k_shared_pack_body(const GraphDev& G, int what, double* __restrict__ buf) {       // one thread per (slot, element)
  const int w = what == 0 ? 54 : (what == 1 ? 9 : 15);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int sidx = t / w, k = t - sidx * w;
  if (sidx >= G.n_slots) return;
  const int l = G.sh_lid[sidx];
  const bool live = l >= 0 && (what != 2 || G.sh_owner[sidx]);
  const double* src = what == 0 ? G.lm_Hacc + 54 * (size_t)(l < 0 ? 0 : l) : (what == 1 ? G.lm_t + 9 * (size_t)(l < 0 ? 0 : l) : G.lm_val + 15 * (size_t)(l < 0 ? 0 : l));
  buf[t] = live ? src[k] : 0.0;
}
__global__ void k_shared_pack(GraphDev G, int what, double* __restrict__ buf) { k_shared_pack_body(G, what, buf); }
__global__ void k_shared_unpack(GraphDev G, int what, const double* __restrict__ buf) {
  const int w = what == 0 ? 54 : (what == 1 ? 9 : 15);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int sidx = t / w, k = t - sidx * w;
  if (sidx >= G.n_slots) return;
  const int l = G.sh_lid[sidx];
  if (l < 0) return;
  double* dst = what == 0 ? G.lm_Hacc + 54 * (size_t)l : (what == 1 ? G.lm_t + 9 * (size_t)l : G.lm_val + 15 * (size_t)l);
  const int nv = what == 2 ? (G.lm_type[l] == VT_POINT ? 3 : (G.lm_type[l] == VT_CUBE ? 15 : 7)) : w;
  if (k < nv) dst[k] = buf[t];
}

// ---- exact joint step: the border of a robot's system and the separator system of all shared landmarks ------------------------
// k_border_clear: zero the border rows of S (tile rows T .. T + nbr - 1, every band column) and the border x border block.
// k_border_fill: one wave per shared slot this robot observes: its factors' E = Jp^T Jl (6 x D, k_landmark) are added into the
// border rows at the observing pose's columns (sequentially over the landmark's factor list: two factors of one pose add up, no
// atomics), the robot's OWN H_ll block (packed lower in lm_Hacc, k_landmark<1>) goes onto the diagonal of the border block and
// -g_l into its right-hand-side row.
__global__ __launch_bounds__(256) void k_border_clear_b(const GraphDev* __restrict__ Gs) {
  const GraphDev G = Gs[blockIdx.z];
  if (!G.arrow || G.nbr <= 0) return;
  const long long nrow = (long long)G.nbr * NB, ncol = (long long)G.T * NB;
  const long long nb = (long long)(G.nbr + 1) * NB * G.nbr * NB;
  if (G.seg_tab) {
    // a cut band: only the (tile row, block column) pairs some segment works on — a row's tiles from its first block column in a segment to
    // that segment's end; everything else is never written by fill, extraction or steps and holds the zeros of the allocation
    const int* st = G.seg_tab;
    const int nseg = st[0], ntile = G.nbr * G.T;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
      const int t = tile % G.nbr, c = tile / G.nbr;
      bool on = false;
      for (int q = 0; q < nseg; ++q) on = on || (st[1 + nseg + q * (G.nbr + 1) + t] <= c && c < st[1 + q]);
      if (!on) continue;
      double* base = G.S + (size_t)c * NB * G.ld + (size_t)(G.T + t) * NB;
      for (int e = threadIdx.x; e < NB * NB; e += 256) base[(size_t)(e >> 6) * G.ld + (e & 63)] = 0.0;
    }
    (void)nb;      // (the border x border block is not cleared: GraphDev::bord0)
    return;
  }
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < nrow * ncol; t += (long long)gridDim.x * 256)
    G.S[(size_t)(t / nrow) * G.ld + (size_t)G.T * NB + (size_t)(t % nrow)] = 0.0;
}
__global__ __launch_bounds__(256) void k_border_fill_b(const GraphDev* __restrict__ Gs) {
  const GraphDev G = Gs[blockIdx.z];
  if (!G.arrow || G.nbr <= 0) return;
  const int sidx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (sidx >= G.n_slots) return;
  const int l = G.sh_lid[sidx];
  if (l < 0) return;
  const int o = G.lm_bord[l];
  if (o < 0) return;
  const int D = lm_dim(G.lm_type[l]);
  const int a = lane / D, k = lane - a * D;
  if (lane < 6 * D) {
    for (int q = G.lm_ptr[l]; q < G.lm_ptr[l + 1]; ++q) {
      const int f = G.lm_fids[q];
      const double e = G.ebuf[G.lf_eoff[f] + a * D + k];
      double* dst = G.S + (size_t)(6 * G.lf_pose[f] + a) * G.ld + (size_t)G.T * NB + o + k;
      *dst += e;
    }
  }
  const double* acc = G.lm_Hacc + 54 * (size_t)l;
  if (lane < D * (D + 1) / 2) {
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= lane) ++r;
    const int c = lane - r * (r + 1) / 2;
    G.bord0[(size_t)(o + c) * G.ldb + o + r] = acc[lane];
  }
  if (lane < D) G.bord0[(size_t)(o + lane) * G.ldb + (size_t)G.nbr * NB] = -acc[45 + lane];
}
// Separator poses out of the band (nested dissection of the robot's pose chain, graph_dev.hpp pose_sep).  The assembly wrote the
// whole reduced system in pose order; for every separator pose q (one workgroup) its entries move to where a border variable lives:
//   S(r, c), c in q's columns, r >= c:   pose(r) a separator pose too -> bord(o_r, o_c)           (separator x separator block)
//                                        else (a pose of the segment behind) -> border row o_c + ., column r   (transposed)
//   S(r, c), r in q's rows, c < r, pose(c) NOT a separator pose (the segment before) -> border row o_r + ., column c
//   border rows (shared landmarks, lambdas) at q's columns -> bord(row, o_c);   right-hand side at q's columns -> bord's RHS row
// and the band keeps a unit diagonal / zeros / a zero right-hand side there: the segments on both sides no longer couple.  Every entry
// has ONE owner (the column strip owns separator x separator entries), so nothing is read after another workgroup cleared it.
__global__ __launch_bounds__(256) void k_sep_extract_b(const GraphDev* __restrict__ Gs, int max_sep) {
  const GraphDev G = Gs[blockIdx.z];
  if (!G.arrow || !G.pose_sep || G.nsep <= 0) return;
  // the blockIdx.x-th separator pose of this robot
  __shared__ int s_q;
  if (threadIdx.x == 0) s_q = -1;
  __syncthreads();
  for (int p = threadIdx.x; p < G.P; p += 256)      // (pose_sep numbers the separator poses in pose order: offset 6 i for the i-th)
    if (G.pose_sep[p] == 6 * (int)blockIdx.x) s_q = p;
  __syncthreads();
  const int q = s_q;
  if (q < 0) return;
  (void)max_sep;
  const int tid = threadIdx.x, oq = G.pose_sep[q], NT = G.T * NB;
  const size_t ld = G.ld, ldb = G.ldb;
  const size_t brow = (size_t)G.T * NB;                      // first border row of S
  const int nb_rows = G.nbr * NB;                            // border rows (the separator poses' own rows among them: nothing there yet)
  // (1) column strip: rows 6q .. end of the profile of q's last column's tile, plus the border rows and the right-hand side
  const int r_end = G.prof ? min(NT, (G.prof[(6 * q + 5) / NB] + 1) * NB) : NT;
  const int n_strip = r_end - 6 * q;
  // the three strips of a cut pose on three workgroups (blockIdx.y): every entry has one owner, so they need no order among them —
  // one workgroup walking through all of them one after the other is a chain of ~10 global round trips
  const int part = blockIdx.y;
  for (int e = tid; part == 0 && e < 6 * n_strip; e += 256) {             // (consecutive threads walk down one column of S)
    const int a = e / n_strip, r = 6 * q + e % n_strip, c = 6 * q + a;
    if (r < c) continue;
    double* src = G.S + (size_t)c * ld + r;
    const double v = *src;
    const int pr = r / 6;
    if (pr < G.P) {
      const int orr = G.pose_sep[pr];
      if (orr >= 0) G.bord0[(size_t)(oq + a) * ldb + orr + (r - 6 * pr)] = v;                     // separator x separator (o_r >= o_c: lower)
      else G.S[(size_t)r * ld + brow + oq + a] = v;                                             // segment behind: border row of q, column r
    }
    *src = (r == c) ? 1.0 : 0.0;
  }
  for (int e = tid; part == 1 && e < 6 * nb_rows; e += 256) {             // border rows at q's columns -> bord(row, o_c)
    const int a = e / nb_rows, b = e % nb_rows;
    double* src = G.S + (size_t)(6 * q + a) * ld + brow + b;
    const double v = *src;
    // (bord0 is never cleared: an entry that was non-zero in an earlier pass is refreshed even when it is exactly zero now)
    double* dst = G.bord0 + (size_t)(oq + a) * ldb + b;
    if (v != 0.0) { *dst = v; *src = 0.0; }
    else if (b >= G.nsep * NB && *dst != 0.0) *dst = 0.0;      // (rows below nsep * NB are the separator poses' own: part 0 writes them)
  }
  if (part == 0 && tid < 6) {                                // right-hand side
    double* src = G.S + (size_t)(6 * q + tid) * ld + brow + (size_t)G.nbr * NB;
    G.bord0[(size_t)(oq + tid) * ldb + (size_t)G.nbr * NB] = *src;
    *src = 0.0;
  }
  if (part == 0 && blockIdx.x == 0)                          // unit diagonal on the padding of the separator part (whole tiles)
    for (int p = G.nsep_dim + tid; p < G.nsep * NB; p += 256) G.bord0[(size_t)p * ldb + p] = 1.0;
  // (2) row strip: columns from the first column the profile lets reach q's rows, poses that are no separator poses only
  const int c_beg = G.first ? G.first[(6 * q) / NB] * NB : 0;
  for (int e = tid; part == 2 && e < 6 * (6 * q - c_beg); e += 256) {
    const int a = e % 6, c = c_beg + e / 6, r = 6 * q + a;
    const int pc = c / 6;
    if (pc >= G.P || G.pose_sep[pc] >= 0) continue;
    double* src = G.S + (size_t)c * ld + r;
    G.S[(size_t)c * ld + brow + oq + a] = *src;                                                  // border row of q, column c
    *src = 0.0;
  }
}
// the separator poses' solution (the second level's) into the band's solution vector, which holds zeros there
__global__ __launch_bounds__(256) void k_sep_pose_scatter_b(const GraphDev* __restrict__ Gs, BufPtrs X) {
  const GraphDev G = Gs[blockIdx.z];
  if (!G.arrow || !G.pose_sep || G.nsep <= 0) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 6 * G.P) return;
  const int o = G.pose_sep[t / 6];
  if (o >= 0) G.dp[t] = X.p[blockIdx.z][o + t % 6];
}
// lambda rows of the inter-robot relative-pose factors: 36 threads per ghost factor write J (6 x 6, whitened, w.r.t. the own pose) into
// the border rows at the pose's columns; the first-key side also writes -I onto the border block's diagonal and -r into its RHS row
__global__ __launch_bounds__(256) void k_border_fill_lam_b(const GraphDev* __restrict__ Gs) {
  const GraphDev G = Gs[blockIdx.z];
  if (!G.arrow || !G.gh_bord || G.nbr <= 0) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int q = t / 36, e = t - 36 * q;
  if (q >= G.n_ghost) return;
  const int o = G.gh_bord[q];
  if (o < 0) return;
  const int k = e / 6, a = e - 6 * k;
  G.S[(size_t)(6 * G.gh_pose[q] + a) * G.ld + (size_t)G.T * NB + o + k] = G.gh_J[36 * (size_t)q + 6 * k + a];
  if (G.gh_first[q] && a == 0) {
    G.bord0[(size_t)(o + k) * G.ldb + o + k] = -1.0;
    G.bord0[(size_t)(o + k) * G.ldb + (size_t)G.nbr * NB] = -G.gh_r[6 * (size_t)q + k];
  }
}
// separator system of all shared landmarks = sum over the robots of their border blocks after k_border_syrk, gathered through the
// robots' global -> local coordinate maps into the layout the Cholesky kernels factor (column-major lower, ld = (Ts + 1) * NB, the
// right-hand side as first row of tile row Ts); the padding up to Ts * NB gets a unit diagonal.  Fixed summation order.
// packed: the exchange buffer of a job that spans GPUs — tile column j holds its tile rows j .. Ts only (column-major inside, height
// (Ts + 1 - j) * NB), 4096 * Ts (Ts + 3) / 2 doubles instead of the full rectangle; k_sep_unpack copies it into the factorisation's layout
// A dissected layout (hTa < hTL): the tile rows [hTa, hTL) of the tile columns < hTa — leaf b's rows under leaf a's columns — are
// structurally zero and left out: 4096 * (hTL - hTa) * hTa doubles fewer to all-reduce.
__device__ __forceinline__ size_t sep_packed_addr(int row, int col, int Ts, int hTa, int hTL) {
  const int tj = col / NB, cc = col - tj * NB, hb = hTL - hTa;
  const size_t off = (size_t)(NB * NB) * ((size_t)tj * (Ts + 1) - (size_t)tj * (tj - 1) / 2 - (size_t)hb * min(tj, hTa));
  const int height = Ts + 1 - tj - (tj < hTa ? hb : 0);
  int r = row - tj * NB;
  if (tj < hTa && row >= hTL * NB) r -= hb * NB;
  return off + (size_t)cc * ((size_t)height * NB) + (size_t)r;
}
__device__ __forceinline__ bool sep_packed_hole(int row, int col, int hTa, int hTL) { return col < hTa * NB && row >= hTa * NB && row < hTL * NB; }
// Layout of the separator system the kernels below write: Ts tile columns of landmark coordinates (ms real ones, unit diagonal on the
// padding) in `sys` (column-major, ld = (Ts + nl + 1) * NB: band rows, then nl border row tiles = the lambda coordinates' coupling rows,
// then the right-hand side as first row of tile row Ts + nl), and the lambda x lambda block with its right-hand-side row in `bord`
// (ldb = (nl + 1) * NB).  Virtual index of a separator coordinate: landmark g -> g, lambda b -> Ts * NB + b, right-hand side -> Tt * NB,
// Tt = Ts + nl.  The packed exchange layout is sep_packed_addr over the virtual indices with Tt tile columns.
__device__ __forceinline__ double* sep_slot(const SepLayout& Y, int vr, int vc, bool packed) {
  const int Tt = Y.Ts + Y.nl;
  if (packed) return Y.packed + sep_packed_addr(vr, vc, Tt, Y.hTa, Y.hTL);
  if (vc < Y.Ts * NB) return Y.sys + (size_t)vc * ((size_t)(Tt + 1) * NB) + vr;
  return Y.bord + (size_t)(vc - Y.Ts * NB) * ((size_t)(Y.nl + 1) * NB) + (vr - Y.Ts * NB);
}
struct SepGatherArgs {
  int n, packed;
  SepLayout Y;
  const double* bord[8]; int ldb[8]; int nbr[8]; const int* map[8];      // map: ms + lam ints, separator coordinate -> robot's border coordinate or -1
  const int* tmask;      // or null: per virtual tile (Ts landmark + nl lambda tiles), bit r set = robot r holds a coordinate of the tile
  unsigned robot_mask;   // the robots (bits) whose contributions this launch sums
  // per-half partial sums of the TOP block (a whole pass on one GPU, dissected layout): for the columns from split_col on, the robots in
  // mask_b are summed into sys2 / bord2 (the top block's own shape: ld2 rows per column, first row = row split_col; the lambda block like
  // Y.bord) and the others into the system itself — the two halves of the job then subtract their leaf's Schur complement each and are
  // added, exactly as two ranks owning a leaf each would (enqueue_arrow); split_col < 0: off
  int split_col; unsigned mask_b; double* sys2; int ld2; double* bord2;
};
constexpr int SEP_GATHER_COLS = 8;      // columns per workgroup (a workgroup per column: 62 000 workgroups, half of them above the diagonal)
__global__ __launch_bounds__(256) void k_sep_gather(SepGatherArgs A) {
  const SepLayout& Y = A.Y;
  const int vr = blockIdx.x * 256 + threadIdx.x;
  const int NL = Y.Ts * NB, NT = (Y.Ts + Y.nl) * NB;
  if (vr > NT) return;
  const bool rhs = vr == NT;
  const int gr = rhs ? 0 : (vr < NL ? (vr < Y.ms ? vr : -1) : (vr - NL < Y.lam ? Y.ms + vr - NL : -1));
  const unsigned rmask = A.tmask ? (rhs ? ~0u : (unsigned)A.tmask[vr / NB]) : ~0u;
  // this row's coordinate in every robot's border block (the same for all columns of the workgroup: loaded once, not once per column
  // in front of the value it addresses)
  int lrow[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    lrow[r] = -1;
    if (r < A.n && gr >= 0 && ((rmask >> r) & 1u) && ((A.robot_mask >> r) & 1u)) lrow[r] = rhs ? A.nbr[r] * NB : A.map[r][gr];
  }
#pragma unroll 2
  for (int vc = blockIdx.y * SEP_GATHER_COLS; vc < (int)(blockIdx.y + 1) * SEP_GATHER_COLS && vc < NT; ++vc) {
    if (!rhs && vr < vc) continue;
    // (the block between the leaves of a dissected layout: nobody ever writes it, in either layout — zero since allocation)
    if (sep_packed_hole(vr, vc, Y.hTa, Y.hTL)) continue;
    // separator coordinate of a virtual index, or -1 on the padding
    const int gc = vc < NL ? (vc < Y.ms ? vc : -1) : (vc - NL < Y.lam ? Y.ms + vc - NL : -1);
    double s = 0.0, s2 = 0.0;
    if (gc < 0 || gr < 0) {
      s = (vr == vc && vc < NL) ? 1.0 : 0.0;      // unit diagonal on the landmark padding (the lambda padding is set by k_lam_prepare)
    } else if (vr == vc && ((vc >= Y.gap[0] && vc < Y.gap[1]) || (vc >= Y.gap[2] && vc < Y.gap[3]))) {
      s = A.packed ? 0.0 : 1.0;                   // padding between the blocks of a dissected layout (a packed partial sum gets it in k_sep_unpack)
    } else {
      // only the robots that hold coordinates of BOTH tiles can contribute (two or three of eight on a grid of robot cells)
      // The robots' contributions are added along a BINARY TREE over the robot index — ((0 + 1) + (2 + 3)) + ((4 + 5) + (6 + 7)) — whatever
      // the number of robots on this GPU: a rank that holds an aligned power-of-two range of the job's robots computes a subtree of the
      // same tree, the pairwise exchanges between the ranks (distributed.py) its upper levels, and the job's sums are bit for bit the same
      // at 1, 2, 4 and 8 ranks (an absent contribution is an exact zero; SURVEY 7, hard part 5).  A.first_robot masks the range to sum.
      const unsigned cand = (A.tmask ? (unsigned)A.tmask[vc / NB] & rmask : ~0u) & A.robot_mask;
      double v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        v[r] = 0.0;
        if (r < A.n && ((cand >> r) & 1u) && lrow[r] >= 0) {
          const int lc = A.map[r][gc];
          const int lr = lrow[r];
          if (lc >= 0) v[r] = A.bord[r][(size_t)min(lr, lc) * A.ldb[r] + max(lr, lc)];      // (lower triangle of the robot's block)
        }
      }
      if (A.split_col >= 0 && vc >= A.split_col) {
        double va[8], vb[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { const bool b = (A.mask_b >> r) & 1u; va[r] = b ? 0.0 : v[r]; vb[r] = b ? v[r] : 0.0; }
        s = ((va[0] + va[1]) + (va[2] + va[3])) + ((va[4] + va[5]) + (va[6] + va[7]));
        s2 = ((vb[0] + vb[1]) + (vb[2] + vb[3])) + ((vb[4] + vb[5]) + (vb[6] + vb[7]));
      } else {
        s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
      }
    }
    *sep_slot(Y, vr, vc, A.packed != 0) = s;
    if (A.split_col >= 0 && vc >= A.split_col) {      // the other half's partial sum (zero on the padding)
      if (vc < NL) A.sys2[(size_t)(vc - A.split_col) * A.ld2 + (vr - A.split_col)] = s2;
      else A.bord2[(size_t)(vc - NL) * ((size_t)(Y.nl + 1) * NB) + (vr - NL)] = s2;
    }
  }
}
// c0, c1: the (virtual) tile columns [c0, c1) only; to_packed: the other way (the factorisation's layout -> the packed buffer: a rank that
// owns a leaf of a dissected layout hands its top block back for the exchange)
__global__ __launch_bounds__(256) void k_sep_unpack(SepLayout Y, int c0, int c1, int to_packed) {
  const int vr = blockIdx.x * 256 + threadIdx.x, vc = c0 * NB + blockIdx.y;
  const int NT = (Y.Ts + Y.nl) * NB;
  if (vr > NT || vc >= NT || vc >= c1 * NB || vr < vc / NB * NB) return;
  if (sep_packed_hole(vr, vc, Y.hTa, Y.hTL)) return;      // (never written in the factorisation's layout either: zero since allocation)
  if (to_packed) { *sep_slot(Y, vr, vc, true) = *sep_slot(Y, vr, vc, false); return; }
  double v = *sep_slot(Y, vr, vc, true);
  if (vr == vc && ((vc >= Y.gap[0] && vc < Y.gap[1]) || (vc >= Y.gap[2] && vc < Y.gap[3]))) v = 1.0;
  *sep_slot(Y, vr, vc, false) = v;
}
// the lambda coordinates' own system after the landmark part is eliminated: bord holds K22 - L21 L21^T (negative definite) and
// r2 - L21 z1; M = -(that) is positive definite and M lambda = -(r2 - L21 z1).  out: Tl = nl tile columns, ld = (nl + 1) * NB, unit
// diagonal on the padding.
__global__ __launch_bounds__(256) void k_lam_prepare(const double* __restrict__ bord, int nl, int lam, double* __restrict__ out) {
  const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  const int NT = nl * NB, ld = (nl + 1) * NB;
  if (r > NT || c >= NT) return;
  const bool rhs = r == NT;
  if (!rhs && r < c) return;
  double v;
  if (c < lam && (rhs || r < lam)) v = -bord[(size_t)c * ld + r];
  else v = (r == c) ? 1.0 : 0.0;
  out[(size_t)c * ld + r] = v;
}
// the separator's solution back to the robots: x_loc (border order, for k_border_apply) and, after the landmark back-substitution,
// the shared landmarks' own deltas
struct SepScatterArgs { int n, m, ms; const double* xs; const double* xl; double* xloc[8]; const int* map[8]; };
__global__ __launch_bounds__(256) void k_sep_xloc(SepScatterArgs A) {
  const int g = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (g >= A.m || r >= A.n) return;
  const int lc = A.map[r][g];
  if (lc >= 0) A.xloc[r][lc] = g < A.ms ? A.xs[g] : A.xl[g - A.ms];
}
__global__ __launch_bounds__(256) void k_sep_lm_delta_b(const GraphDev* __restrict__ Gs, SepScatterArgs A, const int* __restrict__ sep_off) {
  const GraphDev G = Gs[blockIdx.z];
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int sidx = t / 9, k = t - 9 * sidx;
  if (sidx >= G.n_slots) return;
  const int l = G.sh_lid[sidx];
  if (l < 0) return;
  if (k < lm_dim(G.lm_type[l])) G.lm_delta[9 * (size_t)l + k] = A.xs[sep_off[sidx] + k];
}
// calculateEstimate(): theta (+) delta
// what 0: buf[12 s ..] = estimate of the local pose owning ghost slot s (zeros when another rank owns it)
// what 1: ghost_val <- buf (after the all-reduce every slot holds its owner's pose)
__global__ void k_ghost_exchange(GraphDev G, int what, double* __restrict__ buf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 12 * G.n_gslots) return;
  if (what == 0) {
    const int p = G.gslot_pose[t / 12];
    buf[t] = p >= 0 ? G.pose_est[12 * (size_t)p + t % 12] : 0.0;
  } else {
    G.ghost_val[t] = buf[t];
  }
}

// the same for all robots of a batched pass in one launch (blockIdx.z = robot): sixteen 2-us launches per pass otherwise
struct GhostBufs { double* p[8]; };
__global__ void k_ghost_exchange_b(const GraphDev* __restrict__ Gs, int what, GhostBufs B) {
  const GraphDev G = Gs[blockIdx.z];
  double* buf = B.p[blockIdx.z];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 12 * G.n_gslots) return;
  if (what == 0) {
    const int p = G.gslot_pose[t / 12];
    buf[t] = p >= 0 ? G.pose_est[12 * (size_t)p + t % 12] : 0.0;
  } else {
    G.ghost_val[t] = buf[t];
  }
}
// whole pass on one GPU: every robot adopts the owners' current estimates directly (pack, sum over the robots in their order, adopt
// in one launch instead of three)
__global__ void k_ghost_refresh_local(const GraphDev* __restrict__ Gs, int n) {
  const GraphDev G = Gs[blockIdx.z];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 12 * G.n_gslots) return;
  double v = 0.0;
  for (int q = 0; q < n; ++q) {
    const int p = Gs[q].gslot_pose[t / 12];
    v += p >= 0 ? Gs[q].pose_est[12 * (size_t)p + t % 12] : 0.0;
  }
  G.ghost_val[t] = v;
}
void launch_ghost_refresh_local(const GraphDev* d, int n, int n_gslots, hipStream_t s) {
  if (n_gslots <= 0 || n <= 0) return;
  hipLaunchKernelGGL(k_ghost_refresh_local, dim3((12 * n_gslots + 127) / 128, 1, n), dim3(128), 0, s, d, n);
}
void launch_ghost_exchange_batched(const GraphDev* d, int n, int n_gslots, int what, double* const* bufs, hipStream_t s) {
  if (n_gslots <= 0 || n <= 0) return;
  GhostBufs B{};
  for (int i = 0; i < n; ++i) B.p[i] = bufs[i];
  hipLaunchKernelGGL(k_ghost_exchange_b, dim3((12 * n_gslots + 127) / 128, 1, n), dim3(128), 0, s, d, what, B);
}

__device__ __forceinline__ void k_estimate_body(const GraphDev& G) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < G.P) {
    pose_retract12(G.pose_val + 12 * (size_t)t, G.pose_delta + 6 * (size_t)t, G.chart, G.pose_est + 12 * (size_t)t);
  } else if (t < G.P + G.L) {
    const int l = t - G.P;
    lm_retract(G.lm_type[l], G.lm_val + 15 * (size_t)l, G.lm_delta + 9 * (size_t)l, G.chart, G.lm_est + 15 * (size_t)l);
  }
}
__global__ void k_estimate(GraphDev G) { k_estimate_body(G); }
// Streaming updates (round 5): k_estimate that also PREDICTS the next update's relinearisation — delta does not change between two
// solves, so the variables k_relin will move next time (|delta|_inf >= threshold) are known now; the lowest pose whose blocks they
// change (mark_dirty_pose / mark_dirty_lm, as k_relin reports it in status[6]) goes to status[5] as P - pose.  The host reads it with
// this update's status words and needs no read-back between k_relin and the rest of the next update.
__global__ void k_estimate_predict(GraphDev G, int pose, double* __restrict__ out) {
  k_estimate_body(G);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  int dirty = 0;
  if (t < G.P) {
    double mx = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) mx = fmax(mx, fabs(G.pose_delta[6 * (size_t)t + k]));
    if (mx >= G.relin_thr) dirty = mark_dirty_pose(G, t);
  } else if (t < G.P + G.L) {
    const int l = t - G.P;
    const int d = lm_dim(G.lm_type[l]);
    double mx = 0.0;
    for (int k = 0; k < d; ++k) mx = fmax(mx, fabs(G.lm_delta[9 * (size_t)l + k]));
    if (mx >= G.relin_thr) dirty = mark_dirty_lm(G, l);
  }
  if (__ballot(dirty > 0)) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dirty = max(dirty, __shfl_xor(dirty, off));
    if ((threadIdx.x & 63) == 0 && dirty > 0) atomicMax(&G.status[5], dirty);
  }
  // the closing pack (k_final_pack) by the LAST workgroup to get here (round 5: one launch less per streaming update): out[16] is the
  // arrival counter (zero before the launch; the last workgroup leaves it at zero)
  if (out) {
    __shared__ int s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      int* cnt = reinterpret_cast<int*>(out + 16);
      const int a = atomicAdd(cnt, 1);
      s_last = (a == (int)gridDim.x - 1) ? 1 : 0;
      if (s_last) *cnt = 0;
    }
    __syncthreads();
    if (s_last) {
      __threadfence();
      const int tt = threadIdx.x;
      if (tt < 8) {
        reinterpret_cast<int*>(out)[tt] = __hip_atomic_load(&G.status[tt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        G.status[tt] = 0;
      } else if (tt < 20) {
        out[4 + (tt - 8)] = pose >= 0 ? __hip_atomic_load(&G.pose_est[12 * (size_t)pose + (tt - 8)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
    }
  }
}
// the update's closing read-back in ONE piece: the eight status words and the estimate of pose `pose` (the newest key frame, what a frame
// returns) side by side; the status words are left at zero for the next update (its k_relin counts into them)
__global__ void k_final_pack(GraphDev G, int pose, double* __restrict__ out) {
  const int t = threadIdx.x;
  if (t < 8) {
    reinterpret_cast<int*>(out)[t] = G.status[t];
    G.status[t] = 0;
  } else if (t < 20) {
    out[4 + (t - 8)] = pose >= 0 ? G.pose_est[12 * (size_t)pose + (t - 8)] : 0.0;
  }
}
__global__ void k_estimate_b(const GraphDev* __restrict__ Gs) { k_estimate_body(Gs[blockIdx.z]); }

// sum of squared whitened residuals of every factor at the last linearisation point (NonlinearFactorGraph::error x 2): one workgroup,
// fixed summation order.  out[0] = total, out[1] = priors, out[2] = betweens (incl. ghost), out[3] = landmark factors.
__global__ __launch_bounds__(256) void k_chi2(GraphDev G, double* __restrict__ out) {
  __shared__ double sh[3][256];
  const int tid = threadIdx.x;
  double a = 0.0, b = 0.0, c = 0.0;
  for (int i = tid; i < 6 * G.n_prior; i += 256) a += G.pr_r[i] * G.pr_r[i];
  for (int i = tid; i < 6 * G.n_between; i += 256) b += G.bt_r[i] * G.bt_r[i];
  for (int i = tid; i < 6 * G.n_ghost; i += 256) b += G.gh_r[i] * G.gh_r[i];
  for (int f = tid; f < G.n_lf; f += 256) {
    const int M = lf_rows(G.lf_type[f]);
    const double* r = G.jbuf + G.lf_joff[f];
    for (int k = 0; k < M; ++k) c += r[k] * r[k];
  }
  sh[0][tid] = a; sh[1][tid] = b; sh[2][tid] = c;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) { sh[0][tid] += sh[0][tid + st]; sh[1][tid] += sh[1][tid + st]; sh[2][tid] += sh[2][tid + st]; }
    __syncthreads();
  }
  if (tid == 0) { out[1] = sh[0][0]; out[2] = sh[1][0]; out[3] = sh[2][0]; out[0] = sh[0][0] + sh[1][0] + sh[2][0]; }
}
void launch_chi2(const GraphDev& G, double* out4, hipStream_t s) { hipLaunchKernelGGL(k_chi2, dim3(1), dim3(256), 0, s, G, out4); }

// ------------------------------------------------------------------------------------------------
static inline unsigned blocks_for(long long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

void init_solver_kernels() {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_schur_b), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_schur_lb), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_schur), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);   // 60000 landmarks (the capacity check in HostGraph::upload_new) beside 12 KB of static LDS
}
void launch_relin(const GraphDev& G, hipStream_t s) {
  if (G.P + G.L == 0) return;
  hipLaunchKernelGGL(k_relin, dim3(blocks_for(G.P + G.L, 256)), dim3(256), 0, s, G);
}
void launch_linearize(const GraphDev& G, hipStream_t s) {
  const int npf = G.n_prior + G.n_between + G.n_ghost;
  if (G.n_lf > 0) {
    const int nb0 = npf > 0 ? (int)blocks_for(npf, 256) : 0;
    hipLaunchKernelGGL(k_lin_lf, dim3(nb0 + blocks_for(G.n_lf, 256) + blocks_for(32LL * G.n_lf, 256)), dim3(256), 0, s, G, nb0, (int)blocks_for(G.n_lf, 256));
  } else if (npf > 0) {
    hipLaunchKernelGGL(k_lin_pose_factors, dim3(blocks_for(npf, 128)), dim3(128), 0, s, G);
  }
}
void launch_landmark(const GraphDev& G, int mode, hipStream_t s) {
  if (G.L == 0) return;
  if (mode == 0) hipLaunchKernelGGL(k_landmark<0>, dim3(blocks_for(G.L, 4)), dim3(256), 0, s, G);
  else if (mode == 1) hipLaunchKernelGGL(k_landmark<1>, dim3(blocks_for(G.L, 4)), dim3(256), 0, s, G);
  else hipLaunchKernelGGL(k_landmark<2>, dim3(blocks_for(G.L, 4)), dim3(256), 0, s, G);
}
void launch_pose(const GraphDev& G, hipStream_t s) {
  if (G.P > 0) hipLaunchKernelGGL(k_pose, dim3(blocks_for(G.P, 4)), dim3(256), 0, s, G);
}
void launch_schur(const GraphDev& G, hipStream_t s) {
  if (G.P == 0) return;
  static const int env_split = getenv("SLIDE_SCHUR_SPLIT") ? atoi(getenv("SLIDE_SCHUR_SPLIT")) : 0;     // diagnostic
  const int split = env_split > 0 ? env_split : (G.schur_split > 0 ? G.schur_split : 2);
  const long long NT = (long long)G.T * NB;
  const long long tot = NT + (NT - 6LL * G.P) * NT;
  const int p_first = G.col0 > 5 ? std::min((G.col0 - 5 + 5) / 6, G.P) : 0;      // smallest pj with 6 pj + 5 >= col0
  hipLaunchKernelGGL(k_schur, dim3((unsigned)(G.P - p_first) + (unsigned)blocks_for(tot, 256), split > 0 ? split : 1), dim3(256),
                     (size_t)((G.L + 7) / 8 * 8) * sizeof(short) + (size_t)((G.P + 31) / 32 + 1) * sizeof(unsigned), s, G, p_first);
}
void launch_backsub(const GraphDev& G, int mode, hipStream_t s) {
  if (G.P + G.L == 0) return;
  const int nb = std::max(blocks_for(G.L, 4), blocks_for(6 * G.P, 256));
  if (mode == 0) hipLaunchKernelGGL(k_backsub<0>, dim3(nb), dim3(256), 0, s, G);
  else if (mode == 1) hipLaunchKernelGGL(k_backsub<1>, dim3(nb), dim3(256), 0, s, G);
  else hipLaunchKernelGGL(k_backsub<2>, dim3(nb), dim3(256), 0, s, G);
}
__global__ void k_shared_unpack_b(const GraphDev* __restrict__ Gs, int what, BufPtrs B) {
  const GraphDev G = Gs[blockIdx.z];
  const int w = what == 0 ? 54 : (what == 1 ? 9 : 15);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int sidx = t / w, k = t - sidx * w;
  if (sidx >= G.n_slots) return;
  const int l = G.sh_lid[sidx];
  if (l < 0) return;
  double* dst = what == 0 ? G.lm_Hacc + 54 * (size_t)l : (what == 1 ? G.lm_t + 9 * (size_t)l : G.lm_val + 15 * (size_t)l);
  const int nv = what == 2 ? (G.lm_type[l] == VT_POINT ? 3 : (G.lm_type[l] == VT_CUBE ? 15 : 7)) : w;
  if (k < nv) dst[k] = B.p[blockIdx.z][t];
}
// Phase 1 up to the factorisation (unpack the exchanged H_ll / g_l, invert, Schur records, pose blocks, reduced system) for ALL the
// robots of a GPU in five launches: blockIdx.z = robot, every grid sized for the largest graph.  h[i] = host copy of d[i].
void launch_phase3_batched(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, hipStream_t s) {
  int L = 0, P = 0, slots = 0;
  long long pad = 0;
  for (int i = 0; i < n; ++i) {
    L = std::max(L, h[i].L); P = std::max(P, h[i].P); slots = std::max(slots, h[i].n_slots);
    const long long NT = (long long)h[i].T * NB;
    pad = std::max(pad, NT + (NT - 6LL * h[i].P) * NT);
  }
  BufPtrs B{};
  for (int i = 0; i < n; ++i) B.p[i] = bufs[i];
  if (slots > 0) hipLaunchKernelGGL(k_shared_unpack_b, dim3(blocks_for(54LL * slots, 128), 1, n), dim3(128), 0, s, d, 0, B);
  if (L > 0) hipLaunchKernelGGL(k_landmark_b<2>, dim3(blocks_for(L, 4), 1, n), dim3(256), 0, s, d);
  if (P > 0) {
    hipLaunchKernelGGL(k_pose_b, dim3(blocks_for(P, 4), 1, n), dim3(256), 0, s, d);
    static const int env_split = getenv("SLIDE_SCHUR_SPLIT") ? atoi(getenv("SLIDE_SCHUR_SPLIT")) : 0;
    int split = 1;
    for (int i = 0; i < n; ++i) split = std::max(split, h[i].schur_split > 0 ? h[i].schur_split : 2);
    if (env_split > 0) split = env_split;
    hipLaunchKernelGGL(k_schur_b, dim3(P, split > 0 ? split : 1, n), dim3(256),
                       (size_t)((L + 7) / 8 * 8) * sizeof(short) + (size_t)((P + 31) / 32 + 1) * sizeof(unsigned), s, d);
    hipLaunchKernelGGL(k_pad_rhs_b, dim3(blocks_for(pad, 256), 1, n), dim3(256), 0, s, d);
  }
}
// the same for an exact joint pass: no exchanged sums to unpack (a separator landmark keeps the robot's own H_ll / g_l), then the border
void launch_phase3_arrow_batched(const GraphDev* d, const GraphDev* h, int n, hipStream_t s) {
  int L = 0, P = 0;
  long long pad = 0;
  for (int i = 0; i < n; ++i) {
    L = std::max(L, h[i].L); P = std::max(P, h[i].P);
    const long long NT = (long long)h[i].T * NB;
    pad = std::max(pad, NT + (NT - 6LL * h[i].P) * NT);
  }
  if (L > 0) hipLaunchKernelGGL(k_landmark_b<3>, dim3(blocks_for(L, 4), 1, n), dim3(256), 0, s, d);
  if (P > 0) {
    hipLaunchKernelGGL(k_pose_b, dim3(blocks_for(P, 4), 1, n), dim3(256), 0, s, d);
    int split = 1;
    bool listed = true;
    for (int i = 0; i < n; ++i) { split = std::max(split, h[i].schur_split > 0 ? h[i].schur_split : 2); listed = listed && h[i].sp_idx != nullptr; }
    const size_t lds = (size_t)((L + 7) / 8 * 8) * sizeof(short) + (size_t)((P + 31) / 32 + 1) * sizeof(unsigned);
    static const int schur_xcd = getenv("SLIDE_SCHUR_XCD") ? atoi(getenv("SLIDE_SCHUR_XCD")) : 0;
    if (listed) hipLaunchKernelGGL(k_schur_lb, dim3(P, split > 0 ? split : 1, n), dim3(256), lds, s, d, schur_xcd);
    else hipLaunchKernelGGL(k_schur_b, dim3(P, split > 0 ? split : 1, n), dim3(256), lds, s, d);
    hipLaunchKernelGGL(k_pad_rhs_b, dim3(blocks_for(pad, 256), 1, n), dim3(256), 0, s, d);
  }
  launch_border_assemble_batched(d, h, n, s);
}
__global__ void k_shared_pack_b(const GraphDev* __restrict__ Gs, int what, BufPtrs B) { k_shared_pack_body(Gs[blockIdx.z], what, B.p[blockIdx.z]); }
// The per-robot phases of a batched pass in ONE launch sequence for all robots (blockIdx.z = robot, grids sized for the largest graph):
// forking every robot's phase onto its own stream and joining again cost three cross-stream joins of ~15 us per pass.
// phase 0: relinearise, linearise, per-landmark partial sums, pack H_ll / g_l of the shared slots
void launch_phase0_batched(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, hipStream_t s, bool pack) {
  int L = 0, P = 0, slots = 0, npf = 0;
  long long nlf = 0;
  for (int i = 0; i < n; ++i) {
    L = std::max(L, h[i].L); P = std::max(P, h[i].P); slots = std::max(slots, h[i].n_slots);
    npf = std::max(npf, h[i].n_prior + h[i].n_between + h[i].n_ghost);
    nlf = std::max<long long>(nlf, h[i].n_lf);
  }
  BufPtrs B{};
  for (int i = 0; i < n; ++i) B.p[i] = bufs[i];
  if (P + L > 0) hipLaunchKernelGGL(k_relin_b, dim3(blocks_for(P + L, 256), 1, n), dim3(256), 0, s, d);
  if (nlf > 0) {
    const int nb0 = npf > 0 ? (int)blocks_for(npf, 256) : 0;
    hipLaunchKernelGGL(k_lin_lf_b, dim3(nb0 + blocks_for(nlf, 256) + blocks_for(32LL * nlf, 256), 1, n), dim3(256), 0, s, d, nb0, (int)blocks_for(nlf, 256));
  } else if (npf > 0) {
    hipLaunchKernelGGL(k_lin_pose_factors_b, dim3(blocks_for(npf, 128), 1, n), dim3(128), 0, s, d);
  }
  if (L > 0 && pack) hipLaunchKernelGGL(k_landmark_b<1>, dim3(blocks_for(L, 4), 1, n), dim3(256), 0, s, d);      // (!pack: the exact joint pass sums and finishes in one launch, k_landmark_b<3>)
  if (slots > 0 && pack) hipLaunchKernelGGL(k_shared_pack_b, dim3(blocks_for(54LL * slots, 128), 1, n), dim3(128), 0, s, d, 0, B);
}
// phase 4: t_l = sum E^T delta_p per landmark, packed for the exchange
void launch_phase4_batched(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, hipStream_t s) {
  int L = 0, P = 0, slots = 0;
  for (int i = 0; i < n; ++i) { L = std::max(L, h[i].L); P = std::max(P, h[i].P); slots = std::max(slots, h[i].n_slots); }
  BufPtrs B{};
  for (int i = 0; i < n; ++i) B.p[i] = bufs[i];
  if (P + L > 0) hipLaunchKernelGGL(k_backsub_b<1>, dim3(std::max(blocks_for(L, 4), blocks_for(6 * P, 256)), 1, n), dim3(256), 0, s, d);
  if (slots > 0) hipLaunchKernelGGL(k_shared_pack_b, dim3(blocks_for(9LL * slots, 128), 1, n), dim3(128), 0, s, d, 1, B);
}
// phase 2: the exchanged t_l back, landmark back-substitution, estimate = theta (+) delta
void launch_phase2_batched(const GraphDev* d, const GraphDev* h, int n, double* const* bufs, hipStream_t s) {
  int L = 0, P = 0, slots = 0;
  for (int i = 0; i < n; ++i) { L = std::max(L, h[i].L); P = std::max(P, h[i].P); slots = std::max(slots, h[i].n_slots); }
  BufPtrs B{};
  for (int i = 0; i < n; ++i) B.p[i] = bufs[i];
  if (slots > 0) hipLaunchKernelGGL(k_shared_unpack_b, dim3(blocks_for(9LL * slots, 128), 1, n), dim3(128), 0, s, d, 1, B);
  if (P + L > 0) {
    hipLaunchKernelGGL(k_backsub_b<2>, dim3(std::max(blocks_for(L, 4), blocks_for(6 * P, 256)), 1, n), dim3(256), 0, s, d);
    hipLaunchKernelGGL(k_estimate_b, dim3(blocks_for(P + L, 256), 1, n), dim3(256), 0, s, d);
  }
}
void launch_shared_pack(const GraphDev& G, int what, double* buf, hipStream_t s) {
  if (G.n_slots > 0) hipLaunchKernelGGL(k_shared_pack, dim3(blocks_for(54LL * G.n_slots, 128)), dim3(128), 0, s, G, what, buf);
}
void launch_shared_unpack(const GraphDev& G, int what, const double* buf, hipStream_t s) {
  if (G.n_slots > 0) hipLaunchKernelGGL(k_shared_unpack, dim3(blocks_for(54LL * G.n_slots, 128)), dim3(128), 0, s, G, what, buf);
}
void launch_ghost_exchange(const GraphDev& G, int what, double* buf, hipStream_t s) {
  if (G.n_gslots > 0) hipLaunchKernelGGL(k_ghost_exchange, dim3(blocks_for(12LL * G.n_gslots, 128)), dim3(128), 0, s, G, what, buf);
}
void launch_estimate(const GraphDev& G, hipStream_t s) {
  if (G.P + G.L == 0) return;
  hipLaunchKernelGGL(k_estimate, dim3(blocks_for(G.P + G.L, 256)), dim3(256), 0, s, G);
}
void launch_estimate_predict(const GraphDev& G, hipStream_t s, int pose, double* out17) {
  if (G.P + G.L == 0) return;
  hipLaunchKernelGGL(k_estimate_predict, dim3(blocks_for(G.P + G.L, 256)), dim3(256), 0, s, G, pose, out17);
}
void launch_final_pack(const GraphDev& G, int pose, double* out16, hipStream_t s) {
  hipLaunchKernelGGL(k_final_pack, dim3(1), dim3(64), 0, s, G, pose, out16);
}

void launch_sep_extract_batched(const GraphDev* d, const GraphDev* h, int n, const int* n_sep_poses, hipStream_t s) {
  int mx = 0;
  for (int i = 0; i < n; ++i) mx = std::max(mx, (h[i].arrow && h[i].pose_sep) ? n_sep_poses[i] : 0);
  if (mx > 0) hipLaunchKernelGGL(k_sep_extract_b, dim3(mx, 3, n), dim3(256), 0, s, d, mx);
}
void launch_sep_pose_scatter_batched(const GraphDev* d, const GraphDev* h, int n, double* const* xloc, hipStream_t s) {
  int P = 0;
  bool any = false;
  for (int i = 0; i < n; ++i) { P = std::max(P, h[i].P); any = any || (h[i].arrow && h[i].pose_sep && h[i].nsep > 0); }
  if (!any || P == 0) return;
  BufPtrs X{};
  for (int i = 0; i < n; ++i) X.p[i] = xloc[i];
  hipLaunchKernelGGL(k_sep_pose_scatter_b, dim3(blocks_for(6LL * P, 256), 1, n), dim3(256), 0, s, d, X);
}
void launch_border_assemble_batched(const GraphDev* d, const GraphDev* h, int n, hipStream_t s) {
  int slots = 0;
  long long work = 0;
  for (int i = 0; i < n; ++i) {
    slots = std::max(slots, h[i].n_slots);
    work = std::max(work, (long long)h[i].nbr * NB * h[i].T * NB);
  }
  if (work <= 0 || slots <= 0) return;
  hipLaunchKernelGGL(k_border_clear_b, dim3((unsigned)std::min<long long>((work + 255) / 256, 4096), 1, n), dim3(256), 0, s, d);
  hipLaunchKernelGGL(k_border_fill_b, dim3((slots + 3) / 4, 1, n), dim3(256), 0, s, d);
  int ngh = 0;
  for (int i = 0; i < n; ++i) ngh = std::max(ngh, h[i].gh_bord ? h[i].n_ghost : 0);
  if (ngh > 0) hipLaunchKernelGGL(k_border_fill_lam_b, dim3((36 * ngh + 255) / 256, 1, n), dim3(256), 0, s, d);
}
void launch_sep_unpack(const SepLayout& Y, hipStream_t s, int c0, int c1, bool to_packed) {
  const int NT = (Y.Ts + Y.nl) * NB;
  if (c1 < 0) c1 = Y.Ts + Y.nl;
  if (NT > 0 && c1 > c0) hipLaunchKernelGGL(k_sep_unpack, dim3((NT + 1 + 255) / 256, (c1 - c0) * NB), dim3(256), 0, s, Y, c0, c1, to_packed ? 1 : 0);
}
// top block of the separator system += the other half's partial (k_sep_gather's split): columns [c0, Ts) of the system and the lambda block
__global__ __launch_bounds__(256) void k_sep_top_add(SepLayout Y, int c0, const double* __restrict__ sys2, int ld2, const double* __restrict__ bord2) {
  const int vr = blockIdx.x * 256 + threadIdx.x, vc = c0 * NB + blockIdx.y;
  const int NL = Y.Ts * NB, NT = (Y.Ts + Y.nl) * NB;
  if (vr > NT || vc >= NT || vr < vc / NB * NB) return;
  double* dst = sep_slot(Y, vr, vc, false);
  if (vc < NL) *dst += sys2[(size_t)(vc - c0 * NB) * ld2 + (vr - c0 * NB)];
  else *dst += bord2[(size_t)(vc - NL) * ((size_t)(Y.nl + 1) * NB) + (vr - NL)];
}
void launch_sep_top_add(const SepLayout& Y, int c0, const double* sys2, int ld2, const double* bord2, hipStream_t s) {
  const int NT = (Y.Ts + Y.nl) * NB;
  const int ncol = NT - c0 * NB;
  if (ncol > 0) hipLaunchKernelGGL(k_sep_top_add, dim3((NT + 1 + 255) / 256, ncol), dim3(256), 0, s, Y, c0, sys2, ld2, bord2);
}
void launch_sep_gather(const GraphDev* h, int n, const int* const* maps, const SepLayout& Y, bool packed, hipStream_t s, const int* tmask,
                       int split_col, unsigned mask_b, double* sys2, int ld2, double* bord2) {
  SepGatherArgs A{};
  A.n = n; A.Y = Y; A.packed = packed ? 1 : 0; A.tmask = tmask;
  A.robot_mask = ~0u; A.split_col = split_col; A.mask_b = mask_b; A.sys2 = sys2; A.ld2 = ld2; A.bord2 = bord2;
  for (int i = 0; i < n; ++i) { A.bord[i] = h[i].bord; A.ldb[i] = h[i].ldb; A.nbr[i] = h[i].nbr; A.map[i] = maps[i]; }
  const int NT = (Y.Ts + Y.nl) * NB;
  if (NT > 0) hipLaunchKernelGGL(k_sep_gather, dim3((NT + 1 + 255) / 256, (NT + SEP_GATHER_COLS - 1) / SEP_GATHER_COLS), dim3(256), 0, s, A);
}
void launch_lam_prepare(const double* bord, int nl, int lam, double* out, hipStream_t s) {
  if (nl > 0) hipLaunchKernelGGL(k_lam_prepare, dim3((nl * NB + 1 + 255) / 256, nl * NB), dim3(256), 0, s, bord, nl, lam, out);
}
void launch_sep_xloc(int n, const int* const* maps, int ms, int lam, const double* xs, const double* xl, double* const* xloc, hipStream_t s) {
  SepScatterArgs A{};
  A.n = n; A.m = ms + lam; A.ms = ms; A.xs = xs; A.xl = xl;
  for (int i = 0; i < n; ++i) { A.xloc[i] = xloc[i]; A.map[i] = maps[i]; }
  if (A.m > 0) hipLaunchKernelGGL(k_sep_xloc, dim3((A.m + 255) / 256, n), dim3(256), 0, s, A);
}
// the last steps of an exact joint pass for all robots of the GPU: pose_delta = dp and the private landmarks' deltas (k_backsub<0>; a
// separator landmark gets 0 there: H_ll^-1 = 0), the separator landmarks' deltas from the separator's solution, estimate
void launch_arrow_finish_batched(const GraphDev* d, const GraphDev* h, int n, const double* xs, const int* sep_off, hipStream_t s) {
  int L = 0, P = 0, slots = 0;
  for (int i = 0; i < n; ++i) { L = std::max(L, h[i].L); P = std::max(P, h[i].P); slots = std::max(slots, h[i].n_slots); }
  if (P + L == 0) return;
  hipLaunchKernelGGL(k_backsub_b<0>, dim3(std::max(blocks_for(L, 4), blocks_for(6 * P, 256)), 1, n), dim3(256), 0, s, d);
  SepScatterArgs A{};
  A.n = n; A.xs = xs;
  if (slots > 0) hipLaunchKernelGGL(k_sep_lm_delta_b, dim3(blocks_for(9LL * slots, 256), 1, n), dim3(256), 0, s, d, A, sep_off);
  hipLaunchKernelGGL(k_estimate_b, dim3(blocks_for(P + L, 256), 1, n), dim3(256), 0, s, d);
}

}  // namespace sl
