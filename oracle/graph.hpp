// ORACLE — TEST INFRASTRUCTURE ONLY (see lie.hpp header).  PARITY UNPINNED at the GTSAM
// boundary: no live reference test pins ISAM2::update, BetweenFactor, BearingRangeFactor,
// numericalDerivative or the custom factors.
//
// CPU restatement of the reference factor-graph backend:
//   SemanticFactorGraph            backend/sloam/src/factorgraph/graph.cpp:14-371
//   CubeMeasurement / CubeFactor   include/factorgraph/cubeFactor.h:25-172, src/factorgraph/cubeFactor.cpp:17-53
//   CylinderMeasurement / Factor   include/factorgraph/cylinderFactor.h:22-128, src/factorgraph/cylinderFactor.cpp:20-51
// plus the GTSAM 4.0.3 pieces those call: PriorFactor / BetweenFactor<Pose3>,
// BearingRangeFactor<Pose3,Point3>, noiseModel::Diagonal whitening, numericalDerivative21/22
// and the ISAM2::update + calculateEstimate step, each marked [GTSAM].
//
// iSAM2-equivalent update rule [GTSAM, Kaess et al. 2012; params graph.cpp:15-17]:
//   every solve(): (1) new factors/variables join (delta = 0); (2) every variable whose
//   |delta|_inf >= relinearizeThreshold (0.1) gets theta <- theta (+) delta; (3) because
//   cached linearisations are only reused for factors none of whose variables moved, the
//   linear system is J(theta) d = -r(theta) over ALL factors; it is solved exactly (the
//   reference's multifrontal Cholesky is exact too); (3b) [GTSAM] iSAM2's WILDFIRE cut-off on the back-substitution
//   (ISAM2GaussNewtonParams::wildfireThreshold, 1e-3 in GTSAM 4.0.3: a clique is not re-solved when its parents' delta changed by
//   less than the threshold) restated on the 64-column block chain of the reduced pose system — Graph::wildfire_bound below; OFF
//   by default (GraphParams::wildfire_threshold = 0: every fixture and parity test compares exact updates); (4) estimate = theta (+) delta.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include <vector>

#include "lie.hpp"

namespace orc {

enum VarType { V_POSE = 0, V_POINT = 1, V_CUBE = 2, V_CYL = 3 };
static inline int var_dim(int t) { return t == V_POSE ? 6 : t == V_POINT ? 3 : t == V_CUBE ? 9 : 7; }

// value layouts: POSE R(9) t(3); POINT xyz; CUBE R(9) t(3) scale(3); CYL root(3) ray(3) radius
struct Var {
  int type;
  double val[15];
  double delta[9];
};

enum FType { F_PRIOR = 0, F_BETWEEN = 1, F_BR = 2, F_CUBE = 3, F_CYL = 4, F_GHOST = 5 };   // F_GHOST: sharded-mode only
static inline int fac_dim(int t) { return t == F_BR ? 3 : t == F_CUBE ? 9 : t == F_CYL ? 7 : 6; }

// measurement layouts: PRIOR/BETWEEN pose R(9) t(3); BR bearing(3) range(1);
// CUBE R(9) t(3) scale(3) in the sensor frame; CYL root(3) ray(3) radius in the sensor frame
struct Factor {
  int type;
  int v0, v1;
  double z[15];
  double sigma[9];
};

struct GraphParams {
  int pose_chart = CHART_CAYLEY;     // GTSAM 4.0.3 default build; CHART_EXPMAP = GTSAM_POSE3_EXPMAP builds
  double relin_threshold = 0.1;      // graph.cpp:17
  double noise_floor = 0.01;         // graph.h:125
  double prior_sigma[6] = {1e-6, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6};  // graphWrapper.cpp:31
  double odom_sigma[6] = {0.1, 0.1, 0.1, 0.1, 0.1, 0.1};         // :32
  double cube_sigma[9] = {0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1};  // :33
  double relmeas_sigma[6] = {0.1, 0.1, 0.1, 0.1, 0.1, 0.1};      // :34
  double cyl_sigma = 400.0;          // graphWrapper.cpp:60  (100 * ones * 4)
  double bearing_sigma = 1.0;        // graphWrapper.cpp:63-64
#ifndef ORC_NUMDIFF_DELTA
#define ORC_NUMDIFF_DELTA 1e-6       // (sensitivity experiments build a second library with another value: tools/chart_sensitivity.py)
#endif
  double numdiff_delta = ORC_NUMDIFF_DELTA;       // 1e-6: cubeFactor.cpp:43,48 ; cylinderFactor.cpp:41,46
  int num_threads = 1;
  double wildfire_threshold = 0.0;   // [GTSAM] ISAM2GaussNewtonParams::wildfireThreshold (1e-3 in the reference's build); 0 = exact back-substitution
};

inline Pose var_pose(const Var& v) {
  Pose T;
  std::memcpy(T.R, v.val, 9 * sizeof(double));
  std::memcpy(T.t, v.val + 9, 3 * sizeof(double));
  return T;
}
inline void set_var_pose(Var& v, const Pose& T) {
  std::memcpy(v.val, T.R, 9 * sizeof(double));
  std::memcpy(v.val + 9, T.t, 3 * sizeof(double));
}

// value (+) tangent per variable type.
//   Pose3: x * ChartAtOrigin::Retract(v) [GTSAM];  Point3: p + v;
//   CubeMeasurement::retract cubeFactor.h:95-114 (pose.retract(v[0:6]) default chart, scale + v[6:9]);
//   CylinderMeasurement::retract cylinderFactor.h:59-64 — tangent order [ray(3), root(3), radius].
inline void var_retract(const Var& in, const double* d, int chart, Var& out) {
  out.type = in.type;
  switch (in.type) {
    case V_POSE: {
      set_var_pose(out, pose_retract(var_pose(in), d, chart));
      break;
    }
    case V_POINT:
      for (int i = 0; i < 3; ++i) out.val[i] = in.val[i] + d[i];
      break;
    case V_CUBE: {
      set_var_pose(out, pose_retract(var_pose(in), d, chart));
      for (int i = 0; i < 3; ++i) out.val[12 + i] = in.val[12 + i] + d[6 + i];
      break;
    }
    case V_CYL:
      for (int i = 0; i < 3; ++i) out.val[3 + i] = in.val[3 + i] + d[i];   // ray
      for (int i = 0; i < 3; ++i) out.val[i] = in.val[i] + d[3 + i];       // root
      out.val[6] = in.val[6] + d[6];
      break;
  }
}

// ---------------------------------------------------------------------------------------
// Unwhitened factor errors.
// ---------------------------------------------------------------------------------------

// [GTSAM] PriorFactor<Pose3>::evaluateError: -Local(x, prior);  H = I.
inline void err_prior(const Factor& f, const Var& x, int chart, double* e) {
  Pose Z;
  std::memcpy(Z.R, f.z, 72); std::memcpy(Z.t, f.z + 9, 24);
  double l[6];
  pose_local(var_pose(x), Z, l, chart);
  for (int i = 0; i < 6; ++i) e[i] = -l[i];
}

// [GTSAM] BetweenFactor<Pose3>::evaluateError: Local(measured, x1^-1 x2);
// H1 = -Ad((x1^-1 x2)^-1), H2 = I (chart Jacobian dropped; no SLOW_BUT_CORRECT_BETWEENFACTOR).
inline void err_between(const Factor& f, const Var& x1, const Var& x2, int chart, double* e) {
  Pose Z;
  std::memcpy(Z.R, f.z, 72); std::memcpy(Z.t, f.z + 9, 24);
  Pose h = pose_between(var_pose(x1), var_pose(x2));
  pose_local(Z, h, e, chart);
}

// [GTSAM] BearingRangeFactor<Pose3,Point3>: Local(measured, (bearing, range)) =
// [Unit3 local at the MEASURED bearing (2); range - measured (1)].
inline void err_br(const Factor& f, const Var& x, const Var& p, double* e) {
  double q[3];
  pose_transform_to(var_pose(x), p.val, q);
  const double rho = norm3(q);
  double b[3] = {q[0] / rho, q[1] / rho, q[2] / rho};
  unit3_local(f.z, b, e);
  e[2] = rho - f.z[3];
}

// CubeFactor::evaluateError cubeFactor.cpp:35: m_.project(p).localCoordinates(cube_lmrk)
//   = [ Pose3::Logmap(q.pose^-1 * (p * m.pose)) ; m.scale - q.scale ]   (cubeFactor.h:46-87,121-127)
inline void err_cube(const Factor& f, const Var& x, const Var& c, double* e) {
  Pose M;
  std::memcpy(M.R, f.z, 72); std::memcpy(M.t, f.z + 9, 24);
  Pose proj = pose_compose(var_pose(x), M);
  Pose err = pose_compose(pose_inverse(var_pose(c)), proj);
  pose_logmap(err, e);
  for (int i = 0; i < 3; ++i) e[6 + i] = f.z[12 + i] - c.val[12 + i];
}

// CylinderFactor::evaluateError cylinderFactor.cpp:35: m_.project(p).localCoordinates(q)
//   project: root' = p * root, ray' = R * ray (cylinderFactor.h:71-77)
//   localCoordinates(q) = [ q.ray - ray' ; q.root - root' ; radius' - q.radius ] (cylinderFactor.h:45-51;
//   Point3::localCoordinates(q) = q - p [GTSAM])
inline void err_cyl(const Factor& f, const Var& x, const Var& q, double* e) {
  Pose T = var_pose(x);
  double root[3], ray[3];
  pose_transform_from(T, f.z, root);
  mat3_vec(T.R, f.z + 3, ray);
  for (int i = 0; i < 3; ++i) e[i] = q.val[3 + i] - ray[i];
  for (int i = 0; i < 3; ++i) e[3 + i] = q.val[i] - root[i];
  e[6] = f.z[6] - q.val[6];
}

// [GTSAM] numericalDerivative11 with Y = Vector: H.col(j) = ((h(x (+) d e_j) - hx) - (h(x (-) d e_j) - hx)) / (2 d)
template <class ErrFn>
inline void numdiff(ErrFn fn, const Var& x, int m, int chart, double delta, double* H /* m x dim row-major */) {
  const int n = var_dim(x.type);
  double hx[9], e1[9], e2[9], dx[9];
  fn(x, hx);
  const double factor = 1.0 / (2.0 * delta);
  for (int j = 0; j < n; ++j) dx[j] = 0.0;
  for (int j = 0; j < n; ++j) {
    Var xp, xm;
    dx[j] = delta;
    var_retract(x, dx, chart, xp);
    fn(xp, e1);
    dx[j] = -delta;
    var_retract(x, dx, chart, xm);
    fn(xm, e2);
    dx[j] = 0.0;
    for (int i = 0; i < m; ++i) H[i * n + j] = ((e1[i] - hx[i]) - (e2[i] - hx[i])) * factor;
  }
}

struct LinFactor {
  int m, d0, d1;     // rows, dims of var0/var1 (d1 = 0 for prior)
  double r[9];       // whitened residual
  double J0[81];     // m x d0 row-major, whitened
  double J1[81];     // m x d1
};

// ghosts: 12 doubles (R row-major, t) per ghost slot — the current value of a pose owned by another rank (sharded mode)
inline void linearize_factor(const Factor& f, const std::vector<Var>& vars, const GraphParams& P, LinFactor& L,
                             const double* ghosts = nullptr) {
  const int chart = P.pose_chart;
  L.m = fac_dim(f.type);
  const Var& x0 = vars[f.v0];
  L.d0 = var_dim(x0.type);
  L.d1 = 0;
  switch (f.type) {
    case F_PRIOR: {
      err_prior(f, x0, chart, L.r);
      for (int i = 0; i < 36; ++i) L.J0[i] = 0.0;
      for (int i = 0; i < 6; ++i) L.J0[7 * i] = 1.0;
      break;
    }
    case F_BETWEEN: {
      const Var& x1 = vars[f.v1];
      L.d1 = 6;
      err_between(f, x0, x1, chart, L.r);
      Pose hinv = pose_between(var_pose(x1), var_pose(x0));
      double Ad[36];
      pose_adjoint(hinv, Ad);
      for (int i = 0; i < 36; ++i) { L.J0[i] = -Ad[i]; L.J1[i] = 0.0; }
      for (int i = 0; i < 6; ++i) L.J1[7 * i] = 1.0;
      break;
    }
    case F_GHOST: {
      // BetweenFactor<Pose3> (graph.cpp:247-258) whose other pose is a constant of this pass: a unary factor.
      // z[12] != 0: the local pose is the FIRST key (J = -Ad(h^-1)), else the second (J = I).
      Var xo{};
      xo.type = V_POSE;
      std::memcpy(xo.val, ghosts + 12 * (size_t)f.v1, 96);
      const bool first = f.z[12] != 0.0;
      if (first) {
        err_between(f, x0, xo, chart, L.r);
        Pose hinv = pose_between(var_pose(xo), var_pose(x0));
        double Ad[36];
        pose_adjoint(hinv, Ad);
        for (int i = 0; i < 36; ++i) L.J0[i] = -Ad[i];
      } else {
        err_between(f, xo, x0, chart, L.r);
        for (int i = 0; i < 36; ++i) L.J0[i] = 0.0;
        for (int i = 0; i < 6; ++i) L.J0[7 * i] = 1.0;
      }
      break;
    }
    case F_BR: {
      const Var& p = vars[f.v1];
      L.d1 = 3;
      err_br(f, x0, p, L.r);
      // [GTSAM] Pose3::bearing / Pose3::range analytic Jacobians:
      //   q = R^T (p - t);  D_q_pose = [skew(q), -I];  D_q_point = R^T
      //   D_b_q = B(b)^T (I - b b^T)/|q|  (basis at the PREDICTED bearing);  D_rho_q = b^T
      Pose T = var_pose(x0);
      double q[3];
      pose_transform_to(T, p.val, q);
      const double rho = norm3(q);
      double b[3] = {q[0] / rho, q[1] / rho, q[2] / rho};
      double B[6];
      unit3_basis(b, B);
      double Dn[9];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Dn[3 * i + j] = ((i == j ? 1.0 : 0.0) - b[i] * b[j]) / rho;
      double Dm[9];  // 3x3: rows 0-1 = B^T Dn, row 2 = b^T
      for (int r = 0; r < 2; ++r)
        for (int j = 0; j < 3; ++j) Dm[3 * r + j] = B[r] * Dn[j] + B[2 + r] * Dn[3 + j] + B[4 + r] * Dn[6 + j];
      for (int j = 0; j < 3; ++j) Dm[6 + j] = b[j];
      double Sq[9];
      skew(q, Sq);
      double Rt[9];
      mat3_T(T.R, Rt);
      for (int r = 0; r < 3; ++r) {
        for (int j = 0; j < 3; ++j) {
          double s = 0, u = 0;
          for (int k = 0; k < 3; ++k) { s += Dm[3 * r + k] * Sq[3 * k + j]; u += Dm[3 * r + k] * Rt[3 * k + j]; }
          L.J0[6 * r + j] = s;
          L.J0[6 * r + 3 + j] = -Dm[3 * r + j];
          L.J1[3 * r + j] = u;
        }
      }
      break;
    }
    case F_CUBE: {
      const Var& c = vars[f.v1];
      L.d1 = 9;
      err_cube(f, x0, c, L.r);
      numdiff([&](const Var& xx, double* e) { err_cube(f, xx, c, e); }, x0, 9, chart, P.numdiff_delta, L.J0);
      numdiff([&](const Var& cc, double* e) { err_cube(f, x0, cc, e); }, c, 9, chart, P.numdiff_delta, L.J1);
      break;
    }
    case F_CYL: {
      const Var& c = vars[f.v1];
      L.d1 = 7;
      err_cyl(f, x0, c, L.r);
      numdiff([&](const Var& xx, double* e) { err_cyl(f, xx, c, e); }, x0, 7, chart, P.numdiff_delta, L.J0);
      numdiff([&](const Var& cc, double* e) { err_cyl(f, x0, cc, e); }, c, 7, chart, P.numdiff_delta, L.J1);
      break;
    }
  }
  // [GTSAM] noiseModel::Diagonal::WhitenSystem: rows divided by sigma.
  for (int i = 0; i < L.m; ++i) {
    const double inv = 1.0 / f.sigma[i];
    L.r[i] *= inv;
    for (int j = 0; j < L.d0; ++j) L.J0[i * L.d0 + j] *= inv;
    for (int j = 0; j < L.d1; ++j) L.J1[i * L.d1 + j] *= inv;
  }
}

// ---------------------------------------------------------------------------------------
// Profile of an assembled symmetric matrix (lower triangle, row-major): fnz[i] = first non-zero column of row i (exact zeros are
// structural: the assembly writes nothing else there), rend[b] = 1 + last row whose first non-zero lies in or left of the 64-column
// block b (monotone).  The factor of a matrix has the same row profile (no fill left of the first non-zero of a row), so the
// blocked Cholesky and the substitutions below skip everything outside it — the same elimination order and arithmetic on the
// entries that are not structurally zero, i.e. the same result as the dense loops.  (A sparse direct solver such as the reference's
// exploits at least this much; without it the timing of this restatement on a pose chain would be a dense n^3 / 3.)
// ---------------------------------------------------------------------------------------
struct CholProfile { std::vector<int> fnz, rend; };
inline CholProfile chol_profile(const double* A, int n, int lda) {
  const int NB = 64;
  CholProfile P;
  P.fnz.resize(n);
  for (int i = 0; i < n; ++i) {
    const double* Ai = A + (size_t)i * lda;
    int f = 0;
    while (f < i && Ai[f] == 0.0) ++f;
    P.fnz[i] = f;
  }
  const int nblk = (n + NB - 1) / NB;
  P.rend.assign(nblk, 0);
  for (int i = 0; i < n; ++i) {
    const int b = P.fnz[i] / NB;
    P.rend[b] = std::max(P.rend[b], i + 1);
  }
  for (int b = 0; b < nblk; ++b) {
    P.rend[b] = std::max(P.rend[b], std::min(n, (b + 1) * NB));
    if (b) P.rend[b] = std::max(P.rend[b], P.rend[b - 1]);
  }
  // rows inside a column block's extent may start further right than the block: the loops below run over full rows of the
  // extent, which only adds exact zeros
  return P;
}
// ---------------------------------------------------------------------------------------
// Blocked Cholesky (lower, row-major, in place) inside the profile (null: dense).  Returns 0 or 1+index of the failing pivot.
// ---------------------------------------------------------------------------------------
inline int chol_lower(double* A, int n, int lda, int nthreads, const CholProfile* prof = nullptr) {
  const int NB = 64;
  (void)nthreads;
  const int n_all = n;
  for (int k0 = 0; k0 < n_all; k0 += NB) {
    const int nb = std::min(NB, n_all - k0);
    const int n = prof ? prof->rend[k0 / NB] : n_all;      // rows below this one are structurally zero in this column block
    // diagonal block
    for (int j = k0; j < k0 + nb; ++j) {
      double* Aj = A + (size_t)j * lda;
      double s = Aj[j];
      for (int k = k0; k < j; ++k) s -= Aj[k] * Aj[k];
      if (!(s > 0.0)) return j + 1;
      const double d = std::sqrt(s);
      Aj[j] = d;
      for (int i = j + 1; i < k0 + nb; ++i) {
        double* Ai = A + (size_t)i * lda;
        double t = Ai[j];
        for (int k = k0; k < j; ++k) t -= Ai[k] * Aj[k];
        Ai[j] = t / d;
      }
    }
    const int r0 = k0 + nb;
    if (r0 >= n) {
      if (r0 >= n_all) break;
      continue;
    }
    // panel: A[i, k0:k0+nb] <- A[i, k0:k0+nb] * L_kk^-T
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int i = r0; i < n; ++i) {
      double* Ai = A + (size_t)i * lda;
      for (int j = k0; j < k0 + nb; ++j) {
        const double* Aj = A + (size_t)j * lda;
        double t = Ai[j];
        for (int k = k0; k < j; ++k) t -= Ai[k] * Aj[k];
        Ai[j] = t / Aj[j];
      }
    }
    // trailing update: A[i, j] -= sum_k A[i,k] A[j,k],  r0 <= j <= i < n
    const int nt = (n - r0 + NB - 1) / NB;
#pragma omp parallel for schedule(dynamic) num_threads(nthreads)
    for (int tile = 0; tile < nt * nt; ++tile) {
      const int ti = tile / nt, tj = tile % nt;
      if (tj > ti) continue;
      const int i0 = r0 + ti * NB, i1 = std::min(n, i0 + NB);
      const int j0 = r0 + tj * NB, j1 = std::min(n, j0 + NB);
      for (int i = i0; i < i1; ++i) {
        double* Ai = A + (size_t)i * lda;
        const double* Pi = Ai + k0;
        const int jend = (ti == tj) ? std::min(j1, i + 1) : j1;
        int j = j0;
        for (; j + 4 <= jend; j += 4) {
          const double* P0 = A + (size_t)j * lda + k0;
          const double* P1 = P0 + lda;
          const double* P2 = P1 + lda;
          const double* P3 = P2 + lda;
          double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma omp simd reduction(+ : s0, s1, s2, s3)
          for (int k = 0; k < nb; ++k) {
            const double a = Pi[k];
            s0 += a * P0[k]; s1 += a * P1[k]; s2 += a * P2[k]; s3 += a * P3[k];
          }
          Ai[j] -= s0; Ai[j + 1] -= s1; Ai[j + 2] -= s2; Ai[j + 3] -= s3;
        }
        for (; j < jend; ++j) {
          const double* P0 = A + (size_t)j * lda + k0;
          double s0 = 0;
#pragma omp simd reduction(+ : s0)
          for (int k = 0; k < nb; ++k) s0 += Pi[k] * P0[k];
          Ai[j] -= s0;
        }
      }
    }
  }
  return 0;
}
inline void chol_solve_lower(const double* L, int n, int lda, double* b, const CholProfile* prof = nullptr) {
  for (int i = 0; i < n; ++i) {
    const double* Li = L + (size_t)i * lda;
    double s = b[i];
    for (int k = prof ? prof->fnz[i] : 0; k < i; ++k) s -= Li[k] * b[k];
    b[i] = s / Li[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i] / L[(size_t)i * lda + i];
    b[i] = s;
    for (int k = prof ? prof->fnz[i] : 0; k < i; ++k) b[k] -= L[(size_t)i * lda + k] * s;
  }
}
// small SPD inverse (d <= 9) via Cholesky; returns false if not SPD
inline bool spd_inverse(const double* H, int d, double* Hinv) {
  double L[81];
  std::memcpy(L, H, sizeof(double) * d * d);
  if (chol_lower(L, d, d, 1) != 0) return false;
  for (int c = 0; c < d; ++c) {
    double e[9];
    for (int i = 0; i < d; ++i) e[i] = (i == c) ? 1.0 : 0.0;
    chol_solve_lower(L, d, d, e);
    for (int i = 0; i < d; ++i) Hinv[i * d + c] = e[i];
  }
  return true;
}

// ---------------------------------------------------------------------------------------
// The graph: SemanticFactorGraph API (graph.h:70-121) over the iSAM2-equivalent solver.
// ---------------------------------------------------------------------------------------
struct SolveStats {
  int n_pose = 0, n_lm = 0, n_factors = 0, n_relin = 0;
  double t_linearize = 0, t_schur = 0, t_chol = 0, t_total = 0;
  int chol_fail = 0;
};

class Graph {
 public:
  GraphParams P;
  std::vector<Var> vars;          // theta (linearisation points) + delta, merged ("in isam")
  std::vector<Var> estimate;      // currEstimate (graph.cpp:267)
  std::vector<Factor> factors;    // merged
  std::vector<Var> pend_vars;     // fvalues (graph.h:151)
  std::vector<uint64_t> pend_keys;
  std::vector<Factor> pend_factors;  // fgraph, v0/v1 hold KEYS until merge
  std::vector<uint64_t> pend_fk0, pend_fk1;
  std::unordered_map<uint64_t, int> key2var;   // merged keys
  std::vector<uint64_t> var_keys;
  SolveStats stats;

  // key = (char << 56) | idx, mirroring gtsam::Symbol; robot chars graph.cpp:325-371
  static uint64_t pose_key(int robot, uint64_t idx) {
    static const char cs[13] = {'x', 'y', 'z', 'm', 'n', 'o', 'p', 'q', 'r', 's', 't', 'v', 'w'};
    const char c = (robot >= 0 && robot < 13) ? cs[robot] : 0;
    return ((uint64_t)(unsigned char)c << 56) | idx;
  }
  static uint64_t lm_key(char c, uint64_t idx) { return ((uint64_t)(unsigned char)c << 56) | idx; }

  bool value_exists(uint64_t k) const { return key2var.count(k) != 0; }

  void insert_value(uint64_t key, const Var& v) {
    pend_keys.push_back(key);
    pend_vars.push_back(v);
  }
  void add_factor(const Factor& f, uint64_t k0, uint64_t k1) {
    pend_factors.push_back(f);
    pend_fk0.push_back(k0);
    pend_fk1.push_back(k1);
  }

  // graph.cpp:24-42
  void setPriors(const Pose& prior, int robot) {
    Factor f{};
    f.type = F_PRIOR;
    std::memcpy(f.z, prior.R, 72); std::memcpy(f.z + 9, prior.t, 24);
    for (int i = 0; i < 6; ++i) f.sigma[i] = P.prior_sigma[i];
    add_factor(f, pose_key(robot, 0), 0);
    Var v{}; v.type = V_POSE; set_var_pose(v, prior);
    insert_value(pose_key(robot, 0), v);
  }
  void add_between_raw(uint64_t k0, uint64_t k1, const Pose& rel, const double* sigma6) {
    Factor f{};
    f.type = F_BETWEEN;
    std::memcpy(f.z, rel.R, 72); std::memcpy(f.z + 9, rel.t, 24);
    for (int i = 0; i < 6; ++i) f.sigma[i] = sigma6[i];
    add_factor(f, k0, k1);
  }
  // graph.cpp:44-151 (non-loop-closure branch; the loopClosureFound branch adds a second
  // between factor with noise_model_closure and a different initial value, see addLoopClosureFactor)
  void addKeyPoseAndBetween(uint64_t prevIdx, uint64_t curIdx, const Pose& rel, const Pose& est, int robot) {
    const double dist = std::max(norm3(rel.t), P.noise_floor);
    double s[6];
    for (int i = 0; i < 6; ++i) s[i] = P.odom_sigma[i] * dist;
    add_between_raw(pose_key(robot, prevIdx), pose_key(robot, curIdx), rel, s);
    Var v{}; v.type = V_POSE; set_var_pose(v, est);
    insert_value(pose_key(robot, curIdx), v);
  }
  // graph.cpp:153-156
  void addPointLandmarkKey(uint64_t idx, const double* xyz) {
    Var v{}; v.type = V_POINT;
    v.val[0] = xyz[0]; v.val[1] = xyz[1]; v.val[2] = xyz[2];
    insert_value(lm_key('u', idx), v);
  }
  // graph.cpp:158-180; Pose3().bearing(p) = Unit3(p) = p/|p| [GTSAM]
  void addRangeBearingFactor(uint64_t poseIdx, uint64_t lmIdx, const double* bearing, double range, int robot) {
    Factor f{};
    f.type = F_BR;
    const double n = norm3(bearing);
    for (int i = 0; i < 3; ++i) f.z[i] = bearing[i] / n;
    f.z[3] = range;
    for (int i = 0; i < 3; ++i) f.sigma[i] = P.bearing_sigma;
    add_factor(f, pose_key(robot, poseIdx), lm_key('u', lmIdx));
  }
  // graph.cpp:182-196 ; cylinder given in the world frame: root, ray, radius
  void addCylinderFactor(uint64_t poseIdx, uint64_t cylIdx, const Pose& pose, const double* root, const double* ray,
                         double radius, bool exists, int robot) {
    Pose inv = pose_inverse(pose);
    Factor f{};
    f.type = F_CYL;
    pose_transform_from(inv, root, f.z);
    mat3_vec(inv.R, ray, f.z + 3);
    f.z[6] = radius;
    for (int i = 0; i < 7; ++i) f.sigma[i] = P.cyl_sigma;
    add_factor(f, pose_key(robot, poseIdx), lm_key('l', cylIdx));
    if (!exists) {
      Var v{}; v.type = V_CYL;
      for (int i = 0; i < 3; ++i) { v.val[i] = root[i]; v.val[3 + i] = ray[i]; }
      v.val[6] = radius;
      insert_value(lm_key('l', cylIdx), v);
    }
  }
  // graph.cpp:198-231 ; cube given in the world frame
  void addCubeFactor(uint64_t poseIdx, uint64_t cubeIdx, const Pose& pose, const Pose& cube_world, const double* scale,
                     bool exists, int robot) {
    Pose local = pose_compose(pose_inverse(pose), cube_world);
    const double dist = std::max(norm3(local.t), 0.1);
    Factor f{};
    f.type = F_CUBE;
    std::memcpy(f.z, local.R, 72); std::memcpy(f.z + 9, local.t, 24);
    for (int i = 0; i < 3; ++i) f.z[12 + i] = scale[i];
    for (int i = 0; i < 9; ++i) f.sigma[i] = P.cube_sigma[i] * dist;
    add_factor(f, pose_key(robot, poseIdx), lm_key('c', cubeIdx));
    if (!exists) {
      Var v{}; v.type = V_CUBE;
      set_var_pose(v, cube_world);
      for (int i = 0; i < 3; ++i) v.val[12 + i] = scale[i];
      insert_value(lm_key('c', cubeIdx), v);
    }
  }
  // graph.cpp:233-245 ; noise_model_closure = 0.01 * odom (graphWrapper.cpp:55)
  void addLoopClosureFactor(const Pose& rel, uint64_t prevIdx, int robot1, uint64_t curIdx, int robot2) {
    double s[6];
    for (int i = 0; i < 6; ++i) s[i] = P.odom_sigma[i] * 0.01;
    add_between_raw(pose_key(robot1, prevIdx), pose_key(robot2, curIdx), rel, s);
  }
  // graph.cpp:247-258
  void addRelativeMeasFactor(const Pose& rel, uint64_t prevIdx, int robot1, uint64_t curIdx, int robot2) {
    const double dist = std::max(norm3(rel.t), P.noise_floor);
    double s[6];
    for (int i = 0; i < 6; ++i) s[i] = P.relmeas_sigma[i] * dist;
    add_between_raw(pose_key(robot1, prevIdx), pose_key(robot2, curIdx), rel, s);
  }

  // graph.cpp:290-312: false + identity when the key is absent
  bool getPose(uint64_t idx, int robot, Pose& out) const {
    auto it = key2var.find(pose_key(robot, idx));
    if (it == key2var.end() || (size_t)it->second >= estimate.size()) { pose_identity(out); return false; }
    out = var_pose(estimate[it->second]);
    return true;
  }
  const Var* getLandmark(char c, uint64_t idx) const {
    auto it = key2var.find(lm_key(c, idx));
    if (it == key2var.end() || (size_t)it->second >= estimate.size()) return nullptr;
    return &estimate[it->second];
  }

  // graph.cpp:260-272
  int solve();

  // ---- [GTSAM] bounded back-substitution (iSAM2's wildfire threshold; graph.cpp:15-18, 260-272) ------------------------------------
  // State the rule needs across solves: the pose index of every variable, the first (lowest) pose observing a landmark, the lowest
  // pose touched by the factors merged since the last solve, and the last solve's reduced solution.
  std::vector<int> wf_pose_idx;        // variable -> pose index (order of insertion among the poses) or -1
  std::vector<int> wf_lm_first;        // variable -> lowest pose index observing the landmark (1 << 30: none yet)
  int wf_np = 0;
  int wf_dirty_min_pose = 1 << 30;     // lowest pose whose blocks of the reduced system the merged factors / relinearisations change
  std::vector<double> wf_prev;         // the last solve's dp, padded with zeros to whole 64-blocks
  int wf_Tprev = 0;                    // its block count (0: no previous solution)
  long long wf_kept_total = 0;
  int wf_kept_last = 0, wf_last_cd = -1;
  void wildfire_bound(std::vector<double>& dp, int n, const struct CholProfile& prof);

  // ---- one-robot-per-rank restatement of the distributed Gauss-Newton pass (checker of the multi-GPU path;
  // the reference has no such mode: every sloam_node holds a full replica) ---------------------
  std::vector<int> sh_var, sh_owner;   // slot -> variable id (or -1), owner flag
  struct DLm { double H[81]; double g[9]; double t[9]; std::vector<int> fac; };
  struct DistState {
    std::vector<LinFactor> lin;
    std::vector<int> pidx, lidx, pose_vars, lm_vars;
    std::vector<DLm> acc;
    std::vector<double> Hinv_all, dp;
    std::vector<std::vector<double>> Eall, Fall;
    // joint solve (phases 31 / 32 / 33): the robot's block before factoring (S0), its factor, the PCG vectors and scalars
    std::vector<double> S0, Lf, g, r, u, w, p, s, x;
    CholProfile prof;
    double gamma_old = 0.0, alpha_old = 0.0, gamma0 = 0.0, gamma_last = 0.0;
    int pcg_done = 0, pcg_its = 0;
  } D;
  // PCG on the global reduced pose system after the factorisation of the robot's own block: the restatement of
  // slide_slam_amd/csrc/pcg_kernels.hip in plain f64 loops (Chronopoulos-Gear recurrences; the preconditioner is the robot's own
  // Cholesky factor; the reference solves the joint graph of all robots directly, graph.cpp:260-272 on a full replica).
  // pcg_iters = 0: every robot's own block solve only.  pcg_tol: the solve counts as converged once gamma = r^T M^-1 r has fallen
  // to pcg_tol^2 times its first value — later iterations are no-ops.
  int pcg_iters = 0;
  double pcg_tol = 0.0;
  // EXACT joint step ("arrow" solve, phases 40 / 41 / 42): the shared landmarks are NOT eliminated into the robots' pose systems —
  // they stay as the separator of the joint graph.  Robot a eliminates its private landmarks and its poses from
  //     [A_a B_a; B_a^T C_a]   (A_a: its reduced pose block, B_a: pose x shared-landmark blocks J_p^T J_l, C_a: its own J_l^T J_l)
  // and contributes C_a - B_a^T A_a^-1 B_a (and the matching right-hand side) to the separator system of all shared landmarks;
  // the sum over the robots is the Schur complement of the JOINT graph onto the shared landmarks, so solving it and substituting
  // back gives exactly the Gauss-Newton step of the full replica the reference solves (graph.cpp:260-272 on a graph holding every
  // robot, sloamNode.cpp:912-1002).  sep_off[slot] = offset of the slot's tangent coordinates in the separator (global, every rank
  // the same; sep_off[nslots] = its dimension m; the slots may be laid out in any order).
  std::vector<int> sep_off;
  struct ArrowState {
    std::vector<double> L, W, yp;     // factor of A_a (row-major lower, n x n), W = L^-1 B_a (n x ma, row-major), y_p = L^-1 b_p
    std::vector<int> lcol, first_row; // local separator column -> global separator coordinate; first non-zero row of the column
    CholProfile prof;
    int n = 0, ma = 0;
  } AR;
  int set_separator(const int* off, int n) { sep_off.assign(off, off + n); return 0; }
  // Inter-robot relative-pose factors inside the exact joint step: the factor between pose a of robot A and pose b of robot B is the
  // rank-6 term U U^T, U = [J_a^T; J_b^T], of the joint normal equations.  It enters as the bordered (quasi-definite) system
  //     [H_rest  U; U^T  -I] [delta; lambda] = [b; -r]        lambda = J_a delta_a + J_b delta_b + r, the factor's linearised residual
  // i.e. six more separator coordinates per factor ("lambda"), coupled to each robot's band by its OWN Jacobian only: neither robot
  // needs the other's pose in its system, and the step is still exactly the joint replica's.  ghost_gid[i] = the factor's index in the
  // job's list of relative-pose measurements (same on every rank) for ghost factor i (in the order of `factors`), lam_total = 6 x the
  // length of that list; the ghost factor with the FIRST key (z[12] != 0) contributes the -I block and the right-hand side -r.
  // Separator coordinates: [landmark slots (sep_off) | lambda].
  std::vector<int> ghost_gid;
  int lam_total = 0;
  int set_ghost_ids(const int* ids, int n, int n_total) { ghost_gid.assign(ids, ids + n); lam_total = 6 * n_total; return 0; }
  void pcg_pack_tl(const std::vector<double>& v, double* buf);
  void pcg_scalars_update(const double* buf);
  // graph.cpp:314-323  isam->marginalCovariance(X(idx)): the (pose, pose) block of the inverse information matrix at the
  // current linearisation = the pose's block of S^-1 (S = landmark-eliminated pose system) = Y^T Y with L Y = E_pose.
  bool keep_factor = false;
  std::vector<double> last_L;
  std::vector<double> S_work;      // the assembled reduced system: kept between solves (no fresh pages every iteration)
  std::vector<int> last_pidx;
  int last_n = 0;
  int pose_covariance(uint64_t key, double* cov36) const {
    auto it = key2var.find(key);
    if (it == key2var.end() || last_n == 0 || (size_t)it->second >= last_pidx.size() || last_pidx[it->second] < 0) return 1;
    const int n = last_n, p0 = 6 * last_pidx[it->second];
    std::vector<double> Y((size_t)n * 6, 0.0);              // Y[row * 6 + c]
    for (int c = 0; c < 6; ++c) {
      for (int i = p0; i < n; ++i) {
        double s = (i == p0 + c) ? 1.0 : 0.0;
        for (int k = p0; k < i; ++k) s -= last_L[(size_t)i * n + k] * Y[(size_t)k * 6 + c];
        Y[(size_t)i * 6 + c] = s / last_L[(size_t)i * n + i];
      }
    }
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) {
        double s = 0.0;
        for (int i = p0; i < n; ++i) s += Y[(size_t)i * 6 + a] * Y[(size_t)i * 6 + b];
        cov36[6 * a + b] = s;
      }
    return 0;
  }
  void merge_pending();
  int set_shared(const int* cls, const int64_t* idx, const int* owner, int n);
  int dist_phase(int phase, double* buf);
  // sharded mode: inter-robot relative-pose factors.  Ghost slots are a global enumeration of the poses that such
  // factors touch; own_var[s] = this rank's variable of slot s or -1.  Phase 20 packs the owned poses' estimates
  // (12 doubles per slot, zeros elsewhere) for an all-reduce(sum), phase 21 adopts the summed buffer.
  std::vector<double> ghost_val;
  std::vector<int> ghost_own;
  int set_ghosts(const int* own_robot, const int64_t* own_idx, int n) {
    ghost_own.assign(n, -1);
    ghost_val.assign(12 * (size_t)n, 0.0);
    for (int i = 0; i < n; ++i) { ghost_val[12 * (size_t)i] = ghost_val[12 * (size_t)i + 4] = ghost_val[12 * (size_t)i + 8] = 1.0; }
    for (int i = 0; i < n; ++i) {
      if (own_robot[i] < 0) continue;
      auto it = key2var.find(pose_key(own_robot[i], (uint64_t)own_idx[i]));
      if (it == key2var.end()) return -1;
      ghost_own[i] = it->second;
    }
    return 0;
  }
  // graph.cpp:247-258 with the other pose on another rank
  void addRelativeMeasGhost(const Pose& rel, uint64_t idx, int robot, int slot, bool local_first) {
    Factor f{};
    f.type = F_GHOST;
    std::memcpy(f.z, rel.R, 72); std::memcpy(f.z + 9, rel.t, 24);
    f.z[12] = local_first ? 1.0 : 0.0;
    const double dist = std::max(norm3(rel.t), P.noise_floor);
    for (int i = 0; i < 6; ++i) f.sigma[i] = P.relmeas_sigma[i] * dist;
    add_factor(f, pose_key(robot, idx), (uint64_t)slot);
  }
};

inline double now_sec();

inline void Graph::merge_pending() {
  // (1) merge fvalues / fgraph  [GTSAM ISAM2::update: new variables get delta = 0]
  for (size_t i = 0; i < pend_vars.size(); ++i) {
    if (key2var.count(pend_keys[i])) continue;  // GTSAM would throw ValuesKeyAlreadyExists
    Var v = pend_vars[i];
    for (int k = 0; k < 9; ++k) v.delta[k] = 0.0;
    key2var[pend_keys[i]] = (int)vars.size();
    var_keys.push_back(pend_keys[i]);
    vars.push_back(v);
    wf_pose_idx.push_back(v.type == V_POSE ? wf_np++ : -1);
    wf_lm_first.push_back(1 << 30);
  }
  for (size_t i = 0; i < pend_factors.size(); ++i) {
    Factor f = pend_factors[i];
    auto a = key2var.find(pend_fk0[i]);
    if (a == key2var.end()) continue;
    f.v0 = a->second;
    f.v1 = -1;
    if (f.type == F_GHOST) {
      f.v1 = (int)pend_fk1[i];                // ghost slot, not a variable
    } else if (f.type != F_PRIOR) {
      auto b = key2var.find(pend_fk1[i]);
      if (b == key2var.end()) continue;
      f.v1 = b->second;
    }
    // (wildfire rule) the lowest pose whose blocks of the reduced system this factor changes: its pose(s); for a landmark factor
    // also the first pose that observes the landmark — a new factor changes H_ll, which enters the Schur terms of EVERY observer
    const int p0 = wf_pose_idx[f.v0];
    wf_dirty_min_pose = std::min(wf_dirty_min_pose, p0);
    if (f.type == F_BETWEEN) wf_dirty_min_pose = std::min(wf_dirty_min_pose, wf_pose_idx[f.v1]);
    else if (f.type == F_BR || f.type == F_CUBE || f.type == F_CYL) {
      wf_dirty_min_pose = std::min(wf_dirty_min_pose, wf_lm_first[f.v1]);
      wf_lm_first[f.v1] = std::min(wf_lm_first[f.v1], p0);
    }
    factors.push_back(f);
  }
  pend_vars.clear(); pend_keys.clear(); pend_factors.clear(); pend_fk0.clear(); pend_fk1.clear();
}

inline int Graph::solve() {
  const double t0 = now_sec();
  merge_pending();

  // (2) relinearisation  [GTSAM CheckRelinearizationFull: maxDelta >= threshold]
  stats = SolveStats();
  std::vector<char> wf_moved;      // (wildfire rule) the variables that were relinearised
  for (size_t vi = 0; vi < vars.size(); ++vi) {
    Var& v = vars[vi];
    const int d = var_dim(v.type);
    double mx = 0.0;
    for (int k = 0; k < d; ++k) mx = std::max(mx, std::fabs(v.delta[k]));
    if (mx >= P.relin_threshold) {
      Var nv;
      var_retract(v, v.delta, P.pose_chart, nv);
      std::memcpy(v.val, nv.val, sizeof(v.val));
      ++stats.n_relin;
      if (P.wildfire_threshold > 0.0) {
        if (wf_moved.empty()) wf_moved.assign(vars.size(), 0);
        wf_moved[vi] = 1;
      }
    }
  }
  if (!wf_moved.empty()) {
    // a relinearised landmark changes the blocks of all its observers: from its first one on; a relinearised pose its own, those of
    // its relative-pose partners and of every pose observing one of its landmarks
    for (size_t vi = 0; vi < vars.size(); ++vi)
      if (wf_moved[vi]) wf_dirty_min_pose = std::min(wf_dirty_min_pose, vars[vi].type == V_POSE ? wf_pose_idx[vi] : wf_lm_first[vi]);
    for (const Factor& f : factors) {
      if (f.type == F_BETWEEN) {
        if (wf_moved[f.v0]) wf_dirty_min_pose = std::min(wf_dirty_min_pose, wf_pose_idx[f.v1]);
        if (wf_moved[f.v1]) wf_dirty_min_pose = std::min(wf_dirty_min_pose, wf_pose_idx[f.v0]);
      } else if ((f.type == F_BR || f.type == F_CUBE || f.type == F_CYL) && wf_moved[f.v0]) {
        wf_dirty_min_pose = std::min(wf_dirty_min_pose, wf_lm_first[f.v1]);
      }
    }
  }

  // (3) linearise everything at theta; landmark-eliminating Schur complement; dense Cholesky.
  std::vector<int> pidx(vars.size(), -1), lidx(vars.size(), -1);
  std::vector<int> pose_vars, lm_vars;
  for (size_t i = 0; i < vars.size(); ++i) {
    if (vars[i].type == V_POSE) { pidx[i] = (int)pose_vars.size(); pose_vars.push_back((int)i); }
    else { lidx[i] = (int)lm_vars.size(); lm_vars.push_back((int)i); }
  }
  const int np = (int)pose_vars.size(), nl = (int)lm_vars.size();
  const int n = 6 * np;
  stats.n_pose = np; stats.n_lm = nl; stats.n_factors = (int)factors.size();
  std::vector<LinFactor> lin(factors.size());
  const int nthreads = P.num_threads;
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (size_t i = 0; i < factors.size(); ++i) linearize_factor(factors[i], vars, P, lin[i], ghost_val.data());
  const double t1 = now_sec();

  std::vector<double>& S = S_work;
  S.assign((size_t)n * n, 0.0);
  std::vector<double> g(n, 0.0);
  struct LmAcc { double H[81]; double g[9]; std::vector<int> fac; };
  std::vector<LmAcc> acc(nl);
  for (auto& a : acc) { std::memset(a.H, 0, sizeof(a.H)); std::memset(a.g, 0, sizeof(a.g)); }
  auto add_block = [&](int pi, int pj, const double* Ja, int da, const double* Jb, int db, int m) {
    // S[6pi.., 6pj..] += Ja^T Jb   (only if pi >= pj blockwise; full block on the diagonal)
    (void)da; (void)db;
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b < 6; ++b) {
        double s = 0.0;
        for (int r = 0; r < m; ++r) s += Ja[r * 6 + a] * Jb[r * 6 + b];
        S[(size_t)(6 * pi + a) * n + 6 * pj + b] += s;
      }
  };
  for (size_t i = 0; i < factors.size(); ++i) {
    const Factor& f = factors[i];
    const LinFactor& L = lin[i];
    const int p0 = pidx[f.v0];
    add_block(p0, p0, L.J0, 6, L.J0, 6, L.m);
    for (int a = 0; a < 6; ++a) {
      double s = 0.0;
      for (int r = 0; r < L.m; ++r) s += L.J0[r * 6 + a] * L.r[r];
      g[6 * p0 + a] += s;
    }
    if (f.type == F_BETWEEN) {
      const int p1 = pidx[f.v1];
      add_block(p1, p1, L.J1, 6, L.J1, 6, L.m);
      if (p1 >= p0) add_block(p1, p0, L.J1, 6, L.J0, 6, L.m);
      else add_block(p0, p1, L.J0, 6, L.J1, 6, L.m);
      for (int a = 0; a < 6; ++a) {
        double s = 0.0;
        for (int r = 0; r < L.m; ++r) s += L.J1[r * 6 + a] * L.r[r];
        g[6 * p1 + a] += s;
      }
    } else if (f.type != F_PRIOR && f.type != F_GHOST) {
      LmAcc& A = acc[lidx[f.v1]];
      const int d = L.d1;
      for (int a = 0; a < d; ++a) {
        for (int b = 0; b < d; ++b) {
          double s = 0.0;
          for (int r = 0; r < L.m; ++r) s += L.J1[r * d + a] * L.J1[r * d + b];
          A.H[a * d + b] += s;
        }
        double s = 0.0;
        for (int r = 0; r < L.m; ++r) s += L.J1[r * d + a] * L.r[r];
        A.g[a] += s;
      }
      A.fac.push_back((int)i);
    }
  }
  // Schur complement of every landmark block
  std::vector<double> Hinv_all((size_t)nl * 81, 0.0);
  std::vector<std::vector<double>> Eall(nl);
  for (int l = 0; l < nl; ++l) {
    LmAcc& A = acc[l];
    const int d = var_dim(vars[lm_vars[l]].type);
    double* Hinv = &Hinv_all[(size_t)l * 81];
    if (A.fac.empty()) continue;
    if (!spd_inverse(A.H, d, Hinv)) {
      stats.chol_fail = 1;
      if (getenv("ORC_DEBUG")) {
        fprintf(stderr, "[oracle] landmark %d type %d dim %d nfac %zu H not SPD:\n", l, vars[lm_vars[l]].type, d, A.fac.size());
        for (int a = 0; a < d; ++a) { for (int b = 0; b < d; ++b) fprintf(stderr, " %.6e", A.H[a * d + b]); fprintf(stderr, "\n"); }
      }
      return -2;
    }
    const int nf = (int)A.fac.size();
    std::vector<double>& E = Eall[l];
    E.assign((size_t)nf * 6 * d, 0.0);  // E_a = J0^T J1 (6 x d)
    std::vector<double> Fm((size_t)nf * 6 * d);
    for (int a = 0; a < nf; ++a) {
      const LinFactor& L = lin[A.fac[a]];
      double* Ea = &E[(size_t)a * 6 * d];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < d; ++c) {
          double s = 0.0;
          for (int k = 0; k < L.m; ++k) s += L.J0[k * 6 + r] * L.J1[k * d + c];
          Ea[r * d + c] = s;
        }
      double* Fa = &Fm[(size_t)a * 6 * d];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < d; ++c) {
          double s = 0.0;
          for (int k = 0; k < d; ++k) s += Ea[r * d + k] * Hinv[k * d + c];
          Fa[r * d + c] = s;
        }
    }
    for (int a = 0; a < nf; ++a) {
      const int pa = pidx[factors[A.fac[a]].v0];
      const double* Fa = &Fm[(size_t)a * 6 * d];
      for (int r = 0; r < 6; ++r) {
        double s = 0.0;
        for (int k = 0; k < d; ++k) s += Fa[r * d + k] * A.g[k];
        g[6 * pa + r] -= s;
      }
      for (int b = 0; b < nf; ++b) {
        const int pb = pidx[factors[A.fac[b]].v0];
        if (pb > pa) continue;
        const double* Eb = &E[(size_t)b * 6 * d];
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c) {
            double s = 0.0;
            for (int k = 0; k < d; ++k) s += Fa[r * d + k] * Eb[c * d + k];
            S[(size_t)(6 * pa + r) * n + 6 * pb + c] -= s;
          }
      }
    }
  }
  const double t2 = now_sec();
  const CholProfile prof = chol_profile(S.data(), n, n);
  const int cf = chol_lower(S.data(), n, n, nthreads, &prof);
  if (cf != 0) { stats.chol_fail = cf; return -1; }
  std::vector<double> dp(n);
  for (int i = 0; i < n; ++i) dp[i] = -g[i];
  chol_solve_lower(S.data(), n, n, dp.data(), &prof);
  wildfire_bound(dp, n, prof);
  const double t3 = now_sec();
  if (keep_factor) { last_L = S; last_n = n; last_pidx = pidx; }   // for getPoseCovariance (test sizes only: n^2 doubles)
  for (int p = 0; p < np; ++p)
    for (int k = 0; k < 6; ++k) vars[pose_vars[p]].delta[k] = dp[6 * p + k];
  for (int l = 0; l < nl; ++l) {
    LmAcc& A = acc[l];
    Var& v = vars[lm_vars[l]];
    const int d = var_dim(v.type);
    if (A.fac.empty()) { for (int k = 0; k < d; ++k) v.delta[k] = 0.0; continue; }
    double rhs[9];
    for (int k = 0; k < d; ++k) rhs[k] = A.g[k];
    for (size_t a = 0; a < A.fac.size(); ++a) {
      const int pa = pidx[factors[A.fac[a]].v0];
      const double* Ea = &Eall[l][a * 6 * d];
      for (int k = 0; k < d; ++k) {
        double s = 0.0;
        for (int r = 0; r < 6; ++r) s += Ea[r * d + k] * dp[6 * pa + r];
        rhs[k] += s;
      }
    }
    const double* Hinv = &Hinv_all[(size_t)l * 81];
    for (int k = 0; k < d; ++k) {
      double s = 0.0;
      for (int c = 0; c < d; ++c) s += Hinv[k * d + c] * rhs[c];
      v.delta[k] = -s;
    }
  }
  // (4) calculateEstimate: theta (+) delta
  estimate.resize(vars.size());
  for (size_t i = 0; i < vars.size(); ++i) {
    var_retract(vars[i], vars[i].delta, P.pose_chart, estimate[i]);
    std::memcpy(estimate[i].delta, vars[i].delta, sizeof(vars[i].delta));
  }
  const double t4 = now_sec();
  stats.t_linearize = t1 - t0; stats.t_schur = t2 - t1; stats.t_chol = t3 - t2; stats.t_total = t4 - t0;
  return 0;
}

// [GTSAM] iSAM2's wildfire cut-off (ISAM2::update + calculateEstimate, graph.cpp:260-272; threshold 1e-3 in GTSAM 4.0.3) on the block
// chain of the reduced pose system, as the product states it (slide_graph_set_wildfire; chol_kernels.hip bwd_chain_body<.., true>):
// T = ceil(n / 64) blocks of 64 coordinates; block c's solution depends, through L^T x = y, on the blocks c+1 .. prof[c] (prof: the
// monotone tile profile of the factor).  Below the first DIRTY block column c_d = floor(6 p_min / 64) — p_min the lowest pose whose
// blocks the factors merged since the last solve or this solve's relinearisations change — a block's factor column and
// forward-substituted right-hand side are the last solve's, so its solution moves only through the blocks it depends on: the HIGHEST
// block c < min(T_prev, c_d) all of whose dependencies changed by less than the threshold (infinity norm over the block, against the
// last solve's values; a block without a previous value counts as changed; a block without dependencies is always re-solved) keeps
// the last solve's solution, and so does everything below it (the profile is monotone).  dp comes in as the exact solution.
inline void Graph::wildfire_bound(std::vector<double>& dp, int n, const CholProfile& prof) {
  const int NB = 64;
  const int T = (n + NB - 1) / NB;
  const double thr = P.wildfire_threshold;
  wf_kept_last = 0;
  const int np = n / 6;
  const int pmin = wf_dirty_min_pose;
  const int c_d = pmin >= np ? T : std::min(T, (6 * std::max(pmin, 0)) / NB);
  wf_last_cd = c_d;
  if (thr > 0.0 && wf_Tprev > 0 && c_d > 0) {
    const int lim = std::min(std::min(wf_Tprev, c_d), T);
    auto val = [&](const std::vector<double>& v, int i) { return i < (int)v.size() ? v[i] : 0.0; };
    int stop = -1;
    for (int c = lim - 1; c >= 0 && stop < 0; --c) {
      const int last = (prof.rend[c] - 1) / NB;      // prof[c]: the last block row inside the profile of block column c
      if (last <= c) continue;
      bool quiet = true;
      for (int j = c + 1; j <= last && quiet; ++j) {
        if (j >= wf_Tprev) { quiet = false; break; }
        for (int k = 0; k < NB; ++k)
          if (!(std::fabs(val(dp, j * NB + k) - val(wf_prev, j * NB + k)) < thr)) { quiet = false; break; }
      }
      if (quiet) stop = c;
    }
    if (stop >= 0) {
      for (int i = 0; i < (stop + 1) * NB && i < n; ++i) dp[i] = val(wf_prev, i);
      wf_kept_last = stop + 1;
      wf_kept_total += stop + 1;
    }
  }
  wf_prev.assign((size_t)T * NB, 0.0);
  for (int i = 0; i < n; ++i) wf_prev[i] = dp[i];
  wf_Tprev = T;
  wf_dirty_min_pose = 1 << 30;
}

inline int Graph::set_shared(const int* cls, const int64_t* idx, const int* owner, int n) {
  sh_var.assign(n, -1);
  sh_owner.assign(n, 0);
  for (int i = 0; i < n; ++i) {
    if (cls[i] < 0) continue;
    const char c = cls[i] == 0 ? 'l' : (cls[i] == 1 ? 'c' : 'u');
    auto it = key2var.find(lm_key(c, (uint64_t)idx[i]));
    if (it == key2var.end()) return -1;
    sh_var[i] = it->second;
    sh_owner[i] = owner[i] ? 1 : 0;
  }
  return 0;
}

// t_l(v) = sum_f E_f^T v_pose(f) of every landmark (kept in acc[l].t), the shared slots' packed 9 per slot into buf (zeros for the
// slots this robot does not observe)
inline void Graph::pcg_pack_tl(const std::vector<double>& v, double* buf) {
  const int nl = (int)D.lm_vars.size(), nslots = (int)sh_var.size();
  for (int l = 0; l < nl; ++l) {
    DLm& A = D.acc[l];
    const int d = var_dim(vars[D.lm_vars[l]].type);
    for (int k = 0; k < 9; ++k) A.t[k] = 0.0;
    for (size_t a = 0; a < A.fac.size(); ++a) {
      const int pa = D.pidx[factors[A.fac[a]].v0];
      const double* Ea = &D.Eall[l][a * 6 * d];
      for (int k = 0; k < d; ++k) {
        double s2 = 0.0;
        for (int r = 0; r < 6; ++r) s2 += Ea[r * d + k] * v[6 * pa + r];
        A.t[k] += s2;
      }
    }
  }
  for (int sidx = 0; sidx < nslots; ++sidx) {
    double* o = buf + 9 * (size_t)sidx;
    for (int k = 0; k < 9; ++k) o[k] = (sh_var[sidx] >= 0) ? D.acc[D.lidx[sh_var[sidx]]].t[k] : 0.0;
  }
}
// alpha, beta from the all-reduced (gamma, delta) — Chronopoulos-Gear: beta = gamma / gamma_old, alpha = gamma / (delta - beta gamma /
// alpha_old) — then p = u + beta p, s = w + beta s, x += alpha p, r -= alpha s.  Converged (gamma <= tol^2 gamma_0, or gamma == 0):
// alpha = beta = 0, the iteration is a no-op.
inline void Graph::pcg_scalars_update(const double* buf) {
  const double gamma = buf[0], delta = buf[1];
  double beta = 0.0, alpha;
  if (D.gamma_old > 0.0) {
    beta = gamma / D.gamma_old;
    alpha = gamma / (delta - beta * gamma / D.alpha_old);
  } else {
    alpha = gamma / delta;
    D.gamma0 = gamma;
  }
  if (!(gamma > 0.0) || gamma <= pcg_tol * pcg_tol * D.gamma0 || D.pcg_done) { alpha = 0.0; beta = 0.0; D.pcg_done = 1; }
  else if (!(alpha > 0.0) || !(alpha < 1e300)) { alpha = 0.0; beta = 0.0; D.pcg_done = 2; }       // breakdown: not positive definite
  else ++D.pcg_its;
  if (gamma > 0.0) D.gamma_old = gamma;
  if (alpha > 0.0) D.alpha_old = alpha;
  D.gamma_last = gamma;
  const int n = (int)D.u.size();
  for (int i = 0; i < n; ++i) {
    const double p = D.u[i] + beta * D.p[i];
    const double s2 = D.w[i] + beta * D.s[i];
    D.p[i] = p; D.s[i] = s2;
    D.x[i] += alpha * p;
    D.r[i] -= alpha * s2;
  }
}

// phases and buffer layouts identical to slide_graph_dist_phase (include/slide_gpu.h)
inline int Graph::dist_phase(int phase, double* buf) {
  const int nslots = (int)sh_var.size();
  if (phase == 20) {
    merge_pending();
    for (size_t sidx = 0; sidx < ghost_own.size(); ++sidx) {
      double* o = buf + 12 * sidx;
      for (int k = 0; k < 12; ++k) o[k] = 0.0;
      if (ghost_own[sidx] < 0) continue;
      const Var& v = vars[ghost_own[sidx]];
      Var e;
      var_retract(v, v.delta, P.pose_chart, e);        // current estimate theta (+) delta
      std::memcpy(o, e.val, 96);
    }
    return 0;
  }
  if (phase == 21) {
    std::memcpy(ghost_val.data(), buf, sizeof(double) * ghost_val.size());
    return 0;
  }
  if (phase == 0) {
    merge_pending();
    for (auto& v : vars) {   // batch GN: theta <- theta (+) delta for every variable
      Var nv;
      var_retract(v, v.delta, P.pose_chart, nv);
      std::memcpy(v.val, nv.val, sizeof(v.val));
    }
    D = DistState();
    D.pidx.assign(vars.size(), -1);
    D.lidx.assign(vars.size(), -1);
    for (size_t i = 0; i < vars.size(); ++i) {
      if (vars[i].type == V_POSE) { D.pidx[i] = (int)D.pose_vars.size(); D.pose_vars.push_back((int)i); }
      else { D.lidx[i] = (int)D.lm_vars.size(); D.lm_vars.push_back((int)i); }
    }
    D.lin.resize(factors.size());
    for (size_t i = 0; i < factors.size(); ++i) linearize_factor(factors[i], vars, P, D.lin[i], ghost_val.data());
    D.acc.assign(D.lm_vars.size(), DLm());
    for (auto& a : D.acc) { std::memset(a.H, 0, sizeof(a.H)); std::memset(a.g, 0, sizeof(a.g)); std::memset(a.t, 0, sizeof(a.t)); }
    for (size_t i = 0; i < factors.size(); ++i) {
      const Factor& f = factors[i];
      if (f.type == F_PRIOR || f.type == F_BETWEEN || f.type == F_GHOST) continue;
      const LinFactor& L = D.lin[i];
      DLm& A = D.acc[D.lidx[f.v1]];
      const int d = L.d1;
      for (int a = 0; a < d; ++a) {
        for (int b = 0; b < d; ++b) {
          double s2 = 0.0;
          for (int r = 0; r < L.m; ++r) s2 += L.J1[r * d + a] * L.J1[r * d + b];
          A.H[a * d + b] += s2;
        }
        double s2 = 0.0;
        for (int r = 0; r < L.m; ++r) s2 += L.J1[r * d + a] * L.r[r];
        A.g[a] += s2;
      }
      A.fac.push_back((int)i);
    }
    for (int sidx = 0; sidx < nslots; ++sidx) {
      double* o = buf + 54 * (size_t)sidx;
      for (int k = 0; k < 54; ++k) o[k] = 0.0;
      if (sh_var[sidx] < 0) continue;
      const DLm& A = D.acc[D.lidx[sh_var[sidx]]];
      const int d = var_dim(vars[sh_var[sidx]].type);
      for (int a = 0; a < d; ++a) {
        for (int c = 0; c <= a; ++c) o[a * (a + 1) / 2 + c] = A.H[a * d + c];
        o[45 + a] = A.g[a];
      }
    }
    return 0;
  }
  if (phase == 1) {
    for (int sidx = 0; sidx < nslots; ++sidx) {
      if (sh_var[sidx] < 0) continue;
      const double* in = buf + 54 * (size_t)sidx;
      DLm& A = D.acc[D.lidx[sh_var[sidx]]];
      const int d = var_dim(vars[sh_var[sidx]].type);
      for (int a = 0; a < d; ++a) {
        for (int c = 0; c <= a; ++c) { A.H[a * d + c] = in[a * (a + 1) / 2 + c]; A.H[c * d + a] = in[a * (a + 1) / 2 + c]; }
        A.g[a] = in[45 + a];
      }
    }
    const int np = (int)D.pose_vars.size(), nl = (int)D.lm_vars.size(), n = 6 * np;
    std::vector<double> S((size_t)n * n, 0.0), g(n, 0.0);
    auto add_block = [&](int pi, int pj, const double* Ja, const double* Jb, int m) {
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
          double s2 = 0.0;
          for (int r = 0; r < m; ++r) s2 += Ja[r * 6 + a] * Jb[r * 6 + b];
          S[(size_t)(6 * pi + a) * n + 6 * pj + b] += s2;
        }
    };
    for (size_t i = 0; i < factors.size(); ++i) {
      const Factor& f = factors[i];
      const LinFactor& L = D.lin[i];
      const int p0 = D.pidx[f.v0];
      add_block(p0, p0, L.J0, L.J0, L.m);
      for (int a = 0; a < 6; ++a) {
        double s2 = 0.0;
        for (int r = 0; r < L.m; ++r) s2 += L.J0[r * 6 + a] * L.r[r];
        g[6 * p0 + a] += s2;
      }
      if (f.type == F_BETWEEN) {
        const int p1 = D.pidx[f.v1];
        add_block(p1, p1, L.J1, L.J1, L.m);
        if (p1 >= p0) add_block(p1, p0, L.J1, L.J0, L.m); else add_block(p0, p1, L.J0, L.J1, L.m);
        for (int a = 0; a < 6; ++a) {
          double s2 = 0.0;
          for (int r = 0; r < L.m; ++r) s2 += L.J1[r * 6 + a] * L.r[r];
          g[6 * p1 + a] += s2;
        }
      }
    }
    D.Hinv_all.assign((size_t)nl * 81, 0.0);
    D.Eall.assign(nl, {});
    D.Fall.assign(nl, {});
    for (int l = 0; l < nl; ++l) {
      DLm& A = D.acc[l];
      const int d = var_dim(vars[D.lm_vars[l]].type);
      double* Hinv = &D.Hinv_all[(size_t)l * 81];
      if (A.fac.empty()) continue;
      if (!spd_inverse(A.H, d, Hinv)) return -2;
      const int nf = (int)A.fac.size();
      std::vector<double>& E = D.Eall[l];
      E.assign((size_t)nf * 6 * d, 0.0);
      std::vector<double>& Fm = D.Fall[l];
      Fm.assign((size_t)nf * 6 * d, 0.0);
      for (int a = 0; a < nf; ++a) {
        const LinFactor& L = D.lin[A.fac[a]];
        double* Ea = &E[(size_t)a * 6 * d];
        double* Fa = &Fm[(size_t)a * 6 * d];
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < d; ++c) {
            double s2 = 0.0;
            for (int k = 0; k < L.m; ++k) s2 += L.J0[k * 6 + r] * L.J1[k * d + c];
            Ea[r * d + c] = s2;
          }
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < d; ++c) {
            double s2 = 0.0;
            for (int k = 0; k < d; ++k) s2 += Ea[r * d + k] * Hinv[k * d + c];
            Fa[r * d + c] = s2;
          }
      }
      for (int a = 0; a < nf; ++a) {
        const int pa = D.pidx[factors[A.fac[a]].v0];
        const double* Fa = &Fm[(size_t)a * 6 * d];
        for (int r = 0; r < 6; ++r) {
          double s2 = 0.0;
          for (int k = 0; k < d; ++k) s2 += Fa[r * d + k] * A.g[k];
          g[6 * pa + r] -= s2;
        }
        for (int b = 0; b < nf; ++b) {
          const int pb = D.pidx[factors[A.fac[b]].v0];
          if (pb > pa) continue;
          const double* Eb = &E[(size_t)b * 6 * d];
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) {
              double s2 = 0.0;
              for (int k = 0; k < d; ++k) s2 += Fa[r * d + k] * Eb[c * d + k];
              S[(size_t)(6 * pa + r) * n + 6 * pb + c] -= s2;
            }
        }
      }
    }
    const bool joint = pcg_iters > 0 && nslots > 0;
    const CholProfile prof = chol_profile(S.data(), n, n);
    if (joint) D.S0 = S;
    if (chol_lower(S.data(), n, n, P.num_threads, &prof) != 0) return -1;
    D.dp.assign(n, 0.0);
    for (int i = 0; i < n; ++i) D.dp[i] = -g[i];
    chol_solve_lower(S.data(), n, n, D.dp.data(), &prof);
    if (joint) {
      // PCG head (k_pcg_init): r = b, u = M^-1 b (the block solve just done), x = p = s = 0; t_l(u) goes into the exchange
      D.prof = prof;
      D.Lf.swap(S);
      D.r.assign(n, 0.0);
      for (int i = 0; i < n; ++i) D.r[i] = -g[i];
      D.u = D.dp;
      D.x.assign(n, 0.0); D.p.assign(n, 0.0); D.s.assign(n, 0.0); D.w.assign(n, 0.0);
      D.gamma_old = D.alpha_old = D.gamma0 = D.gamma_last = 0.0;
      D.pcg_done = 0; D.pcg_its = 0;
      pcg_pack_tl(D.u, buf);
      return 0;
    }
    for (int p = 0; p < np; ++p)
      for (int k = 0; k < 6; ++k) vars[D.pose_vars[p]].delta[k] = D.dp[6 * p + k];
    for (int l = 0; l < nl; ++l) {
      DLm& A = D.acc[l];
      const int d = var_dim(vars[D.lm_vars[l]].type);
      for (int k = 0; k < 9; ++k) A.t[k] = 0.0;
      for (size_t a = 0; a < A.fac.size(); ++a) {
        const int pa = D.pidx[factors[A.fac[a]].v0];
        const double* Ea = &D.Eall[l][a * 6 * d];
        for (int k = 0; k < d; ++k) {
          double s2 = 0.0;
          for (int r = 0; r < 6; ++r) s2 += Ea[r * d + k] * D.dp[6 * pa + r];
          A.t[k] += s2;
        }
      }
    }
    for (int sidx = 0; sidx < nslots; ++sidx) {
      double* o = buf + 9 * (size_t)sidx;
      for (int k = 0; k < 9; ++k) o[k] = (sh_var[sidx] >= 0) ? D.acc[D.lidx[sh_var[sidx]]].t[k] : 0.0;
    }
    return 0;
  }
  if (phase == 31) {
    // buf: the all-reduced t_l(u) of the shared slots.  w = S0 u - sum over own factors on shared landmarks F_f (t_sum - t_own)
    // (k_pcg_tl_symv + k_pcg_cross), then the partial dot products (r, u), (w, u) -> buf[0 .. 1] (k_pcg_dots)
    if (!(pcg_iters > 0 && nslots > 0) || D.S0.empty()) return -3;
    const int n = (int)D.u.size();
    for (int i = 0; i < n; ++i) D.w[i] = 0.0;
    for (int i = 0; i < n; ++i) {
      const double* Si = &D.S0[(size_t)i * n];
      double s2 = 0.0;
      const double ui = D.u[i];
      for (int k = D.prof.fnz[i]; k < i; ++k) { s2 += Si[k] * D.u[k]; D.w[k] += Si[k] * ui; }
      D.w[i] += s2 + Si[i] * ui;
    }
    for (int sidx = 0; sidx < nslots; ++sidx) {
      if (sh_var[sidx] < 0) continue;
      const int l = D.lidx[sh_var[sidx]];
      DLm& A = D.acc[l];
      const int d = var_dim(vars[sh_var[sidx]].type);
      double c[9];
      for (int k = 0; k < d; ++k) c[k] = buf[9 * (size_t)sidx + k] - A.t[k];
      for (size_t a = 0; a < A.fac.size(); ++a) {
        const int pa = D.pidx[factors[A.fac[a]].v0];
        const double* Fa = &D.Fall[l][a * 6 * d];
        for (int r = 0; r < 6; ++r) {
          double s2 = 0.0;
          for (int k = 0; k < d; ++k) s2 += Fa[r * d + k] * c[k];
          D.w[6 * pa + r] -= s2;
        }
      }
    }
    double gamma = 0.0, delta = 0.0;
    for (int i = 0; i < n; ++i) { gamma += D.r[i] * D.u[i]; delta += D.w[i] * D.u[i]; }
    buf[0] = gamma; buf[1] = delta;
    return 0;
  }
  if (phase == 32 || phase == 33) {
    // buf[0 .. 1]: the all-reduced (gamma, delta).  k_pcg_scalars + k_pcg_update; 32: u = M^-1 r, t_l(u) -> buf; 33 (last): the
    // joint step replaces the block solve, dp = x, t_l(dp) -> buf (what phase 1 ends with when there is no joint solve)
    if (!(pcg_iters > 0 && nslots > 0) || D.S0.empty()) return -3;
    pcg_scalars_update(buf);
    const int n = (int)D.u.size();
    if (phase == 32) {
      D.u = D.r;
      chol_solve_lower(D.Lf.data(), n, n, D.u.data(), &D.prof);
      pcg_pack_tl(D.u, buf);
      return 0;
    }
    D.dp = D.x;
    for (size_t p = 0; p < D.pose_vars.size(); ++p)
      for (int k = 0; k < 6; ++k) vars[D.pose_vars[p]].delta[k] = D.dp[6 * p + k];
    pcg_pack_tl(D.dp, buf);
    return 0;
  }
  if (phase == 40) {
    // after phase 0 (linearisation, per-landmark sums of the own factors).  buf: [m x m row-major | m] = this robot's contribution
    // to the separator system (lower triangle) and its right-hand side, everything else zero; the caller sums over the robots.
    if ((int)sep_off.size() != nslots + 1) return -3;
    const int ms = sep_off[nslots], m = ms + lam_total;
    std::memset(buf, 0, sizeof(double) * ((size_t)m * m + 2 * (size_t)m));
    const int np = (int)D.pose_vars.size(), nl = (int)D.lm_vars.size(), n = 6 * np;
    // ghost factors (in the order of `factors`) -> their lambda coordinates
    std::vector<int> fac_lam(factors.size(), -1);
    {
      size_t q = 0;
      for (size_t i = 0; i < factors.size(); ++i)
        if (factors[i].type == F_GHOST) {
          if (lam_total > 0) {
            if (q >= ghost_gid.size()) return -5;
            fac_lam[i] = ms + 6 * ghost_gid[q];
          }
          ++q;
        }
    }
    std::vector<int> lm_sep(nl, -1);            // landmark -> global separator offset, or -1 (private)
    for (int sidx = 0; sidx < nslots; ++sidx)
      if (sh_var[sidx] >= 0) lm_sep[D.lidx[sh_var[sidx]]] = sep_off[sidx];
    std::vector<double> S((size_t)n * n, 0.0), g(n, 0.0);
    auto add_block = [&](int pi, int pj, const double* Ja, const double* Jb, int mm) {
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
          double s2 = 0.0;
          for (int r = 0; r < mm; ++r) s2 += Ja[r * 6 + a] * Jb[r * 6 + b];
          S[(size_t)(6 * pi + a) * n + 6 * pj + b] += s2;
        }
    };
    for (size_t i = 0; i < factors.size(); ++i) {
      const Factor& f = factors[i];
      const LinFactor& L = D.lin[i];
      const int p0 = D.pidx[f.v0];
      if (fac_lam[i] >= 0) continue;        // a relative-pose factor: enters through its lambda coordinates, not through H_pp / g_p
      add_block(p0, p0, L.J0, L.J0, L.m);
      for (int a = 0; a < 6; ++a) {
        double s2 = 0.0;
        for (int r = 0; r < L.m; ++r) s2 += L.J0[r * 6 + a] * L.r[r];
        g[6 * p0 + a] += s2;
      }
      if (f.type == F_BETWEEN) {
        const int p1 = D.pidx[f.v1];
        add_block(p1, p1, L.J1, L.J1, L.m);
        if (p1 >= p0) add_block(p1, p0, L.J1, L.J0, L.m); else add_block(p0, p1, L.J0, L.J1, L.m);
        for (int a = 0; a < 6; ++a) {
          double s2 = 0.0;
          for (int r = 0; r < L.m; ++r) s2 += L.J1[r * 6 + a] * L.r[r];
          g[6 * p1 + a] += s2;
        }
      }
    }
    // local separator columns in slot order
    AR = ArrowState();
    AR.n = n;
    std::vector<int> lm_lcol(nl, -1);
    for (int sidx = 0; sidx < nslots; ++sidx) {
      if (sh_var[sidx] < 0) continue;
      const int l = D.lidx[sh_var[sidx]], d = var_dim(vars[sh_var[sidx]].type);
      lm_lcol[l] = (int)AR.lcol.size();
      for (int k = 0; k < d; ++k) AR.lcol.push_back(sep_off[sidx] + k);
    }
    std::vector<int> fac_lcol(factors.size(), -1);
    for (size_t i = 0; i < factors.size(); ++i)
      if (fac_lam[i] >= 0) {
        fac_lcol[i] = (int)AR.lcol.size();
        for (int k = 0; k < 6; ++k) AR.lcol.push_back(fac_lam[i] + k);
      }
    const int ma = AR.ma = (int)AR.lcol.size();
    std::vector<double> B((size_t)n * ma, 0.0);       // B[row * ma + col]
    for (size_t i = 0; i < factors.size(); ++i) {
      if (fac_lam[i] < 0) continue;
      // lambda rows: U^T = the factor's whitened Jacobian w.r.t. the own pose; the first-key side owns -I and -r
      const LinFactor& L = D.lin[i];
      const int p0 = D.pidx[factors[i].v0];
      for (int k = 0; k < 6; ++k)
        for (int a = 0; a < 6; ++a) B[(size_t)(6 * p0 + a) * ma + fac_lcol[i] + k] += L.J0[k * 6 + a];
      if (factors[i].z[12] != 0.0)
        for (int k = 0; k < 6; ++k) {
          const int gk = fac_lam[i] + k;
          buf[(size_t)gk * m + gk] -= 1.0;
          buf[(size_t)m * m + gk] -= L.r[k];
        }
    }
    D.Hinv_all.assign((size_t)nl * 81, 0.0);
    D.Eall.assign(nl, {});
    D.Fall.assign(nl, {});
    for (int l = 0; l < nl; ++l) {
      DLm& A = D.acc[l];
      const int d = var_dim(vars[D.lm_vars[l]].type);
      if (A.fac.empty()) continue;
      const int nf = (int)A.fac.size();
      std::vector<double>& E = D.Eall[l];
      E.assign((size_t)nf * 6 * d, 0.0);
      for (int a = 0; a < nf; ++a) {
        const LinFactor& L = D.lin[A.fac[a]];
        double* Ea = &E[(size_t)a * 6 * d];
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < d; ++c) {
            double s2 = 0.0;
            for (int k = 0; k < L.m; ++k) s2 += L.J0[k * 6 + r] * L.J1[k * d + c];
            Ea[r * d + c] = s2;
          }
      }
      if (lm_sep[l] >= 0) {
        // separator landmark: its pose coupling goes into B_a, its own block and gradient into the separator contribution
        const int c0 = lm_lcol[l], o = lm_sep[l];
        for (int a = 0; a < nf; ++a) {
          const int pa = D.pidx[factors[A.fac[a]].v0];
          const double* Ea = &E[(size_t)a * 6 * d];
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < d; ++c) B[(size_t)(6 * pa + r) * ma + c0 + c] += Ea[r * d + c];
        }
        for (int a = 0; a < d; ++a) {
          for (int c = 0; c <= a; ++c) buf[(size_t)(o + a) * m + o + c] += A.H[a * d + c];
          buf[(size_t)m * m + o + a] -= A.g[a];
        }
        continue;
      }
      double* Hinv = &D.Hinv_all[(size_t)l * 81];
      if (!spd_inverse(A.H, d, Hinv)) return -2;
      std::vector<double>& Fm = D.Fall[l];
      Fm.assign((size_t)nf * 6 * d, 0.0);
      for (int a = 0; a < nf; ++a) {
        const double* Ea = &E[(size_t)a * 6 * d];
        double* Fa = &Fm[(size_t)a * 6 * d];
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < d; ++c) {
            double s2 = 0.0;
            for (int k = 0; k < d; ++k) s2 += Ea[r * d + k] * Hinv[k * d + c];
            Fa[r * d + c] = s2;
          }
      }
      for (int a = 0; a < nf; ++a) {
        const int pa = D.pidx[factors[A.fac[a]].v0];
        const double* Fa = &Fm[(size_t)a * 6 * d];
        for (int r = 0; r < 6; ++r) {
          double s2 = 0.0;
          for (int k = 0; k < d; ++k) s2 += Fa[r * d + k] * A.g[k];
          g[6 * pa + r] -= s2;
        }
        for (int b = 0; b < nf; ++b) {
          const int pb = D.pidx[factors[A.fac[b]].v0];
          if (pb > pa) continue;
          const double* Eb = &E[(size_t)b * 6 * d];
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) {
              double s2 = 0.0;
              for (int k = 0; k < d; ++k) s2 += Fa[r * d + k] * Eb[c * d + k];
              S[(size_t)(6 * pa + r) * n + 6 * pb + c] -= s2;
            }
        }
      }
    }
    AR.prof = chol_profile(S.data(), n, n);
    if (chol_lower(S.data(), n, n, P.num_threads, &AR.prof) != 0) return -1;
    AR.L.swap(S);
    const double* Lf = AR.L.data();
    // y_p = L^-1 b_p,  W = L^-1 B_a  (forward substitutions inside the row profile; a column starts at its first non-zero row)
    AR.yp.assign(n, 0.0);
    for (int i = 0; i < n; ++i) {
      const double* Li = Lf + (size_t)i * n;
      double s2 = -g[i];
      for (int k = AR.prof.fnz[i]; k < i; ++k) s2 -= Li[k] * AR.yp[k];
      AR.yp[i] = s2 / Li[i];
    }
    AR.first_row.assign(ma, n);
    for (int i = n - 1; i >= 0; --i)
      for (int c = 0; c < ma; ++c)
        if (B[(size_t)i * ma + c] != 0.0) AR.first_row[c] = i;
    AR.W.swap(B);
    double* W = AR.W.data();
    for (int i = 0; i < n; ++i) {
      const double* Li = Lf + (size_t)i * n;
      double* Wi = W + (size_t)i * ma;
      for (int k = AR.prof.fnz[i]; k < i; ++k) {
        const double lik = Li[k];
        if (lik == 0.0) continue;
        const double* Wk = W + (size_t)k * ma;
        for (int c = 0; c < ma; ++c) Wi[c] -= lik * Wk[c];
      }
      const double inv = 1.0 / Li[i];
      for (int c = 0; c < ma; ++c) Wi[c] *= inv;
    }
    // contribution: C_a - W^T W (lower triangle, global coordinates), right-hand side -g_l - W^T y_p
    const int nthreads = P.num_threads;
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads)
    for (int a = 0; a < ma; ++a) {
      const int ga = AR.lcol[a];
      std::vector<double> col(n);
      for (int i = 0; i < n; ++i) col[i] = W[(size_t)i * ma + a];
      for (int b = 0; b <= a; ++b) {
        const int gb = AR.lcol[b];
        double s2 = 0.0;
        for (int i = std::max(AR.first_row[a], AR.first_row[b]); i < n; ++i) s2 += col[i] * W[(size_t)i * ma + b];
        buf[(size_t)std::max(ga, gb) * m + std::min(ga, gb)] -= s2;         // (lower triangle)
      }
      double s2 = 0.0;
      for (int i = AR.first_row[a]; i < n; ++i) s2 += col[i] * AR.yp[i];
      buf[(size_t)m * m + ga] -= s2;
    }
    return 0;
  }
  if (phase == 41 || phase == 42) {
    // buf: [summed separator system | summed right-hand side | solution].  41: factor and solve it (the solution is left behind the
    // right-hand side for the other robots of the process, which run 42), then back-substitute: poses, private landmarks, retract.
    if ((int)sep_off.size() != nslots + 1) return -3;
    const int ms = sep_off[nslots], m = ms + lam_total, n = AR.n, ma = AR.ma;
    double* xs = buf + (size_t)m * m + m;
    if (phase == 41) {
      // K = [K11 K21^T; K21 K22], K11 (landmarks) positive, K22 (lambda) negative definite: K11 = L11 L11^T, L21 = K21 L11^-T,
      // M = -(K22 - L21 L21^T) = Lm Lm^T positive definite; forward z1 = L11^-1 r1, lambda = -M^-1 (r2 - L21 z1),
      // x1 = L11^-T (z1 - L21^T lambda)
      std::vector<double> Ks(buf, buf + (size_t)m * m);
      for (int i = 0; i < m; ++i) xs[i] = buf[(size_t)m * m + i];
      std::vector<double> K11((size_t)ms * ms);
      for (int i = 0; i < ms; ++i) std::memcpy(&K11[(size_t)i * ms], &Ks[(size_t)i * m], sizeof(double) * ms);
      // a coordinate of the layout that no slot uses (the padding between the blocks of a dissected layout, distributed.py
      // separator_offsets) receives no contribution at all: unit pivot, solution 0
      for (int i = 0; i < ms; ++i)
        if (K11[(size_t)i * ms + i] == 0.0 && xs[i] == 0.0) K11[(size_t)i * ms + i] = 1.0;
      if (ms > 0 && chol_lower(K11.data(), ms, ms, P.num_threads) != 0) return -4;
      for (int i = 0; i < ms; ++i) {          // z1
        double s2 = xs[i];
        for (int k = 0; k < i; ++k) s2 -= K11[(size_t)i * ms + k] * xs[k];
        xs[i] = s2 / K11[(size_t)i * ms + i];
      }
      const int ml = lam_total;
      std::vector<double> L21((size_t)ml * ms), M((size_t)ml * ml, 0.0);
      for (int r = 0; r < ml; ++r) {
        double* Lr = &L21[(size_t)r * ms];
        for (int j = 0; j < ms; ++j) {
          double s2 = Ks[(size_t)(ms + r) * m + j];
          for (int k = 0; k < j; ++k) s2 -= Lr[k] * K11[(size_t)j * ms + k];
          Lr[j] = s2 / K11[(size_t)j * ms + j];
        }
      }
      for (int r = 0; r < ml; ++r) {
        for (int c = 0; c <= r; ++c) {
          double s2 = Ks[(size_t)(ms + r) * m + ms + c];
          for (int k = 0; k < ms; ++k) s2 -= L21[(size_t)r * ms + k] * L21[(size_t)c * ms + k];
          M[(size_t)r * ml + c] = -s2;
        }
        double s2 = xs[ms + r];
        for (int k = 0; k < ms; ++k) s2 -= L21[(size_t)r * ms + k] * xs[k];
        xs[ms + r] = -s2;
      }
      if (ml > 0) {
        if (chol_lower(M.data(), ml, ml, P.num_threads) != 0) return -6;
        chol_solve_lower(M.data(), ml, ml, xs + ms);
      }
      for (int i = 0; i < ms; ++i)
        for (int r = 0; r < ml; ++r) xs[i] -= L21[(size_t)r * ms + i] * xs[ms + r];
      for (int i = ms - 1; i >= 0; --i) {
        const double s2 = xs[i] / K11[(size_t)i * ms + i];
        xs[i] = s2;
        for (int k = 0; k < i; ++k) xs[k] -= K11[(size_t)i * ms + k] * s2;
      }
    }
    // L^T dp = y_p - W x_s
    D.dp.assign(n, 0.0);
    const double* W = AR.W.data();
    for (int i = 0; i < n; ++i) {
      double s2 = AR.yp[i];
      const double* Wi = W + (size_t)i * ma;
      for (int c = 0; c < ma; ++c) s2 -= Wi[c] * xs[AR.lcol[c]];
      D.dp[i] = s2;
    }
    const double* Lf = AR.L.data();
    for (int i = n - 1; i >= 0; --i) {
      const double s2 = D.dp[i] / Lf[(size_t)i * n + i];
      D.dp[i] = s2;
      for (int k = AR.prof.fnz[i]; k < i; ++k) D.dp[k] -= Lf[(size_t)i * n + k] * s2;
    }
    for (size_t p = 0; p < D.pose_vars.size(); ++p)
      for (int k = 0; k < 6; ++k) vars[D.pose_vars[p]].delta[k] = D.dp[6 * p + k];
    std::vector<int> lm_sep(D.lm_vars.size(), -1);
    for (int sidx = 0; sidx < nslots; ++sidx)
      if (sh_var[sidx] >= 0) lm_sep[D.lidx[sh_var[sidx]]] = sep_off[sidx];
    for (size_t l = 0; l < D.lm_vars.size(); ++l) {
      DLm& A = D.acc[l];
      Var& v = vars[D.lm_vars[l]];
      const int d = var_dim(v.type);
      if (lm_sep[l] >= 0) { for (int k = 0; k < d; ++k) v.delta[k] = xs[lm_sep[l] + k]; continue; }
      if (A.fac.empty()) { for (int k = 0; k < d; ++k) v.delta[k] = 0.0; continue; }
      double rhs[9];
      for (int k = 0; k < d; ++k) rhs[k] = A.g[k];
      for (size_t a = 0; a < A.fac.size(); ++a) {
        const int pa = D.pidx[factors[A.fac[a]].v0];
        const double* Ea = &D.Eall[l][a * 6 * d];
        for (int k = 0; k < d; ++k) {
          double s2 = 0.0;
          for (int r = 0; r < 6; ++r) s2 += Ea[r * d + k] * D.dp[6 * pa + r];
          rhs[k] += s2;
        }
      }
      const double* Hinv = &D.Hinv_all[l * 81];
      for (int k = 0; k < d; ++k) {
        double s2 = 0.0;
        for (int c = 0; c < d; ++c) s2 += Hinv[k * d + c] * rhs[c];
        v.delta[k] = -s2;
      }
    }
    estimate.resize(vars.size());
    for (size_t i = 0; i < vars.size(); ++i) {
      var_retract(vars[i], vars[i].delta, P.pose_chart, estimate[i]);
      std::memcpy(estimate[i].delta, vars[i].delta, sizeof(vars[i].delta));
    }
    return 0;
  }
  if (phase == 2) {
    for (int sidx = 0; sidx < nslots; ++sidx) {
      if (sh_var[sidx] < 0) continue;
      DLm& A = D.acc[D.lidx[sh_var[sidx]]];
      for (int k = 0; k < 9; ++k) A.t[k] = buf[9 * (size_t)sidx + k];
    }
    for (size_t l = 0; l < D.lm_vars.size(); ++l) {
      DLm& A = D.acc[l];
      Var& v = vars[D.lm_vars[l]];
      const int d = var_dim(v.type);
      const double* Hinv = &D.Hinv_all[l * 81];
      for (int k = 0; k < d; ++k) {
        double s2 = 0.0;
        for (int c = 0; c < d; ++c) s2 += Hinv[k * d + c] * (A.g[c] + A.t[c]);
        v.delta[k] = A.fac.empty() ? 0.0 : -s2;
      }
    }
    estimate.resize(vars.size());
    for (size_t i = 0; i < vars.size(); ++i) {
      var_retract(vars[i], vars[i].delta, P.pose_chart, estimate[i]);
      std::memcpy(estimate[i].delta, vars[i].delta, sizeof(vars[i].delta));
    }
    return 0;
  }
  if (phase == 10) {
    for (auto& v : vars) {   // commit: theta <- theta (+) delta, delta <- 0
      Var nv;
      var_retract(v, v.delta, P.pose_chart, nv);
      std::memcpy(v.val, nv.val, sizeof(v.val));
      for (int k = 0; k < 9; ++k) v.delta[k] = 0.0;
    }
    for (int sidx = 0; sidx < nslots; ++sidx) {
      double* o = buf + 15 * (size_t)sidx;
      for (int k = 0; k < 15; ++k) o[k] = 0.0;
      if (sh_var[sidx] >= 0 && sh_owner[sidx]) std::memcpy(o, vars[sh_var[sidx]].val, 15 * sizeof(double));
    }
    return 0;
  }
  if (phase == 11) {
    for (int sidx = 0; sidx < nslots; ++sidx) {
      if (sh_var[sidx] < 0) continue;
      Var& v = vars[sh_var[sidx]];
      const int nv = v.type == V_POINT ? 3 : (v.type == V_CUBE ? 15 : 7);
      std::memcpy(v.val, buf + 15 * (size_t)sidx, nv * sizeof(double));
    }
    estimate.resize(vars.size());
    for (size_t i = 0; i < vars.size(); ++i) {
      var_retract(vars[i], vars[i].delta, P.pose_chart, estimate[i]);
      std::memcpy(estimate[i].delta, vars[i].delta, sizeof(vars[i].delta));
    }
    return 0;
  }
  return -3;
}

}  // namespace orc

#include <chrono>
namespace orc {
inline double now_sec() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace orc
