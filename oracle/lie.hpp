// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
// product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// build, load or call it (as the checker / the timed CPU baseline, never as the product).
//
// CPU restatement of the SO(3)/SE(3) arithmetic the reference obtains from GTSAM 4.0.3
// (un-vendored dependency, README.md:139-146 of the reference; not present in this
// container).  PARITY UNPINNED at the GTSAM boundary: the reference's own live tests pin
// none of these numerics; every statement marked [GTSAM] below
// restates GTSAM 4.0.3's published behaviour (gtsam/geometry/{SO3,Rot3M,Pose3,Unit3}.cpp)
// from the algorithm description, not from source in this container.
//
// Conventions: 3x3 matrices row-major double[9]; Pose = {R, t} is T_world<-sensor
// (reference graph.h:44 "tf_sensor_to_map"); Pose3 tangent order [rot(3), trans(3)]
// (reference graphWrapper.cpp:45-48).
#pragma once
#include <cmath>
#include <cstring>
#include <limits>

namespace orc {

struct Pose {
  double R[9];
  double t[3];
};

enum PoseChart { CHART_CAYLEY = 0, CHART_EXPMAP = 1 };
// Sensitivity experiment only (tools/chart_sensitivity.py builds a second library with -DORC_PERTURB_TRIG): the transcendental functions
// of the Pose3 chart return the next representable number — what another correctly working libm (the device's) may return.  The
// default build calls the standard functions.
#ifdef ORC_PERTURB_TRIG
inline double orc_up(double x) { return std::nextafter(x, x > 0 ? 1e300 : -1e300); }
#define ORC_SIN(x) orc_up(std::sin(x))
#define ORC_ACOS(x) orc_up(std::acos(x))
#define ORC_TAN(x) orc_up(std::tan(x))
#else
#define ORC_SIN(x) std::sin(x)
#define ORC_ACOS(x) std::acos(x)
#define ORC_TAN(x) std::tan(x)
#endif

inline void mat3_identity(double* R) {
  for (int i = 0; i < 9; ++i) R[i] = 0.0;
  R[0] = R[4] = R[8] = 1.0;
}
inline void mat3_mul(const double* A, const double* B, double* C) {
  double out[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += A[3 * i + k] * B[3 * k + j];
      out[3 * i + j] = s;
    }
  std::memcpy(C, out, sizeof(out));
}
inline void mat3_T(const double* A, double* At) {
  double out[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) out[3 * i + j] = A[3 * j + i];
  std::memcpy(At, out, sizeof(out));
}
inline void mat3_vec(const double* A, const double* v, double* out) {
  double o[3];
  for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
  out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
}
inline void mat3T_vec(const double* A, const double* v, double* out) {
  double o[3];
  for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
  out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
}
inline void skew(const double* w, double* W) {
  W[0] = 0;     W[1] = -w[2]; W[2] = w[1];
  W[3] = w[2];  W[4] = 0;     W[5] = -w[0];
  W[6] = -w[1]; W[7] = w[0];  W[8] = 0;
}
inline double norm3(const double* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void cross3(const double* a, const double* b, double* c) {
  double o[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  c[0] = o[0]; c[1] = o[1]; c[2] = o[2];
}

// [GTSAM] SO3::Expmap (Rodrigues): near-zero uses I + W.
inline void so3_expmap(const double* w, double* R) {
  const double theta2 = dot3(w, w);
  double W[9];
  skew(w, W);
  if (theta2 <= std::numeric_limits<double>::epsilon()) {
    mat3_identity(R);
    for (int i = 0; i < 9; ++i) R[i] += W[i];
    return;
  }
  const double theta = std::sqrt(theta2);
  const double s = ORC_SIN(theta);
  const double s2 = ORC_SIN(0.5 * theta);
  const double omc = 2.0 * s2 * s2;
  double K[9], KK[9];
  for (int i = 0; i < 9; ++i) K[i] = W[i] / theta;
  mat3_mul(K, K, KK);
  mat3_identity(R);
  for (int i = 0; i < 9; ++i) R[i] += s * K[i] + omc * KK[i];
}

// [GTSAM] SO3::Logmap with the trace == -1 special cases and the near-identity Taylor branch.
inline void so3_logmap(const double* R, double* w) {
  const double R11 = R[0], R12 = R[1], R13 = R[2];
  const double R21 = R[3], R22 = R[4], R23 = R[5];
  const double R31 = R[6], R32 = R[7], R33 = R[8];
  const double tr = R11 + R22 + R33;
  if (std::fabs(tr + 1.0) < 1e-10) {
    if (std::fabs(R33 + 1.0) > 1e-10) {
      const double k = M_PI / std::sqrt(2.0 + 2.0 * R33);
      w[0] = k * R13; w[1] = k * R23; w[2] = k * (1.0 + R33);
    } else if (std::fabs(R22 + 1.0) > 1e-10) {
      const double k = M_PI / std::sqrt(2.0 + 2.0 * R22);
      w[0] = k * R12; w[1] = k * (1.0 + R22); w[2] = k * R32;
    } else {
      const double k = M_PI / std::sqrt(2.0 + 2.0 * R11);
      w[0] = k * (1.0 + R11); w[1] = k * R21; w[2] = k * R31;
    }
    return;
  }
  double magnitude;
  const double tr_3 = tr - 3.0;
  if (tr_3 < -1e-7) {
    double c = (tr - 1.0) / 2.0;
    if (c > 1.0) c = 1.0;
    if (c < -1.0) c = -1.0;
    const double theta = ORC_ACOS(c);
    magnitude = theta / (2.0 * ORC_SIN(theta));
  } else {
    magnitude = 0.5 - tr_3 / 12.0;
  }
  w[0] = magnitude * (R32 - R23);
  w[1] = magnitude * (R13 - R31);
  w[2] = magnitude * (R21 - R12);
}

// [GTSAM] Rot3::CayleyChart::Retract — closed form of (I - W/2)^-1 (I + W/2).
inline void so3_cayley(const double* w, double* R) {
  const double x = w[0], y = w[1], z = w[2];
  const double x2 = x * x, y2 = y * y, z2 = z * z;
  const double xy = x * y, xz = x * z, yz = y * z;
  const double f = 1.0 / (4.0 + x2 + y2 + z2), _2f = 2.0 * f;
  R[0] = (4 + x2 - y2 - z2) * f; R[1] = (xy - 2 * z) * _2f;     R[2] = (xz + 2 * y) * _2f;
  R[3] = (xy + 2 * z) * _2f;     R[4] = (4 - x2 + y2 - z2) * f; R[5] = (yz - 2 * x) * _2f;
  R[6] = (xz - 2 * y) * _2f;     R[7] = (yz + 2 * x) * _2f;     R[8] = (4 - x2 - y2 + z2) * f;
}

// [GTSAM] Rot3::CayleyChart::Local — inverse of so3_cayley (closed form).
inline void so3_cayley_local(const double* A, double* w) {
  const double a = A[0], b = A[1], c = A[2];
  const double d = A[3], e = A[4], f = A[5];
  const double g = A[6], h = A[7], i = A[8];
  const double di = d * i, ce = c * e, cd = c * d, fg = f * g;
  const double M = 1 + e - f * h + i + e * i;
  const double K = -4.0 / (cd * h + M + a * M - g * (c + ce) - b * (d + di - fg));
  const double x = a * f - cd + f;
  const double y = b * f - ce - c;
  const double z = fg - di - d;
  w[0] = K * x; w[1] = K * y; w[2] = K * z;
}

inline void pose_identity(Pose& T) {
  mat3_identity(T.R);
  T.t[0] = T.t[1] = T.t[2] = 0.0;
}
inline Pose pose_compose(const Pose& A, const Pose& B) {
  Pose C;
  mat3_mul(A.R, B.R, C.R);
  double Rt[3];
  mat3_vec(A.R, B.t, Rt);
  for (int i = 0; i < 3; ++i) C.t[i] = Rt[i] + A.t[i];
  return C;
}
inline Pose pose_inverse(const Pose& A) {
  Pose C;
  mat3_T(A.R, C.R);
  double v[3];
  mat3_vec(C.R, A.t, v);
  for (int i = 0; i < 3; ++i) C.t[i] = -v[i];
  return C;
}
inline Pose pose_between(const Pose& A, const Pose& B) { return pose_compose(pose_inverse(A), B); }
inline void pose_transform_from(const Pose& T, const double* p, double* out) {
  double v[3];
  mat3_vec(T.R, p, v);
  for (int i = 0; i < 3; ++i) out[i] = v[i] + T.t[i];
}
inline void pose_transform_to(const Pose& T, const double* p, double* out) {
  double d[3] = {p[0] - T.t[0], p[1] - T.t[1], p[2] - T.t[2]};
  mat3T_vec(T.R, d, out);
}

// [GTSAM] Pose3::Expmap (full SE(3) exponential), tangent [w, v].
inline Pose pose_expmap(const double* xi) {
  Pose T;
  const double* w = xi;
  const double* v = xi + 3;
  so3_expmap(w, T.R);
  const double theta2 = dot3(w, w);
  if (theta2 > std::numeric_limits<double>::epsilon()) {
    const double wv = dot3(w, v);
    double t_par[3] = {w[0] * wv, w[1] * wv, w[2] * wv};
    double wxv[3];
    cross3(w, v, wxv);
    double Rwxv[3];
    mat3_vec(T.R, wxv, Rwxv);
    for (int i = 0; i < 3; ++i) T.t[i] = (wxv[i] - Rwxv[i] + t_par[i]) / theta2;
  } else {
    for (int i = 0; i < 3; ++i) T.t[i] = v[i];
  }
  return T;
}

// [GTSAM] Pose3::Logmap (Agrawal06iros eq. 14 form).
inline void pose_logmap(const Pose& p, double* xi) {
  double w[3];
  so3_logmap(p.R, w);
  const double t = norm3(w);
  xi[0] = w[0]; xi[1] = w[1]; xi[2] = w[2];
  if (t < 1e-10) {
    xi[3] = p.t[0]; xi[4] = p.t[1]; xi[5] = p.t[2];
    return;
  }
  double wn[3] = {w[0] / t, w[1] / t, w[2] / t};
  double W[9];
  skew(wn, W);
  const double Tan = ORC_TAN(0.5 * t);
  double WT[3], WWT[3];
  mat3_vec(W, p.t, WT);
  mat3_vec(W, WT, WWT);
  const double c = 1.0 - t / (2.0 * Tan);
  for (int i = 0; i < 3; ++i) xi[3 + i] = p.t[i] - (0.5 * t) * WT[i] + c * WWT[i];
}

// [GTSAM] Pose3::ChartAtOrigin::Retract.  Default 4.0.3 build (no GTSAM_POSE3_EXPMAP /
// GTSAM_ROT3_EXPMAP): Pose3(Rot3::Retract(w) = Cayley(w), Point3(v)).  The reference's own
// comment at cubeFactor.h:96-97 names this chart ("Cayley map (default in GTSAM)").
inline Pose pose_chart_retract(const double* xi, int chart) {
  if (chart == CHART_EXPMAP) return pose_expmap(xi);
  Pose T;
  so3_cayley(xi, T.R);
  T.t[0] = xi[3]; T.t[1] = xi[4]; T.t[2] = xi[5];
  return T;
}
inline void pose_chart_local(const Pose& T, double* xi, int chart) {
  if (chart == CHART_EXPMAP) { pose_logmap(T, xi); return; }
  so3_cayley_local(T.R, xi);
  xi[3] = T.t[0]; xi[4] = T.t[1]; xi[5] = T.t[2];
}
// x.retract(v) = x * ChartAtOrigin::Retract(v);  x.localCoordinates(y) = Local(x^-1 y).
inline Pose pose_retract(const Pose& x, const double* xi, int chart) {
  return pose_compose(x, pose_chart_retract(xi, chart));
}
inline void pose_local(const Pose& x, const Pose& y, double* xi, int chart) {
  pose_chart_local(pose_between(x, y), xi, chart);
}

// [GTSAM] Pose3::AdjointMap in [rot, trans] ordering: [R 0; [t]x R  R]  (6x6 row-major).
inline void pose_adjoint(const Pose& T, double* Ad) {
  double tx[9], txR[9];
  skew(T.t, tx);
  mat3_mul(tx, T.R, txR);
  for (int i = 0; i < 36; ++i) Ad[i] = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Ad[6 * i + j] = T.R[3 * i + j];
      Ad[6 * (i + 3) + j] = txR[3 * i + j];
      Ad[6 * (i + 3) + (j + 3)] = T.R[3 * i + j];
    }
}

// pose7 = tx,ty,tz,qx,qy,qz,qw  (geometry_msgs/Pose order; trajectory file order,
// reference sloamNode.cpp:318-337).
inline void quat_to_R(const double* q, double* R) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  x /= n; y /= n; z /= n; w /= n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
inline void R_to_quat(const double* R, double* q) {
  const double tr = R[0] + R[4] + R[8];
  double x, y, z, w;
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    w = 0.25 * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
    w = (R[7] - R[5]) / s; x = 0.25 * s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
    w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = 0.25 * s; z = (R[5] + R[7]) / s;
  } else {
    const double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
    w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = 0.25 * s;
  }
  if (w < 0) { x = -x; y = -y; z = -z; w = -w; }
  q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}
inline Pose pose_from7(const double* p7) {
  Pose T;
  quat_to_R(p7 + 3, T.R);
  T.t[0] = p7[0]; T.t[1] = p7[1]; T.t[2] = p7[2];
  return T;
}
inline void pose_to7(const Pose& T, double* p7) {
  p7[0] = T.t[0]; p7[1] = T.t[1]; p7[2] = T.t[2];
  R_to_quat(T.R, p7 + 3);
}

// [GTSAM] Unit3::basis(): axis = unit vector of the smallest |component| (ties: x, then y),
// b1 = normalize(n x axis), b2 = n x b1.  B is 3x2 row-major.
inline void unit3_basis(const double* n, double* B) {
  const double mx = std::fabs(n[0]), my = std::fabs(n[1]), mz = std::fabs(n[2]);
  double axis[3] = {0, 0, 0};
  if ((mx <= my) && (mx <= mz)) axis[0] = 1.0;
  else if ((my <= mx) && (my <= mz)) axis[1] = 1.0;
  else axis[2] = 1.0;
  double b1[3], b2[3];
  cross3(n, axis, b1);
  const double nb = norm3(b1);
  b1[0] /= nb; b1[1] /= nb; b1[2] /= nb;
  cross3(n, b1, b2);
  for (int i = 0; i < 3; ++i) { B[2 * i] = b1[i]; B[2 * i + 1] = b2[i]; }
}

// [GTSAM] Unit3::localCoordinates(q) evaluated at p: sphere log map in p's basis.
inline void unit3_local(const double* p, const double* q, double* out2) {
  double B[6];
  unit3_basis(p, B);
  const double d = dot3(p, q);
  // GTSAM tests |d -+ 1| < 1e-16 and would take acos of a dot product that rounding pushed an ulp
  // past +-1 (NaN); the restatement folds d > 1 / d < -1 into the two special cases.
  if (d - 1.0 > -1e-16) { out2[0] = out2[1] = 0.0; return; }
  if (d + 1.0 < 1e-16) { out2[0] = M_PI; out2[1] = 0.0; return; }
  const double theta = std::acos(d);
  const double k = theta / std::sin(theta);
  double r[3] = {k * (q[0] - p[0] * d), k * (q[1] - p[1] * d), k * (q[2] - p[2] * d)};
  out2[0] = B[0] * r[0] + B[2] * r[1] + B[4] * r[2];
  out2[1] = B[1] * r[0] + B[3] * r[1] + B[5] * r[2];
}

}  // namespace orc
