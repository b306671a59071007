// SO(3)/SE(3) arithmetic of the product path (host + device).  Conventions follow what the
// reference obtains from GTSAM 4.0.3 Pose3 / Rot3 / Unit3 (call sites: backend/sloam/src/factorgraph/
// graph.cpp:27-30,58-60,163-171; include/factorgraph/cubeFactor.h:54-55,97,133): Pose3 tangent
// order [rot, trans]; default chart = Cayley rotation + additive translation (cubeFactor.h:96-97),
// SLIDE_CHART_EXPMAP = full SE(3) exponential.  Written independently of oracle/ (which checks it).
#pragma once
#include <hip/hip_runtime.h>

#include <math.h>

#define SL_HD __host__ __device__ inline

namespace sl {

struct V3 {
  double x, y, z;
};
struct M3 {
  double a[9];  // row-major
};
struct SE3 {
  M3 R;
  V3 t;
};

SL_HD V3 v3(double x, double y, double z) { return V3{x, y, z}; }
SL_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
SL_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
SL_HD V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
SL_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
SL_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
SL_HD double norm(V3 a) { return sqrt(dot(a, a)); }

SL_HD M3 eye3() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
SL_HD M3 mul(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.a[3 * i + j] = A.a[3 * i] * B.a[j] + A.a[3 * i + 1] * B.a[3 + j] + A.a[3 * i + 2] * B.a[6 + j];
  return C;
}
SL_HD M3 transpose(const M3& A) { return M3{{A.a[0], A.a[3], A.a[6], A.a[1], A.a[4], A.a[7], A.a[2], A.a[5], A.a[8]}}; }
SL_HD V3 mul(const M3& A, V3 v) {
  return V3{A.a[0] * v.x + A.a[1] * v.y + A.a[2] * v.z, A.a[3] * v.x + A.a[4] * v.y + A.a[5] * v.z,
            A.a[6] * v.x + A.a[7] * v.y + A.a[8] * v.z};
}
SL_HD V3 mulT(const M3& A, V3 v) {  // A^T v
  return V3{A.a[0] * v.x + A.a[3] * v.y + A.a[6] * v.z, A.a[1] * v.x + A.a[4] * v.y + A.a[7] * v.z,
            A.a[2] * v.x + A.a[5] * v.y + A.a[8] * v.z};
}
SL_HD M3 hat(V3 w) { return M3{{0, -w.z, w.y, w.z, 0, -w.x, -w.y, w.x, 0}}; }

SL_HD SE3 compose(const SE3& A, const SE3& B) { return SE3{mul(A.R, B.R), mul(A.R, B.t) + A.t}; }
SL_HD SE3 inverse(const SE3& A) {
  M3 Rt = transpose(A.R);
  V3 v = mul(Rt, A.t);
  return SE3{Rt, V3{-v.x, -v.y, -v.z}};
}
SL_HD SE3 between(const SE3& A, const SE3& B) { return compose(inverse(A), B); }
SL_HD V3 transform_from(const SE3& T, V3 p) { return mul(T.R, p) + T.t; }
SL_HD V3 transform_to(const SE3& T, V3 p) { return mulT(T.R, p - T.t); }

// Rodrigues; |w|^2 <= eps uses I + hat(w)
SL_HD M3 so3_exp(V3 w) {
  const double th2 = dot(w, w);
  const M3 W = hat(w);
  M3 R = eye3();
  if (th2 <= 2.220446049250313e-16) {
    for (int i = 0; i < 9; ++i) R.a[i] += W.a[i];
    return R;
  }
  const double th = sqrt(th2);
  const double s = sin(th) / th;
  const double h = sin(0.5 * th);
  const double c = 2.0 * h * h / th2;
  const M3 WW = mul(W, W);
  for (int i = 0; i < 9; ++i) R.a[i] += s * W.a[i] + c * WW.a[i];
  return R;
}
SL_HD V3 so3_log(const M3& R) {
  const double tr = R.a[0] + R.a[4] + R.a[8];
  if (fabs(tr + 1.0) < 1e-10) {
    if (fabs(R.a[8] + 1.0) > 1e-10) {
      const double k = M_PI / sqrt(2.0 + 2.0 * R.a[8]);
      return V3{k * R.a[2], k * R.a[5], k * (1.0 + R.a[8])};
    } else if (fabs(R.a[4] + 1.0) > 1e-10) {
      const double k = M_PI / sqrt(2.0 + 2.0 * R.a[4]);
      return V3{k * R.a[1], k * (1.0 + R.a[4]), k * R.a[7]};
    } else {
      const double k = M_PI / sqrt(2.0 + 2.0 * R.a[0]);
      return V3{k * (1.0 + R.a[0]), k * R.a[3], k * R.a[6]};
    }
  }
  double mag;
  const double tr3 = tr - 3.0;
  if (tr3 < -1e-7) {
    double c = 0.5 * (tr - 1.0);
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double th = acos(c);
    mag = th / (2.0 * sin(th));
  } else {
    mag = 0.5 - tr3 / 12.0;
  }
  return V3{mag * (R.a[7] - R.a[5]), mag * (R.a[2] - R.a[6]), mag * (R.a[3] - R.a[1])};
}
// Cayley chart and its inverse (closed forms)
SL_HD M3 so3_cayley(V3 w) {
  const double x = w.x, y = w.y, z = w.z;
  const double x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z;
  const double f = 1.0 / (4.0 + x2 + y2 + z2), f2 = 2.0 * f;
  return M3{{(4 + x2 - y2 - z2) * f, (xy - 2 * z) * f2, (xz + 2 * y) * f2, (xy + 2 * z) * f2, (4 - x2 + y2 - z2) * f,
             (yz - 2 * x) * f2, (xz - 2 * y) * f2, (yz + 2 * x) * f2, (4 - x2 - y2 + z2) * f}};
}
SL_HD V3 so3_cayley_inv(const M3& A) {
  const double a = A.a[0], b = A.a[1], c = A.a[2], d = A.a[3], e = A.a[4], f = A.a[5], g = A.a[6], h = A.a[7], i = A.a[8];
  const double di = d * i, ce = c * e, cd = c * d, fg = f * g;
  const double M = 1 + e - f * h + i + e * i;
  const double K = -4.0 / (cd * h + M + a * M - g * (c + ce) - b * (d + di - fg));
  return V3{K * (a * f - cd + f), K * (b * f - ce - c), K * (fg - di - d)};
}

SL_HD SE3 se3_exp(const double* xi) {
  const V3 w{xi[0], xi[1], xi[2]}, v{xi[3], xi[4], xi[5]};
  SE3 T;
  T.R = so3_exp(w);
  const double th2 = dot(w, w);
  if (th2 > 2.220446049250313e-16) {
    const V3 par = dot(w, v) * w;
    const V3 wxv = cross(w, v);
    const V3 t = wxv - mul(T.R, wxv) + par;
    T.t = (1.0 / th2) * t;
  } else {
    T.t = v;
  }
  return T;
}
SL_HD void se3_log(const SE3& T, double* xi) {
  const V3 w = so3_log(T.R);
  const double th = norm(w);
  xi[0] = w.x; xi[1] = w.y; xi[2] = w.z;
  if (th < 1e-10) {
    xi[3] = T.t.x; xi[4] = T.t.y; xi[5] = T.t.z;
    return;
  }
  const V3 n = (1.0 / th) * w;
  const V3 WT = cross(n, T.t);
  const V3 WWT = cross(n, WT);
  const double c = 1.0 - th / (2.0 * tan(0.5 * th));
  const V3 u = T.t - (0.5 * th) * WT + c * WWT;
  xi[3] = u.x; xi[4] = u.y; xi[5] = u.z;
}

SL_HD SE3 chart_retract(const double* xi, int chart) {
  if (chart == 1) return se3_exp(xi);
  return SE3{so3_cayley(V3{xi[0], xi[1], xi[2]}), V3{xi[3], xi[4], xi[5]}};
}
SL_HD void chart_local(const SE3& T, double* xi, int chart) {
  if (chart == 1) { se3_log(T, xi); return; }
  const V3 w = so3_cayley_inv(T.R);
  xi[0] = w.x; xi[1] = w.y; xi[2] = w.z; xi[3] = T.t.x; xi[4] = T.t.y; xi[5] = T.t.z;
}
SL_HD SE3 retract(const SE3& X, const double* xi, int chart) { return compose(X, chart_retract(xi, chart)); }
SL_HD void local(const SE3& X, const SE3& Y, double* xi, int chart) { chart_local(between(X, Y), xi, chart); }

// Adjoint in [rot, trans] order, 6x6 row-major: [R 0; hat(t) R, R]
SL_HD void adjoint(const SE3& T, double* Ad) {
  const M3 tR = mul(hat(T.t), T.R);
#pragma unroll
  for (int i = 0; i < 36; ++i) Ad[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Ad[6 * i + j] = T.R.a[3 * i + j];
      Ad[6 * (i + 3) + j] = tR.a[3 * i + j];
      Ad[6 * (i + 3) + j + 3] = T.R.a[3 * i + j];
    }
}

// Tangent basis of the unit sphere at n (3x2, columns b1 b2): axis of the smallest |component|
SL_HD void sphere_basis(V3 n, V3& b1, V3& b2) {
  const double mx = fabs(n.x), my = fabs(n.y), mz = fabs(n.z);
  V3 axis{0, 0, 0};
  if (mx <= my && mx <= mz) axis.x = 1.0;
  else if (my <= mx && my <= mz) axis.y = 1.0;
  else axis.z = 1.0;
  V3 c = cross(n, axis);
  b1 = (1.0 / norm(c)) * c;
  b2 = cross(n, b1);
}
// sphere log map of q at p, expressed in p's basis
SL_HD void sphere_local(V3 p, V3 q, double* out2) {
  const double d = dot(p, q);
  if (d - 1.0 > -1e-16) { out2[0] = 0.0; out2[1] = 0.0; return; }
  if (d + 1.0 < 1e-16) { out2[0] = M_PI; out2[1] = 0.0; return; }
  V3 b1, b2;
  sphere_basis(p, b1, b2);
  const double th = acos(d);
  const double k = th / sin(th);
  const V3 r = k * (q - d * p);
  out2[0] = dot(b1, r);
  out2[1] = dot(b2, r);
}

SL_HD M3 quat_to_R(const double* q) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  const double n = sqrt(x * x + y * y + z * z + w * w);
  x /= n; y /= n; z /= n; w /= n;
  return M3{{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
             2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
}
SL_HD void R_to_quat(const M3& M, double* q) {
  const double* R = M.a;
  const double tr = R[0] + R[4] + R[8];
  double x, y, z, w;
  if (tr > 0) {
    const double s = sqrt(tr + 1.0) * 2;
    w = 0.25 * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
    w = (R[7] - R[5]) / s; x = 0.25 * s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
    w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = 0.25 * s; z = (R[5] + R[7]) / s;
  } else {
    const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
    w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = 0.25 * s;
  }
  if (w < 0) { x = -x; y = -y; z = -z; w = -w; }
  q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}
SL_HD SE3 from7(const double* p) { return SE3{quat_to_R(p + 3), V3{p[0], p[1], p[2]}}; }
SL_HD void to7(const SE3& T, double* p) {
  p[0] = T.t.x; p[1] = T.t.y; p[2] = T.t.z;
  R_to_quat(T.R, p + 3);
}
SL_HD SE3 from12(const double* p) {
  SE3 T;
  for (int i = 0; i < 9; ++i) T.R.a[i] = p[i];
  T.t = V3{p[9], p[10], p[11]};
  return T;
}
SL_HD void to12(const SE3& T, double* p) {
  for (int i = 0; i < 9; ++i) p[i] = T.R.a[i];
  p[9] = T.t.x; p[10] = T.t.y; p[11] = T.t.z;
}

}  // namespace sl
