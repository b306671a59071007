// ORACLE — TEST INFRASTRUCTURE ONLY (see lie.hpp header).
//
// CPU restatement of SlideMatch, the reference's exhaustive (x, y, yaw) map-to-map association:
//   PlaceRecognition::MatchMaps           backend/sloam/src/core/place_recognition.cpp:98-387
//   PlaceRecognition::findTransformation  :736-945 (inter-robot branch), getCentroid :713-722,
//   getMapBoundaries :724-734, solveLSQ :632-695, getxyzYawfromTF :697-711, findInterLoopClosure :498-538
// Objects are Vector7d rows [label, x, y, z, d1, d2, d3].  No reference test pins these numerics
// (place_recognition_test.cpp only logs), so this part is pinned by construction only.
// Deviations (documented): the wall-clock "anytime" budget (:181-196) is replaced by an explicit
// max_rings cap; the reflection fix in solveLSQ (:680-686 — a second JacobiSVD of an orthogonal
// matrix, whose factors are not unique) is restated as the standard Kabsch sign flip.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "lie.hpp"

namespace orc {

struct OrcPlaceParams {
  double dilation_factor;            // 1.2   place_recognition.cpp:29
  double xy_step;                    // 0.5   :30
  double yaw_half_range;             // rad, 180 deg :33-35
  double yaw_step;                   // rad, 2 deg   :39-41
  double match_threshold;            // 0.5   :42
  double match_threshold_dimension;  // 1.0   :44
  int disable_yaw_search;            // false :36
  int ignore_dimension;              // false :46
  int min_num_inliers;               // 5     :50
  int use_lsq;                       // true  :51
  int min_num_map_objects_to_start;  // 1 (the reference passes `true` as the int default, :52)
  int max_rings;                     // -1 = all rings (replaces compute_budget_sec)
};
using PlaceParams = OrcPlaceParams;
inline void place_default_params(OrcPlaceParams* p) {
  p->dilation_factor = 1.2; p->xy_step = 0.5; p->yaw_half_range = 180. * M_PI / 180.;
  p->yaw_step = 2.0 * M_PI / 180.; p->match_threshold = 0.5; p->match_threshold_dimension = 1.0;
  p->disable_yaw_search = 0; p->ignore_dimension = 0; p->min_num_inliers = 5; p->use_lsq = 1;
  p->min_num_map_objects_to_start = 1; p->max_rings = -1;
}
inline PlaceParams place_params_from(const OrcPlaceParams* p) {
  PlaceParams P;
  if (p) P = *p; else place_default_params(&P);
  return P;
}

struct MatchMapsResult {
  double x = 0, y = 0, yaw = 0;
  int best_inliers = -10000;
  std::vector<int> ref_idx, qry_idx;   // matched pairs of the best transform, in query order
  long long candidates = 0;
};

// The candidate lattice exactly as the reference's nested for-loops produce it (repeated
// floating-point addition, :140-146 and :230-233).
inline void slidematch_yaws(const PlaceParams& P, std::vector<double>& yaws) {
  yaws.clear();
  if (P.disable_yaw_search) { yaws.push_back(0.0); return; }
  for (double y = -P.yaw_half_range; y < P.yaw_half_range; y += P.yaw_step) yaws.push_back(y);
}

// Count inliers of one (x, y, yaw) candidate: place_recognition.cpp:246-357.
inline int slidematch_count(const double* ref7, int nr, const double* qry7, int nq, double x, double y, double yaw,
                            const PlaceParams& P, std::vector<int>* ref_idx, std::vector<int>* qry_idx) {
  const double c = std::cos(yaw), s = std::sin(yaw);
  int inl = 0;
  for (int j = 0; j < nq; ++j) {
    const double* q = qry7 + 7 * j;
    // cur_R_t * [qx, qy, 1], then divided by the homogeneous coordinate (== 1)
    double tx = c * q[1] + (-s) * q[2] + x * 1.0;
    double ty = s * q[1] + c * q[2] + y * 1.0;
    const double tw = 0.0 * q[1] + 0.0 * q[2] + 1.0 * 1.0;
    tx = tx / tw; ty = ty / tw;
    for (int k = 0; k < nr; ++k) {
      const double* m = ref7 + 7 * k;
      if (m[0] != q[0]) continue;
      const double xd = m[1] - tx, yd = m[2] - ty;
      double avg = 0;
      if (m[5] == 0 && m[6] == 0) {
        avg = std::fabs(m[4] - q[4]);
      } else {
        for (int d = 4; d < 7; ++d) avg += std::fabs(m[d] - q[d]);
        avg /= 3;
      }
      const bool dist_ok = std::sqrt(xd * xd + yd * yd) < P.match_threshold;
      const bool dim_ok = P.ignore_dimension ? true : (avg < P.match_threshold_dimension);
      if (dist_ok && dim_ok) {
        ++inl;
        if (ref_idx) { ref_idx->push_back(k); qry_idx->push_back(j); }
        break;
      }
    }
  }
  return inl;
}

// place_recognition.cpp:98-387 with explicit half ranges.
inline void match_maps_ranges(const double* ref7, int nr, const double* qry7, int nq, double x_half, double y_half,
                              const PlaceParams& P, MatchMapsResult& R) {
  std::vector<double> yaws;
  slidematch_yaws(P, yaws);
  R = MatchMapsResult();
  const double outer = 10 * P.xy_step;
  const int steps = (int)std::ceil(std::min(x_half, y_half) / outer);
  if (steps <= 0) return;
  const double sx = x_half / (double)steps, sy = y_half / (double)steps;
  if (sx < P.xy_step || sy < P.xy_step) return;   // :169-175
  const int nrings = (P.max_rings >= 0) ? std::min(P.max_rings, steps) : steps;
  for (int cur = 0; cur < nrings; ++cur) {
    const double cs = (double)cur;
    const double x_pe = (cs + 1) * sx, x_ns = -(cs + 1) * sx, x_lb = -cs * sx, x_rb = cs * sx;
    const double y_pe = (cs + 1) * sy, y_ns = -(cs + 1) * sy, y_lb = -cs * sy, y_rb = cs * sy;
    for (double x = x_ns; x <= x_pe; x += P.xy_step) {
      for (double y = y_ns; y <= y_pe; y += P.xy_step) {
        if ((x >= x_lb && x <= x_rb) && (y >= y_lb && y <= y_rb)) continue;
        for (double yaw : yaws) {
          ++R.candidates;
          const int inl = slidematch_count(ref7, nr, qry7, nq, x, y, yaw, P, nullptr, nullptr);
          if (inl > R.best_inliers) { R.best_inliers = inl; R.x = x; R.y = y; R.yaw = yaw; }
        }
      }
    }
  }
  if (R.best_inliers > -10000) slidematch_count(ref7, nr, qry7, nq, R.x, R.y, R.yaw, P, &R.ref_idx, &R.qry_idx);
}

inline void match_maps(const double* ref7, int nr, const double* qry7, int nq, const PlaceParams& P, MatchMapsResult& R) {
  // caller passes already-centred maps; half range = dilation * max |coord| (findTransformation :768-787)
  double mx = 0, my = 0;
  for (int i = 0; i < nr; ++i) { mx = std::max(mx, std::fabs(ref7[7 * i + 1])); my = std::max(my, std::fabs(ref7[7 * i + 2])); }
  for (int i = 0; i < nq; ++i) { mx = std::max(mx, std::fabs(qry7[7 * i + 1])); my = std::max(my, std::fabs(qry7[7 * i + 2])); }
  if (!P.disable_yaw_search) { mx = my = std::max(mx, my); }
  match_maps_ranges(ref7, nr, qry7, nq, mx * P.dilation_factor, my * P.dilation_factor, P, R);
}

// 3x3 SVD by one-sided Jacobi (Hestenes): A = U diag(s) V^T, columns sorted by descending s.
inline void svd3(const double* A, double* U, double* S, double* V) {
  double B[9], Vv[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) B[i] = A[i];
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int k = 0; k < 3; ++k) { al += B[3 * k + p] * B[3 * k + p]; be += B[3 * k + q] * B[3 * k + q]; ga += B[3 * k + p] * B[3 * k + q]; }
        off = std::max(off, std::fabs(ga) / std::sqrt(std::max(al * be, 1e-300)));
        if (std::fabs(ga) < 1e-300) continue;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 3; ++k) {
          const double bp = B[3 * k + p], bq = B[3 * k + q];
          B[3 * k + p] = c * bp - s * bq; B[3 * k + q] = s * bp + c * bq;
          const double vp = Vv[3 * k + p], vq = Vv[3 * k + q];
          Vv[3 * k + p] = c * vp - s * vq; Vv[3 * k + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  double sv[3];
  for (int j = 0; j < 3; ++j) sv[j] = std::sqrt(B[j] * B[j] + B[3 + j] * B[3 + j] + B[6 + j] * B[6 + j]);
  int ord[3] = {0, 1, 2};
  std::sort(ord, ord + 3, [&](int a, int b) { return sv[a] > sv[b]; });
  for (int jj = 0; jj < 3; ++jj) {
    const int j = ord[jj];
    S[jj] = sv[j];
    for (int k = 0; k < 3; ++k) {
      V[3 * k + jj] = Vv[3 * k + j];
      U[3 * k + jj] = sv[j] > 1e-300 ? B[3 * k + j] / sv[j] : 0.0;
    }
  }
  // complete U for (near-)zero singular values: u2 = u0 x u1
  if (S[2] <= 1e-12 * std::max(S[0], 1e-300)) {
    if (S[1] <= 1e-12 * std::max(S[0], 1e-300)) {
      // rank <= 1: pick any unit vector orthogonal to u0
      double u0[3] = {U[0], U[3], U[6]};
      double a[3] = {1, 0, 0};
      if (std::fabs(u0[0]) > 0.9) { a[0] = 0; a[1] = 1; }
      double u1[3] = {u0[1] * a[2] - u0[2] * a[1], u0[2] * a[0] - u0[0] * a[2], u0[0] * a[1] - u0[1] * a[0]};
      const double n1 = std::sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
      for (int k = 0; k < 3; ++k) U[3 * k + 1] = u1[k] / n1;
    }
    const double u0[3] = {U[0], U[3], U[6]}, u1[3] = {U[1], U[4], U[7]};
    U[2] = u0[1] * u1[2] - u0[2] * u1[1];
    U[5] = u0[2] * u1[0] - u0[0] * u1[2];
    U[8] = u0[0] * u1[1] - u0[1] * u1[0];
  }
}

// solveLSQ :632-695 (3-D Kabsch), returns 4x4 row-major tf and (x, y, z, yaw).
inline void solve_lsq(const std::vector<double>& map_xyz, const std::vector<double>& det_xyz, double* tf16, double* xyzyaw) {
  const int n = (int)(map_xyz.size() / 3);
  double cs[3] = {0, 0, 0}, ct[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) { cs[k] += det_xyz[3 * i + k]; ct[k] += map_xyz[3 * i + k]; }
  for (int k = 0; k < 3; ++k) { cs[k] /= n; ct[k] /= n; }
  double H[9] = {0};
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) H[3 * a + b] += (det_xyz[3 * i + a] - cs[a]) * (map_xyz[3 * i + b] - ct[b]);
  double U[9], S[3], V[9];
  svd3(H, U, S, V);
  double R[9];
  auto VUt = [&](const double* Vm) {
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += Vm[3 * a + k] * U[3 * b + k];
        R[3 * a + b] = s;
      }
  };
  VUt(V);
  const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
  if (det < 0) {
    double V2[9];
    for (int i = 0; i < 9; ++i) V2[i] = V[i];
    for (int k = 0; k < 3; ++k) V2[3 * k + 2] = -V2[3 * k + 2];
    VUt(V2);
  }
  double t[3];
  for (int a = 0; a < 3; ++a) t[a] = ct[a] - (R[3 * a] * cs[0] + R[3 * a + 1] * cs[1] + R[3 * a + 2] * cs[2]);
  for (int i = 0; i < 16; ++i) tf16[i] = 0;
  for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) tf16[4 * a + b] = R[3 * a + b]; tf16[4 * a + 3] = t[a]; }
  tf16[15] = 1;
  xyzyaw[0] = t[0]; xyzyaw[1] = t[1]; xyzyaw[2] = t[2];
  xyzyaw[3] = std::atan2(R[3], R[0]);
}

// findTransformation :736-945.  inter (inter_loop_closure == true, :745-800): both maps centred on their XY centroids, ranges from
// the extents (inside match_maps).  intra (:801-816): maps as they are, the three intra half ranges.
inline bool find_transformation(const double* ref7_in, int nr, const double* qry7_in, int nq, PlaceParams P, bool inter,
                                double x_half_intra, double y_half_intra, double yaw_half_intra, int* inliers_out, double* xyzyaw) {
  std::vector<double> ref(ref7_in, ref7_in + 7 * (size_t)nr), qry(qry7_in, qry7_in + 7 * (size_t)nq);
  double cr[2] = {0, 0}, cq[2] = {0, 0};
  MatchMapsResult R;
  if (inter) {
    for (int i = 0; i < nr; ++i) { cr[0] += ref[7 * i + 1]; cr[1] += ref[7 * i + 2]; }
    for (int i = 0; i < nq; ++i) { cq[0] += qry[7 * i + 1]; cq[1] += qry[7 * i + 2]; }
    cr[0] /= nr; cr[1] /= nr; cq[0] /= nq; cq[1] /= nq;
    for (int i = 0; i < nr; ++i) { ref[7 * i + 1] -= cr[0]; ref[7 * i + 2] -= cr[1]; }
    for (int i = 0; i < nq; ++i) { qry[7 * i + 1] -= cq[0]; qry[7 * i + 2] -= cq[1]; }
    match_maps(ref.data(), nr, qry.data(), nq, P, R);
  } else {
    P.yaw_half_range = yaw_half_intra;
    match_maps_ranges(ref.data(), nr, qry.data(), nq, x_half_intra, y_half_intra, P, R);
  }
  if (inliers_out) *inliers_out = R.best_inliers;
  if (R.best_inliers < P.min_num_inliers) return false;
  if (!P.use_lsq) {
    // revertCentroidShift :947-967: T(c_ref) * [Rz(yaw) | (x,y,0)] * T(-c_query)   (zero centroids for intra: the raw transform)
    const double c = std::cos(R.yaw), s = std::sin(R.yaw);
    xyzyaw[0] = cr[0] + R.x + (c * (-cq[0]) - s * (-cq[1]));
    xyzyaw[1] = cr[1] + R.y + (s * (-cq[0]) + c * (-cq[1]));
    xyzyaw[2] = 0.0;
    xyzyaw[3] = std::atan2(s, c);
  } else {
    std::vector<double> mp, dt;
    for (size_t i = 0; i < R.ref_idx.size(); ++i) {
      const double* m = &ref[7 * R.ref_idx[i]];
      const double* q = &qry[7 * R.qry_idx[i]];
      mp.push_back(m[1] + cr[0]); mp.push_back(m[2] + cr[1]); mp.push_back(m[3]);
      dt.push_back(q[1] + cq[0]); dt.push_back(q[2] + cq[1]); dt.push_back(q[3]);
    }
    double tf[16];
    solve_lsq(mp, dt, tf, xyzyaw);
  }
  return true;
}

// findInterLoopClosure :498-538.  tf16_out = yaw + xyz transform "from query to reference" composed as :528-535.
inline bool find_inter_loop_closure(const double* ref7_in, int nr, const double* qry7_in, int nq, const PlaceParams& P,
                                    double* tf16_out, int* inliers_out, double* xyzyaw_out) {
  if (inliers_out) *inliers_out = 0;
  if (nr < P.min_num_map_objects_to_start || nq < P.min_num_map_objects_to_start || nr == 0 || nq == 0) return false;
  double xyzyaw[4];
  if (!find_transformation(ref7_in, nr, qry7_in, nq, P, true, 0, 0, 0, inliers_out, xyzyaw)) return false;
  for (int i = 0; i < 16; ++i) tf16_out[i] = 0;
  tf16_out[0] = std::cos(xyzyaw[3]); tf16_out[1] = -std::sin(xyzyaw[3]);
  tf16_out[4] = std::sin(xyzyaw[3]); tf16_out[5] = std::cos(xyzyaw[3]);
  tf16_out[10] = 1; tf16_out[15] = 1;
  tf16_out[3] = xyzyaw[0]; tf16_out[7] = xyzyaw[1]; tf16_out[11] = xyzyaw[2];
  if (xyzyaw_out) for (int i = 0; i < 4; ++i) xyzyaw_out[i] = xyzyaw[i];
  return true;
}

// findIntraLoopClosure :389-496: measurements (local frame of the query pose) -> map frame with the drifted query pose (:418-440),
// findTransformation with inter_loop_closure == false, tfFromQuery2Candidate = candidate^-1 * query * [Rz(yaw) | (x, y, 0)] (:456-492)
inline bool find_intra_loop_closure(const double* meas7, int nm, const double* submap7, int ns, const Pose& query, const Pose& candidate,
                                    const PlaceParams& P, double x_half, double y_half, double yaw_half, double* tf16_out,
                                    int* inliers_out, double* xyzyaw_out) {
  if (inliers_out) *inliers_out = 0;
  if (nm == 0 || ns == 0) return false;
  if (nm < 4) return false;
  std::vector<double> mw(meas7, meas7 + 7 * (size_t)nm);
  for (int i = 0; i < nm; ++i) pose_transform_from(query, meas7 + 7 * (size_t)i + 1, &mw[7 * (size_t)i + 1]);
  double xyzyaw[4];
  if (!find_transformation(submap7, ns, mw.data(), nm, P, false, x_half, y_half, yaw_half, inliers_out, xyzyaw)) return false;
  Pose Lc;
  const double c = std::cos(xyzyaw[3]), s = std::sin(xyzyaw[3]);
  const double Rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) Lc.R[i] = Rz[i];
  Lc.t[0] = xyzyaw[0]; Lc.t[1] = xyzyaw[1]; Lc.t[2] = 0.0;
  const Pose out = pose_compose(pose_compose(pose_inverse(candidate), query), Lc);
  for (int r = 0; r < 3; ++r) {
    for (int k = 0; k < 3; ++k) tf16_out[4 * r + k] = out.R[3 * r + k];
    tf16_out[4 * r + 3] = out.t[r];
    tf16_out[12 + r] = 0.0;
  }
  tf16_out[15] = 1.0;
  if (xyzyaw_out) for (int i = 0; i < 4; ++i) xyzyaw_out[i] = xyzyaw[i];
  return true;
}

}  // namespace orc
