// ORACLE — TEST INFRASTRUCTURE ONLY (see lie.hpp header).
//
// CPU restatement of SlideGraph's triangle matching (sources in the reference tree):
//   compute_triangle_diff   clipper_semantic_object/src/semantic_clipper.cpp:49-108
//   match_triangles         :111-118   (model-major, data-minor double loop; semantic labels are ignored there)
//   estimate_tf             :122-138   (2-D Kabsch, reflection fixed by flipping column 1 of R)
//   run_semantic_clipper    :140-274   (identity association list, CLIPPER, min_num_pairs gate, yaw + xy out)
// The Delaunay triangulation itself (observation.cpp:13-88, qhull) is an INPUT here: triangles come as 3 x (x, y)
// per triangle.  argsort of the three centroid distances: the reference calls std::sort on three
// indices (insertion sort in libstdc++ for short ranges, i.e. stable); restated as a stable three-element sort.
// Parity unpinned by reference fixtures: the reference holds no test for these functions.
#pragma once
#include <cmath>
#include <vector>

#include "clipper.hpp"

namespace orc {

inline void tri_sorted(const double* tri, int* order, double* dist) {
  const double cx = (tri[0] + tri[2] + tri[4]) / 3.0, cy = (tri[1] + tri[3] + tri[5]) / 3.0;
  double d[3];
  for (int i = 0; i < 3; ++i) {
    const double dx = tri[2 * i] - cx, dy = tri[2 * i + 1] - cy;
    d[i] = std::sqrt(dx * dx + dy * dy);
  }
  int o[3] = {0, 1, 2};
  for (int i = 1; i < 3; ++i)                      // stable insertion sort, comparator d[a] < d[b]
    for (int j = i; j > 0 && d[o[j]] < d[o[j - 1]]; --j) { const int t = o[j]; o[j] = o[j - 1]; o[j - 1] = t; }
  for (int i = 0; i < 3; ++i) { order[i] = o[i]; dist[i] = d[o[i]]; }
}

// out: per matched triangle pair three rows [mx, my, dx, dy] in sorted-vertex order; diffs one per pair
inline size_t match_triangles(const double* tm, int ntm, const double* td, int ntd, double thr, std::vector<double>& pts,
                              std::vector<double>& diffs) {
  pts.clear();
  diffs.clear();
  for (int i = 0; i < ntm; ++i) {
    int om[3];
    double dm[3];
    tri_sorted(tm + 6 * (size_t)i, om, dm);
    for (int j = 0; j < ntd; ++j) {
      int od[3];
      double dd[3];
      tri_sorted(td + 6 * (size_t)j, od, dd);
      double s = 0;
      for (int k = 0; k < 3; ++k) s += std::pow(dm[k] - dd[k], 2);
      const double diff = std::sqrt(s);
      if (diff < thr) {
        diffs.push_back(diff);
        for (int k = 0; k < 3; ++k) {
          pts.push_back(tm[6 * (size_t)i + 2 * om[k]]);
          pts.push_back(tm[6 * (size_t)i + 2 * om[k] + 1]);
          pts.push_back(td[6 * (size_t)j + 2 * od[k]]);
          pts.push_back(td[6 * (size_t)j + 2 * od[k] + 1]);
        }
      }
    }
  }
  return diffs.size();
}

// tf3: row-major 3x3, maps a -> b
inline void estimate_tf2d(const double* a, const double* b, int n, double* tf3) {
  double ca[2] = {0, 0}, cb[2] = {0, 0};
  for (int i = 0; i < n; ++i) { ca[0] += a[2 * i]; ca[1] += a[2 * i + 1]; cb[0] += b[2 * i]; cb[1] += b[2 * i + 1]; }
  for (int k = 0; k < 2; ++k) { ca[k] /= n; cb[k] /= n; }
  double H[4] = {0, 0, 0, 0};                    // H = A_c B_c^T
  for (int i = 0; i < n; ++i) {
    const double ax = a[2 * i] - ca[0], ay = a[2 * i + 1] - ca[1], bx = b[2 * i] - cb[0], by = b[2 * i + 1] - cb[1];
    H[0] += ax * bx; H[1] += ax * by; H[2] += ay * bx; H[3] += ay * by;
  }
  // R = V U^T of H = U S V^T is the orthogonal matrix maximising trace(R H).  det H >= 0: it is the rotation by
  // atan2(H01 - H10, H00 + H11).  det H < 0: it is the reflection [[c, s], [s, -c]] with (c, s) at the angle
  // atan2(H01 + H10, H00 - H11), and the reference then negates column 1 of R (semantic_clipper.cpp:130-132), which
  // turns exactly that reflection into the rotation by the same angle.
  const double detH = H[0] * H[3] - H[1] * H[2];
  const double th = detH >= 0.0 ? std::atan2(H[1] - H[2], H[0] + H[3]) : std::atan2(H[1] + H[2], H[0] - H[3]);
  const double c = std::cos(th), s = std::sin(th);
  tf3[0] = c; tf3[1] = -s; tf3[3] = s; tf3[4] = c;
  tf3[2] = cb[0] - (c * ca[0] - s * ca[1]);
  tf3[5] = cb[1] - (s * ca[0] + c * ca[1]);
  tf3[6] = 0; tf3[7] = 0; tf3[8] = 1;
}

struct SemanticClipperOut {
  bool ok = false;
  int n_putative = 0, n_inliers = 0;
  double tf16[16];
  std::vector<int> inliers;
};

// u0 explicit (length 3 * matched pairs) — see clipper.hpp on the reference's unseeded generator
inline SemanticClipperOut semantic_clipper(const double* tm, int ntm, const double* td, int ntd, ClipperParams P, int min_num_pairs,
                                           double matching_threshold, const double* u0) {
  SemanticClipperOut out;
  for (int i = 0; i < 16; ++i) out.tf16[i] = (i % 5 == 0) ? 1.0 : 0.0;
  std::vector<double> pts, diffs;
  const size_t np = match_triangles(tm, ntm, td, ntd, matching_threshold, pts, diffs);
  const int m = (int)(3 * np);
  out.n_putative = m;
  if (m == 0) return out;
  std::vector<double> D1(2 * (size_t)m), D2(2 * (size_t)m);
  for (int i = 0; i < m; ++i) { D1[2 * i] = pts[4 * i]; D1[2 * i + 1] = pts[4 * i + 1]; D2[2 * i] = pts[4 * i + 2]; D2[2 * i + 1] = pts[4 * i + 3]; }
  std::vector<int> A(2 * (size_t)m);
  for (int i = 0; i < m; ++i) { A[2 * i] = i; A[2 * i + 1] = i; }
  std::vector<double> M;
  clipper_affinity(D1.data(), m, D2.data(), m, 2, A, P, M);
  std::vector<double> u(u0, u0 + m);
  ClipperSolution sol = clipper_dense_clique(M, (size_t)m, u, P);
  out.inliers = sol.nodes;
  out.n_inliers = (int)sol.nodes.size();
  if (out.n_inliers < min_num_pairs) return out;
  std::vector<double> a(2 * sol.nodes.size()), b(2 * sol.nodes.size());
  for (size_t k = 0; k < sol.nodes.size(); ++k) {
    const int i = sol.nodes[k];
    a[2 * k] = D1[2 * i]; a[2 * k + 1] = D1[2 * i + 1]; b[2 * k] = D2[2 * i]; b[2 * k + 1] = D2[2 * i + 1];
  }
  double tf3[9];
  estimate_tf2d(a.data(), b.data(), (int)sol.nodes.size(), tf3);
  const double yaw = std::atan2(tf3[3], tf3[0]);
  out.tf16[0] = std::cos(yaw); out.tf16[1] = -std::sin(yaw); out.tf16[4] = std::sin(yaw); out.tf16[5] = std::cos(yaw);
  out.tf16[3] = tf3[2]; out.tf16[7] = tf3[5];
  out.ok = true;
  return out;
}

}  // namespace orc
