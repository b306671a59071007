// ORACLE — TEST INFRASTRUCTURE ONLY (see lie.hpp header).
//
// CPU restatement of the CLIPPER fork the reference vendors (these sources ARE in the
// reference tree, so this part of the oracle is pinned by the reference's own known-answer
// tests: test/affinity_test.cpp:33-107 and test/clipper_test.cpp:14-67):
//   CLIPPER::scorePairwiseConsistency  clipper_semantic_object/src/clipper.cpp:21-65
//   EuclideanDistance::operator()      src/invariants/euclidean_distance.cpp:13-31
//   CLIPPER::findDenseClique           src/clipper.cpp:172-323
//   utils::k2ij / findIndicesOfkLargest / createAllToAll   src/utils.cpp:33-97
// Dense row-major storage (the reference converts the dense m x m matrix to sparse; the
// arithmetic is the same).  u0 is an explicit input: the reference draws it from a
// std::random_device-seeded mt19937 (utils.cpp:22-29), i.e. it is not reproducible there.
#pragma once
#include <cmath>
#include <cstddef>
#include <functional>
#include <queue>
#include <utility>
#include <vector>

namespace orc {

struct ClipperParams {           // clipper.h:27-60
  double tol_u = 1e-8, tol_F = 1e-9;
  int maxiniters = 200, maxoliters = 1000;
  double beta = 0.25;
  int maxlsiters = 99;
  double eps = 1e-9;
  double affinityeps = 1e-4;
  bool rescale_u0 = true;
  // EuclideanDistance::Params euclidean_distance.h:24-29
  double sigma = 0.01, epsilon = 0.06, mindist = 0.0;
};

// utils.cpp:87-97
inline void clipper_k2ij(size_t k, size_t n, size_t& i, size_t& j) {
  k += 1;
  const size_t l = n * (n - 1) / 2 - k;
  const size_t o = (size_t)std::floor((std::sqrt(1.0 + 8.0 * (double)l) - 1.0) / 2.0);
  const size_t p = l - o * (o + 1) / 2;
  i = n - (o + 1) - 1;
  j = n - p - 1;
}

// euclidean_distance.cpp:13-31; points are dim-vectors
inline double clipper_score(const double* ai, const double* aj, const double* bi, const double* bj, int dim,
                            const ClipperParams& P) {
  double s1 = 0, s2 = 0;
  for (int k = 0; k < dim; ++k) {
    s1 += (ai[k] - aj[k]) * (ai[k] - aj[k]);
    s2 += (bi[k] - bj[k]) * (bi[k] - bj[k]);
  }
  const double l1 = std::sqrt(s1), l2 = std::sqrt(s2);
  if (P.mindist > 0 && (l1 < P.mindist || l2 < P.mindist)) return 0.0;
  const double c = std::fabs(l1 - l2);
  return (c < P.epsilon) ? std::exp(-0.5 * c * c / (P.sigma * P.sigma)) : 0.0;
}

// clipper.cpp:21-65.  D1: dim x n1 (column j = point j, stored point-major: D1[j*dim + k]),
// A: m x 2 association list (empty -> all-to-all, i-major: k = i * n2 + j, utils createAllToAll).
// Output M: m x m upper-triangular affinities (zero diagonal), as the reference's M_.
inline void clipper_affinity(const double* D1, int n1, const double* D2, int n2, int dim, std::vector<int>& A,
                             const ClipperParams& P, std::vector<double>& M) {
  if (A.empty()) {
    for (int i = 0; i < n1; ++i)
      for (int j = 0; j < n2; ++j) { A.push_back(i); A.push_back(j); }
  }
  const size_t m = A.size() / 2;
  M.assign(m * m, 0.0);
  if (m < 2) return;
  for (size_t k = 0; k < m * (m - 1) / 2; ++k) {
    size_t i, j;
    clipper_k2ij(k, m, i, j);
    if (A[2 * i] == A[2 * j] || A[2 * i + 1] == A[2 * j + 1]) continue;
    const double scr = clipper_score(D1 + (size_t)A[2 * i] * dim, D1 + (size_t)A[2 * j] * dim,
                                     D2 + (size_t)A[2 * i + 1] * dim, D2 + (size_t)A[2 * j + 1] * dim, dim, P);
    if (scr > P.affinityeps) M[i * m + j] = scr;
  }
}

struct ClipperSolution {
  std::vector<int> nodes;
  std::vector<double> u;
  double score = 0;
  int ifinal = 0;
};

// utils.cpp:33-55
inline std::vector<int> clipper_k_largest(const std::vector<double>& x, int k) {
  using T = std::pair<double, int>;
  if (k < 1) return {};
  std::priority_queue<T, std::vector<T>, std::greater<T>> q;
  for (size_t i = 0; i < x.size(); ++i) {
    if ((int)q.size() < k) q.push({x[i], (int)i});
    else if (q.top().first < x[i]) { q.pop(); q.push({x[i], (int)i}); }
  }
  const int kk = (int)q.size();
  std::vector<int> idx(kk);
  for (int i = 0; i < kk; ++i) { idx[kk - i - 1] = q.top().second; q.pop(); }
  return idx;
}

// clipper.cpp:172-323 with rounding = DSD_HEU (the only mode sloam uses, semantic_clipper.cpp:227-233).
// Mup: m x m upper-triangular affinity (zero diagonal); C = sparsity pattern of Mup.
inline ClipperSolution clipper_dense_clique(const std::vector<double>& Mup, size_t n, const std::vector<double>& u0,
                                            const ClipperParams& P) {
  ClipperSolution sol;
  if (n == 0) return sol;
  // symmetric products: (selfadjointView<Upper> * v)
  std::vector<double> Ms(n * n, 0.0), Cs(n * n, 0.0);
  for (size_t i = 0; i < n; ++i)
    for (size_t j = i + 1; j < n; ++j) {
      const double v = Mup[i * n + j];
      Ms[i * n + j] = Ms[j * n + i] = v;
      if (v != 0.0) Cs[i * n + j] = Cs[j * n + i] = 1.0;
    }
  auto symv = [&](const std::vector<double>& A, const std::vector<double>& v, std::vector<double>& out) {
    for (size_t i = 0; i < n; ++i) {
      double s = 0;
      for (size_t j = 0; j < n; ++j) s += A[i * n + j] * v[j];
      out[i] = s;
    }
  };
  auto vsum = [&](const std::vector<double>& v) { double s = 0; for (double x : v) s += x; return s; };
  auto vnorm = [&](const std::vector<double>& v) { double s = 0; for (double x : v) s += x * x; return std::sqrt(s); };
  std::vector<double> u(n), unew(n), gradF(n), gradFnew(n), Mu(n), Cu(n), Cbu(n);
  if (P.rescale_u0) {
    symv(Ms, u0, Mu);
    for (size_t i = 0; i < n; ++i) u[i] = Mu[i] + u0[i];
  } else {
    u = u0;
  }
  { const double nn = vnorm(u); for (auto& x : u) x /= nn; }

  auto compute_d_terms = [&](const std::vector<double>& uu, double& num_over_den_mean, bool absval) -> int {
    const double su = vsum(uu);
    symv(Cs, uu, Cu);
    for (size_t i = 0; i < n; ++i) Cbu[i] = su - Cu[i] - uu[i];
    int cnt = 0;
    double acc = 0;
    bool have_Mu = false;
    for (size_t i = 0; i < n; ++i) {
      if (Cbu[i] > P.eps && uu[i] > P.eps) {
        if (!have_Mu) { symv(Ms, uu, Mu); have_Mu = true; }
        const double q = (Mu[i] + uu[i]) / Cbu[i];
        acc += absval ? std::fabs(q) : q;
        ++cnt;
      }
    }
    num_over_den_mean = cnt ? acc / cnt : 0.0;
    return cnt;
  };
  double d = 0, tmp;
  if (compute_d_terms(u, tmp, false) > 0) d = tmp;

  auto grad = [&](const std::vector<double>& uu, double dd, std::vector<double>& g) {
    const double su = vsum(uu);
    symv(Ms, uu, Mu);
    symv(Cs, uu, Cu);
    for (size_t i = 0; i < n; ++i) g[i] = (1 + dd) * uu[i] - dd * su + Mu[i] + Cu[i] * dd;
  };
  auto dot = [&](const std::vector<double>& a, const std::vector<double>& b) {
    double s = 0;
    for (size_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
  };
  double F = 0;
  int i;
  for (i = 0; i < P.maxoliters; ++i) {
    grad(u, d, gradF);
    F = dot(u, gradF);
    for (int j = 0; j < P.maxiniters; ++j) {
      double alpha = 1, Fnew = 0, deltaF = 0;
      for (int k = 0; k < P.maxlsiters; ++k) {
        for (size_t q = 0; q < n; ++q) unew[q] = std::max(u[q] + alpha * gradF[q], 0.0);
        const double nn = vnorm(unew);
        for (auto& x : unew) x /= nn;
        grad(unew, d, gradFnew);
        Fnew = dot(unew, gradFnew);
        deltaF = Fnew - F;
        if (deltaF < -P.eps) alpha *= P.beta;
        else break;
      }
      double du = 0;
      for (size_t q = 0; q < n; ++q) du += (unew[q] - u[q]) * (unew[q] - u[q]);
      du = std::sqrt(du);
      F = Fnew;
      u = unew;
      gradF = gradFnew;
      if (du < P.tol_u || std::fabs(deltaF) < P.tol_F) break;
    }
    double deltad;
    if (compute_d_terms(u, deltad, true) > 0) d += deltad;
    else break;
  }
  const int omega = (int)std::round(F);
  sol.nodes = clipper_k_largest(u, omega);
  sol.u = u;
  sol.score = F;
  sol.ifinal = i;
  return sol;
}

}  // namespace orc
