// 2-D Delaunay triangulation on the host (replaces the reference's qhull call, clipper_semantic_object/src/triangulation/
// observation.cpp:13-88, options "Qt Qbb Qc Qz Q12 d"): the triangle SET of a point set in general position is unique,
// so it equals qhull's; only the order of the list differs (here: canonical, vertex ids ascending inside a triangle,
// triangles in lexicographic order), which permutes SlideGraph's putative association list and nothing else.
// Algorithm: sort by (x, y), sweep-hull triangulation of the convex hull (every new point is outside the hull built so
// far and is joined to the hull edges it sees), then Lawson flips until every interior edge is locally Delaunay.
// Predicates in long double; exactly cocircular / collinear inputs are decided by the sign of the rounded determinant.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <vector>

namespace sl {
namespace delaunay {

inline long double orient(const double* a, const double* b, const double* c) {
  return ((long double)b[0] - a[0]) * ((long double)c[1] - a[1]) - ((long double)b[1] - a[1]) * ((long double)c[0] - a[0]);
}
// > 0: d strictly inside the circumcircle of the counter-clockwise triangle (a, b, c)
inline long double incircle(const double* a, const double* b, const double* c, const double* d) {
  const long double ax = (long double)a[0] - d[0], ay = (long double)a[1] - d[1];
  const long double bx = (long double)b[0] - d[0], by = (long double)b[1] - d[1];
  const long double cx = (long double)c[0] - d[0], cy = (long double)c[1] - d[1];
  const long double a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
  return ax * (by * c2 - b2 * cy) - ay * (bx * c2 - b2 * cx) + a2 * (bx * cy - by * cx);
}

struct Tri {
  int v[3];   // counter-clockwise
  int n[3];   // n[i]: triangle across the edge opposite v[i] (v[i+1] -> v[i+2]), -1 on the hull
};

inline int edge_slot(const Tri& t, int a, int b) {   // index i with (v[i+1], v[i+2]) == (a, b)
  for (int i = 0; i < 3; ++i)
    if (t.v[(i + 1) % 3] == a && t.v[(i + 2) % 3] == b) return i;
  return -1;
}

// xy: n points (x, y).  Returns triangles as ascending vertex-id triples in lexicographic order.
inline std::vector<std::array<int32_t, 3>> triangulate(const double* xy, int n) {
  std::vector<std::array<int32_t, 3>> out;
  if (n < 3) return out;
  std::vector<int> ord(n);
  for (int i = 0; i < n; ++i) ord[i] = i;
  std::sort(ord.begin(), ord.end(), [&](int a, int b) {
    if (xy[2 * a] != xy[2 * b]) return xy[2 * a] < xy[2 * b];
    if (xy[2 * a + 1] != xy[2 * b + 1]) return xy[2 * a + 1] < xy[2 * b + 1];
    return a < b;
  });
  // coincident points: keep the first of each (qhull's Qc keeps them out of the facets as "coplanar" points too)
  std::vector<int> pts;
  for (int i = 0; i < n; ++i)
    if (pts.empty() || xy[2 * ord[i]] != xy[2 * pts.back()] || xy[2 * ord[i] + 1] != xy[2 * pts.back() + 1]) pts.push_back(ord[i]);
  const int m = (int)pts.size();
  if (m < 3) return out;
  auto P = [&](int id) { return xy + 2 * (size_t)id; };
  // first point not collinear with the leading ones
  int kfirst = -1;
  for (int i = 2; i < m; ++i)
    if (orient(P(pts[0]), P(pts[1]), P(pts[i])) != 0) { kfirst = i; break; }
  if (kfirst < 0) return out;   // all collinear
  std::vector<Tri> T;
  // hull as a circular doubly linked list over point ids (counter-clockwise); hull_tri[a] = triangle on edge a -> next[a]
  std::vector<int> nxt(n, -1), prv(n, -1), hull_tri(n, -1);
  {
    // fan over the collinear leading run pts[0..kfirst-1] with apex pts[kfirst]
    const int apex = pts[kfirst];
    const bool left = orient(P(pts[0]), P(pts[1]), P(apex)) > 0;
    for (int i = 0; i + 1 < kfirst; ++i) {
      Tri t;
      if (left) { t.v[0] = pts[i]; t.v[1] = pts[i + 1]; t.v[2] = apex; }
      else { t.v[0] = pts[i + 1]; t.v[1] = pts[i]; t.v[2] = apex; }
      t.n[0] = t.n[1] = t.n[2] = -1;
      T.push_back(t);
    }
    for (int i = 0; i + 2 < kfirst; ++i) {       // neighbours inside the fan share the edge (pts[i+1], apex)
      Tri& a = T[i];
      Tri& b = T[i + 1];
      if (left) { a.n[edge_slot(a, pts[i + 1], apex)] = i + 1; b.n[edge_slot(b, apex, pts[i + 1])] = i; }
      else { a.n[edge_slot(a, apex, pts[i + 1])] = i + 1; b.n[edge_slot(b, pts[i + 1], apex)] = i; }
    }
    // hull cycle
    std::vector<int> cyc;
    if (left) { for (int i = 0; i < kfirst; ++i) cyc.push_back(pts[i]); cyc.push_back(apex); }
    else { cyc.push_back(apex); for (int i = kfirst - 1; i >= 0; --i) cyc.push_back(pts[i]); }
    const int h = (int)cyc.size();
    for (int i = 0; i < h; ++i) { nxt[cyc[i]] = cyc[(i + 1) % h]; prv[cyc[(i + 1) % h]] = cyc[i]; }
    for (int i = 0; i < h; ++i) {
      const int a = cyc[i], b = cyc[(i + 1) % h];
      for (int t = 0; t < (int)T.size(); ++t)
        if (edge_slot(T[t], a, b) >= 0) { hull_tri[a] = t; break; }
    }
  }
  int last = pts[kfirst];   // most recently inserted hull vertex (every later point sees an edge at it)
  for (int q = 0; q < m; ++q) {
    if (q <= kfirst) continue;
    const int p = pts[q];
    // visible chain: hull edges a -> nxt[a] with p strictly to their right; it contains an edge incident to `last`
    auto visible = [&](int a) { return orient(P(a), P(nxt[a]), P(p)) < 0; };
    int start = -1;
    if (visible(last)) start = last;
    else if (visible(prv[last])) start = prv[last];
    else {
      // (collinear corner cases) search the whole hull
      int a = last;
      do { if (visible(a)) { start = a; break; } a = nxt[a]; } while (a != last);
      if (start < 0) continue;   // p lies on the hull boundary line: skipped (degenerate)
    }
    int first = start;
    while (visible(prv[first])) first = prv[first];
    int lastv = start;          // last visible edge starts at lastv
    while (visible(nxt[lastv])) lastv = nxt[lastv];
    const int end = nxt[lastv];
    int prev_new = -1;
    int a = first;
    while (a != end) {
      const int b = nxt[a];
      const int told = hull_tri[a];
      Tri t;
      t.v[0] = b; t.v[1] = a; t.v[2] = p;              // counter-clockwise (p is right of a -> b)
      t.n[0] = t.n[1] = t.n[2] = -1;
      const int tn = (int)T.size();
      t.n[2] = told;                                    // edge (b, a) is opposite v[2] = p
      T.push_back(t);
      T[told].n[edge_slot(T[told], a, b)] = tn;
      if (prev_new >= 0) {                              // shares edge (a, p) with the previous new triangle (…, a, p)
        T[tn].n[0] = prev_new;                          // opposite v[0] = b: edge (a, p)
        T[prev_new].n[1] = tn;                          // previous triangle (a, a_prev, p): opposite v[1] = a_prev: edge (p, a)
      }
      prev_new = tn;
      a = b;
    }
    // hull update: first -> p -> end
    const int t_first = hull_tri[first] >= 0 ? -1 : -1;
    (void)t_first;
    // new hull edges: (first, p) belongs to the first new triangle, (p, end) to the last one
    int first_new = -1, last_new = prev_new;
    for (int t = (int)T.size() - 1; t >= 0; --t)
      if (T[t].v[2] == p && T[t].v[1] == first) { first_new = t; break; }
    nxt[first] = p; prv[p] = first; nxt[p] = end; prv[end] = p;
    hull_tri[first] = first_new;
    hull_tri[p] = last_new;
    last = p;
  }
  // Lawson flips
  std::vector<std::pair<int, int>> stack;   // (triangle, edge slot)
  for (int t = 0; t < (int)T.size(); ++t)
    for (int i = 0; i < 3; ++i)
      if (T[t].n[i] > t) stack.push_back({t, i});
  while (!stack.empty()) {
    const int t1 = stack.back().first, i1 = stack.back().second;
    stack.pop_back();
    const int t2 = T[t1].n[i1];
    if (t2 < 0) continue;
    const int a = T[t1].v[(i1 + 1) % 3], b = T[t1].v[(i1 + 2) % 3], c = T[t1].v[i1];
    const int i2 = edge_slot(T[t2], b, a);
    if (i2 < 0) continue;                     // stale entry (the edge was flipped away meanwhile)
    const int d = T[t2].v[i2];
    if (!(incircle(P(a), P(b), P(c), P(d)) > 0)) continue;
    // quadrilateral c, a, d, b: replace diagonal (a, b) by (c, d): t1 = (c, a, d), t2 = (c, d, b)
    const int n_ca = T[t1].n[(i1 + 2) % 3];   // across (c, a)
    const int n_bc = T[t1].n[(i1 + 1) % 3];   // across (b, c)
    const int n_ad = T[t2].n[(i2 + 1) % 3];   // across (a, d)
    const int n_db = T[t2].n[(i2 + 2) % 3];   // across (d, b)
    Tri u, w;
    u.v[0] = c; u.v[1] = a; u.v[2] = d;       // edges: opp c = (a, d), opp a = (d, c), opp d = (c, a)
    u.n[0] = n_ad; u.n[1] = t2; u.n[2] = n_ca;
    w.v[0] = c; w.v[1] = d; w.v[2] = b;       // edges: opp c = (d, b), opp d = (b, c), opp b = (c, d)
    w.n[0] = n_db; w.n[1] = n_bc; w.n[2] = t1;
    T[t1] = u;
    T[t2] = w;
    if (n_ad >= 0) T[n_ad].n[edge_slot(T[n_ad], d, a)] = t1;
    if (n_bc >= 0) T[n_bc].n[edge_slot(T[n_bc], c, b)] = t2;
    // n_ca keeps t1, n_db keeps t2
    for (int i = 0; i < 3; ++i) {
      if (i != 1) stack.push_back({t1, i});
      if (i != 2) stack.push_back({t2, i});
    }
  }
  // hull_tri is stale after flips, but no longer needed
  out.reserve(T.size());
  for (const Tri& t : T) {
    std::array<int32_t, 3> v = {t.v[0], t.v[1], t.v[2]};
    std::sort(v.begin(), v.end());
    out.push_back(v);
  }
  std::sort(out.begin(), out.end());
  return out;
}

}  // namespace delaunay
}  // namespace sl
