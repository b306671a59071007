// ORACLE — TEST INFRASTRUCTURE ONLY (see lie.hpp header).
//
// CPU restatement of the reference per-frame semantic data association:
//   object distance / project   backend/sloam/src/objects/{cube.cpp:22-36, ellipsoid.cpp:24-38, cylinder.cpp:187-242}
//   K-NN submap gate            src/core/{cylinderMapManager.cpp:213-243, cubeMapManager.cpp:36-75, ellipsoidMapManager.cpp:40-80}
//   matchers                    src/core/sloam.cpp:73-203
//   updateMap                   src/core/{cylinderMapManager.cpp:35-68, cubeMapManager.cpp:104-130, ellipsoidMapManager.cpp:111-145}
// The reference's K-NN runs in pcl::KdTreeFLANN<PointXYZI> (third party, absent): FLANN's exact
// K-NN with L2_Simple<float> is restated as a brute-force float32 squared distance
// ((dx*dx + dy*dy) + dz*dz accumulated in float, x->y->z) sorted ascending; equal distances are
// ordered by map index (FLANN's order among exact ties is implementation-defined — documented choice).
#pragma once
#include <algorithm>
#include <vector>

#include "lie.hpp"

namespace orc {

struct CylObj {
  double root[3], ray[3];
  double radius;
  int label;
};
struct BoxObj {  // Cube and Ellipsoid share this shape (pose, scale, label)
  Pose pose;
  double scale[3];
  int label;
};

// Cylinder::project cylinder.cpp:236-242
inline void cyl_project(CylObj& c, const Pose& tf) {
  double other[3] = {c.root[0] + c.ray[0], c.root[1] + c.ray[1], c.root[2] + c.ray[2]};
  double nr[3], no[3];
  pose_transform_from(tf, c.root, nr);
  pose_transform_from(tf, other, no);
  for (int i = 0; i < 3; ++i) { c.root[i] = nr[i]; c.ray[i] = no[i] - nr[i]; }
}
// Cube::project cube.cpp:31-36 / Ellipsoid::project ellipsoid.cpp:33-38
inline void box_project(BoxObj& b, const Pose& tf) { b.pose = pose_compose(tf, b.pose); }

// Cylinder::distance cylinder.cpp:187-224 (use_position_only == true)
inline double cyl_distance(const CylObj& model, const CylObj& tgt) {
  if (tgt.label != model.label) return 1000;
  const double heights[3] = {0.0, 3.0, 6.0};
  double distance = 10000.0;
  for (int h = 0; h < 3; ++h) {
    const double src_t = (heights[h] - model.root[2]) / model.ray[2];
    const double tgt_t = (heights[h] - tgt.root[2]) / tgt.ray[2];
    double d[3];
    for (int i = 0; i < 3; ++i) d[i] = (model.root[i] + src_t * model.ray[i]) - (tgt.root[i] + tgt_t * tgt.ray[i]);
    const double dist = norm3(d);
    if (dist < distance) distance = dist;
  }
  return distance;
}
// Cube::distance cube.cpp:22-24 / Ellipsoid::distance ellipsoid.cpp:24-26
inline double box_distance(const BoxObj& model, const BoxObj& in) {
  double d[3] = {in.pose.t[0] - model.pose.t[0], in.pose.t[1] - model.pose.t[1], in.pose.t[2] - model.pose.t[2]};
  return norm3(d);
}

// Exact float32 K-NN (restated FLANN).  cloud = xyz triples; returns map indices, nearest first.
inline void knn_f32(const std::vector<float>& cloud, const double* query_d, int K, std::vector<int>& out) {
  out.clear();
  const int n = (int)(cloud.size() / 3);
  if (n == 0) return;
  const float qx = (float)query_d[0], qy = (float)query_d[1], qz = (float)query_d[2];
  std::vector<std::pair<float, int>> d(n);
  for (int i = 0; i < n; ++i) {
    const float dx = cloud[3 * i] - qx, dy = cloud[3 * i + 1] - qy, dz = cloud[3 * i + 2] - qz;
    float r = dx * dx;
    r += dy * dy;
    r += dz * dz;
    d[i] = {r, i};
  }
  const int k = std::min(K, n);
  std::partial_sort(d.begin(), d.begin() + k, d.end());
  out.resize(k);
  for (int i = 0; i < k; ++i) out[i] = d[i].second;
}

struct MatchParams {
  double cyl_thresh = 2.0;    // sloamNode.cpp:151
  double cube_thresh = 2.0;   // :152
  double ell_thresh = 0.75;   // :153
};

// sloam.cpp:73-111
inline void match_cylinders(const std::vector<CylObj>& cur, const std::vector<CylObj>& map, double thresh,
                            std::vector<int>& idx) {
  if (cur.empty() || map.empty()) return;
  for (size_t o = 0; o < cur.size(); ++o) {
    double best = thresh + 100;
    size_t bestKey = 0;
    for (size_t k = 0; k < map.size(); ++k) {
      const double d = cyl_distance(map[k], cur[o]);
      if (d < best) { best = d; bestKey = k; }
    }
    if (best < thresh) idx[o] = (int)bestKey;
  }
}
// sloam.cpp:113-156
inline void match_cubes(const std::vector<BoxObj>& cur, const std::vector<BoxObj>& map, double thresh,
                        std::vector<int>& idx) {
  if (cur.empty() || map.empty()) return;
  for (size_t o = 0; o < cur.size(); ++o) {
    double best = 30;
    size_t bestKey = 0;
    for (size_t k = 0; k < map.size(); ++k) {
      const double d = box_distance(map[k], cur[o]);
      if (d < best) { best = d; bestKey = k; }
    }
    if (best < thresh) idx[o] = (int)bestKey;
  }
}
// sloam.cpp:158-203
inline void match_ellipsoids(const std::vector<BoxObj>& cur, const std::vector<BoxObj>& map, double thresh,
                             std::vector<int>& idx) {
  if (cur.empty() || map.empty()) return;
  for (size_t o = 0; o < cur.size(); ++o) {
    double best = 1000;
    size_t bestKey = 0;
    for (size_t k = 0; k < map.size(); ++k) {
      if (map[k].label == cur[o].label) {
        const double d = box_distance(map[k], cur[o]);
        if (d < best) { best = d; bestKey = k; }
      }
    }
    if (best < thresh) idx[o] = (int)bestKey;
  }
}

template <class Obj>
struct MapManager {
  std::vector<Obj> models;
  std::vector<int> hits;
  std::vector<float> cloud;       // first-seen positions, float32, never updated (cubeMapManager.cpp:116-120)
  std::vector<int> matchesMap;    // submap idx -> map idx (std::map<int,int> in the reference)
  int K;
  explicit MapManager(int k) : K(k) {}

  // getSubmap: K = 50 / 30 / 1000 (cylinderMapManager.cpp:230, cubeMapManager.cpp:61, ellipsoidMapManager.cpp:65).
  // NB cylinderMapManager.cpp:215 returns BEFORE clearing the submap when the cloud is empty; the
  // cube/ellipsoid managers clear first — both yield an empty submap from an empty map.
  void getSubmap(const Pose& pose, std::vector<Obj>& submap) {
    submap.clear();
    if (cloud.empty()) return;
    matchesMap.clear();
    knn_f32(cloud, pose.t, K, matchesMap);
    for (int mi : matchesMap) submap.push_back(models[mi]);
  }
};

inline void update_cyl_map(MapManager<CylObj>& M, const std::vector<CylObj>& obs, const std::vector<int>& matches) {
  for (size_t i = 0; i < obs.size(); ++i) {
    if (matches[i] == -1) {
      M.cloud.push_back((float)obs[i].root[0]);
      M.cloud.push_back((float)obs[i].root[1]);
      M.cloud.push_back((float)obs[i].root[2]);
      M.models.push_back(obs[i]);
      M.hits.push_back(1);
    } else {
      M.hits[M.matchesMap.at(matches[i])] += 1;
    }
  }
}
inline void update_box_map(MapManager<BoxObj>& M, const std::vector<BoxObj>& obs, const std::vector<int>& matches,
                           bool ema_scale) {
  for (size_t i = 0; i < obs.size(); ++i) {
    if (matches[i] == -1) {
      M.cloud.push_back((float)obs[i].pose.t[0]);
      M.cloud.push_back((float)obs[i].pose.t[1]);
      M.cloud.push_back((float)obs[i].pose.t[2]);
      M.models.push_back(obs[i]);
      M.hits.push_back(1);
    } else {
      const int mi = M.matchesMap.at(matches[i]);
      M.hits[mi] += 1;
      if (ema_scale) {  // ellipsoidMapManager.cpp:138-141, alpha = 0.2
        const double alpha = 0.2;
        for (int k = 0; k < 3; ++k) M.models[mi].scale[k] = (1. - alpha) * M.models[mi].scale[k] + alpha * obs[i].scale[k];
      }
    }
  }
}

}  // namespace orc
