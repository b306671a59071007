// ORACLE — TEST INFRASTRUCTURE ONLY (see lie.hpp header).
//
// CPU restatement of the step right before the hot path:
//   Input::PickNextMeasurementToAdd          backend/sloam/src/core/input.cpp:26-108   (pinned by src/test/input_test.cpp)
//   CylinderMapManager::InLoopClosureRegion  backend/sloam/src/core/cylinderMapManager.cpp:115-160
// Queues are flat arrays (front = index 0); the function returns how many entries the reference pops from each front.
// ros::Time: (sec, nsec), operator< lexicographic, toSec() = sec + 1e-9 nsec.
#pragma once
#include <algorithm>
#include <utility>
#include <vector>
#include <cmath>
#include <cstdint>

#include "lie.hpp"

namespace orc {

struct PickResult { int meas_to_add, pop_odom, pop_obs, pop_rel; };

inline bool stamp_lt(int64_t as, int64_t an, int64_t bs, int64_t bn) { return as < bs || (as == bs && an < bn); }
inline double stamp_sec(int64_t s, int64_t n) { return (double)s + 1e-9 * (double)n; }

// odom_pose12 / latest12: R row-major (9) + t (3)
inline PickResult pick_next_measurement(const int64_t* odom_sec, const int64_t* odom_nsec, const double* odom_pose12, int n_odom,
                                        const int64_t* obs_sec, const int64_t* obs_nsec, int n_obs, const int64_t* rel_sec,
                                        const int64_t* rel_nsec, int n_rel, int64_t latest_sec, int64_t latest_nsec,
                                        const double* latest12, double current_time, double msg_delay_tolerance,
                                        float min_odom_distance) {
  PickResult r{0, 0, 0, 0};
  while (r.pop_odom < n_odom && stamp_lt(odom_sec[r.pop_odom], odom_nsec[r.pop_odom], latest_sec, latest_nsec)) ++r.pop_odom;
  while (r.pop_obs < n_obs && stamp_lt(obs_sec[r.pop_obs], obs_nsec[r.pop_obs], latest_sec, latest_nsec)) ++r.pop_obs;
  while (r.pop_rel < n_rel && stamp_lt(rel_sec[r.pop_rel], rel_nsec[r.pop_rel], latest_sec, latest_nsec)) ++r.pop_rel;
  bool validObs = false, validRel = false;
  if (r.pop_obs < n_obs) validObs = (current_time - stamp_sec(obs_sec[r.pop_obs], obs_nsec[r.pop_obs])) >= msg_delay_tolerance;
  if (r.pop_rel < n_rel) validRel = (current_time - stamp_sec(rel_sec[r.pop_rel], rel_nsec[r.pop_rel])) >= msg_delay_tolerance;
  if (validObs && validRel) {
    r.meas_to_add = stamp_lt(obs_sec[r.pop_obs], obs_nsec[r.pop_obs], rel_sec[r.pop_rel], rel_nsec[r.pop_rel]) ? 2 : 3;
    return r;
  }
  if (validObs || validRel) { r.meas_to_add = validObs ? 2 : 3; return r; }
  // newest-first scan of the remaining odometry queue (input.cpp:84-104)
  for (int i = n_odom - 1; i >= r.pop_odom; --i) {
    if ((current_time - stamp_sec(odom_sec[i], odom_nsec[i])) >= msg_delay_tolerance) {
      // latest.pose.inverse() * odom_i.pose  (Sophus: t = R_l^T t_i + R_l^T (-t_l))
      const double* Rl = latest12;
      const double* tl = latest12 + 9;
      const double* ti = odom_pose12 + 12 * (size_t)i + 9;
      double t[3];
      for (int k = 0; k < 3; ++k) {
        const double a = Rl[0 * 3 + k] * ti[0] + Rl[1 * 3 + k] * ti[1] + Rl[2 * 3 + k] * ti[2];
        const double b = Rl[0 * 3 + k] * (-tl[0]) + Rl[1 * 3 + k] * (-tl[1]) + Rl[2 * 3 + k] * (-tl[2]);
        t[k] = a + b;
      }
      const double moved = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
      if (moved > (double)min_odom_distance) {
        r.meas_to_add = 1;
        r.pop_odom = i;          // everything older than entry i goes; entry i becomes the front
        return r;
      }
      break;
    }
  }
  r.meas_to_add = 0;
  return r;
}

// cloud: n key-pose positions as float32 xyz (robotPoseCloud_); FLANN radius search = strict '<' on float squared distances
inline bool in_loop_closure_region(const float* cloud, int n, const double* pose_t, double max_dist_xy, double max_dist_z,
                                   uint64_t at_least_num_of_poses_old) {
  if ((uint64_t)n < at_least_num_of_poses_old) return false;
  const float sx = (float)pose_t[0], sy = (float)pose_t[1], sz = (float)pose_t[2];
  const double max3 = std::sqrt(max_dist_xy * max_dist_xy + max_dist_z * max_dist_z);
  const float r2 = (float)(max3 * max3);
  for (int i = 0; i < n; ++i) {
    const float dx = cloud[3 * i] - sx, dy = cloud[3 * i + 1] - sy, dz = cloud[3 * i + 2] - sz;
    const float d2 = dx * dx + dy * dy + dz * dz;
    if (!(d2 < r2)) continue;
    const double dxy = std::sqrt(std::pow((double)(cloud[3 * i] - sx), 2) + std::pow((double)(cloud[3 * i + 1] - sy), 2));
    const double dzz = std::fabs((double)(cloud[3 * i + 2] - sz));
    if (dxy > max_dist_xy || dzz > max_dist_z) continue;
    if ((uint64_t)(n - 1) - (uint64_t)i > at_least_num_of_poses_old) return true;
  }
  return false;
}

// CylinderMapManager::getLoopCandidateIdx cylinderMapManager.cpp:160-184.  kdtree.radiusSearch returns the neighbours sorted by
// squared float distance; the loop takes the first one with nnIdx != poseIdx && poseIdx - nnIdx > at_least (size_t arithmetic: an
// index above poseIdx wraps around and passes).  Order among exactly equidistant neighbours: by index (FLANN: unspecified).
inline bool loop_candidate_idx(const float* cloud, int n, double max_dist, uint64_t pose_idx, uint64_t at_least, uint64_t* cand) {
  if (n < 50) return false;
  const float* sp = cloud + 3 * pose_idx;
  const float r2 = (float)(max_dist * max_dist);
  std::vector<std::pair<float, int>> nn;
  for (int i = 0; i < n; ++i) {
    const float dx = cloud[3 * i] - sp[0], dy = cloud[3 * i + 1] - sp[1], dz = cloud[3 * i + 2] - sp[2];
    float d2 = dx * dx;
    d2 += dy * dy;
    d2 += dz * dz;
    if (d2 < r2) nn.push_back({d2, i});
  }
  std::stable_sort(nn.begin(), nn.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
  for (const auto& e : nn) {
    const uint64_t idx = (uint64_t)e.second;
    if (idx != pose_idx && pose_idx - idx > at_least) { *cand = idx; return true; }
  }
  return false;
}

}  // namespace orc
