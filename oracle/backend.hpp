// ORACLE — TEST INFRASTRUCTURE ONLY (see lie.hpp header).
//
// CPU restatement of the reference per-key-frame update, ROS plumbing removed:
//   SLOAMNode::runSLOAMNode                      backend/sloam/src/core/sloamNode.cpp:762-1036
//   sloam::RunSloam / projectModels              src/core/sloam.cpp:205-306
//   SemanticFactorGraphWrapper::addSLOAMObservation   src/factorgraph/graphWrapper.cpp:99-237
//   updateFactorGraphMap / getCurrPose           src/factorgraph/graphWrapper.cpp:239-297
//   inter-robot packet ingestion                 src/core/sloamNode.cpp:912-1002
#pragma once
#include <vector>

#include "assoc.hpp"
#include "graph.hpp"

namespace orc {

struct Detections {
  std::vector<CylObj> cyl;    // body frame
  std::vector<BoxObj> cube;   // body frame
  std::vector<BoxObj> ell;    // body frame
};

struct FrameResult {
  std::vector<int> cyl_match, cube_match, ell_match;        // submap indices or -1 (sloam.cpp:224-226)
  std::vector<int> cyl_map_idx, cube_map_idx, ell_map_idx;  // global landmark ids actually used in the graph
  Pose out_pose;
  bool optimized = false;
  int solve_status = 0;
  double t_assoc = 0, t_graph = 0;  // the two timers the reference keeps (sloamNode.cpp:845-849, 888-897)
};

class Backend {
 public:
  Graph graph;
  MatchParams mp;
  MapManager<CylObj> cylMap{50};
  MapManager<BoxObj> cubeMap{30};
  MapManager<BoxObj> ellMap{1000};
  bool firstScan = true;                      // sloam.cpp:17,235-248
  // graphWrapper.h:131-134 id allocators
  uint64_t cyl_counter = 0, cube_counter = 0, point_counter = 0;
  std::vector<uint64_t> pose_counter;
  std::vector<int> point_labels;              // graph.h:116

  explicit Backend(int num_robots = 13) : pose_counter(num_robots, 0) {}

  // graphWrapper.cpp:99-237.  Detections are in the WORLD frame here.
  bool addSLOAMObservation(const std::vector<int>& cyl_m, const std::vector<CylObj>& cyls,
                           const std::vector<int>& cube_m, const std::vector<BoxObj>& cubes,
                           const std::vector<int>& ell_m, const std::vector<BoxObj>& ells, const Pose& rel,
                           const Pose& pose, int robot, bool opt, FrameResult* res) {
    const uint64_t pc = pose_counter[robot];
    if (pc == 0) graph.setPriors(pose, robot);
    else graph.addKeyPoseAndBetween(pc - 1, pc, rel, pose, robot);
    for (size_t i = 0; i < cyl_m.size(); ++i) {
      uint64_t id;
      if (cyl_m[i] == -1) {
        id = cyl_counter++;
        graph.addCylinderFactor(pc, id, pose, cyls[i].root, cyls[i].ray, cyls[i].radius, false, robot);
      } else {
        id = (uint64_t)cylMap.matchesMap.at(cyl_m[i]);
        graph.addCylinderFactor(pc, id, pose, cyls[i].root, cyls[i].ray, cyls[i].radius, true, robot);
      }
      if (res) res->cyl_map_idx.push_back((int)id);
    }
    for (size_t i = 0; i < cube_m.size(); ++i) {
      uint64_t id;
      if (cube_m[i] == -1) {
        id = cube_counter++;
        graph.addCubeFactor(pc, id, pose, cubes[i].pose, cubes[i].scale, false, robot);
      } else {
        id = (uint64_t)cubeMap.matchesMap.at(cube_m[i]);
        graph.addCubeFactor(pc, id, pose, cubes[i].pose, cubes[i].scale, true, robot);
      }
      if (res) res->cube_map_idx.push_back((int)id);
    }
    // ellipsoids -> Point3 landmarks with bearing-range factors (graphWrapper.cpp:157-202)
    for (size_t i = 0; i < ell_m.size(); ++i) {
      double body[3];
      pose_transform_to(pose, ells[i].pose.t, body);   // (pose^-1 * ell_world).translation()
      const double range = norm3(body);
      double bearing[3] = {body[0] / range, body[1] / range, body[2] / range};
      uint64_t id;
      if (ell_m[i] == -1) {
        id = point_counter++;
        graph.addPointLandmarkKey(id, ells[i].pose.t);
        graph.addRangeBearingFactor(pc, id, bearing, range, robot);
        point_labels.push_back(ells[i].label);
      } else {
        id = (uint64_t)ellMap.matchesMap.at(ell_m[i]);
        graph.addRangeBearingFactor(pc, id, bearing, range, robot);
      }
      if (res) res->ell_map_idx.push_back((int)id);
    }
    pose_counter[robot] = pc + 1;
    if (opt) {
      const int st = graph.solve();
      if (res) res->solve_status = st;
      return true;
    }
    return false;
  }

  // graphWrapper.cpp:239-275
  void updateFactorGraphMap() {
    for (uint64_t i = 0; i < cyl_counter; ++i) {
      const Var* v = graph.getLandmark('l', i);
      if (!v) continue;  // Values::at would throw in the reference
      for (int k = 0; k < 3; ++k) { cylMap.models[i].root[k] = v->val[k]; cylMap.models[i].ray[k] = v->val[3 + k]; }
      cylMap.models[i].radius = v->val[6];
    }
    for (uint64_t i = 0; i < cube_counter; ++i) {
      const Var* v = graph.getLandmark('c', i);
      if (!v) continue;
      cubeMap.models[i].pose = var_pose(*v);
      for (int k = 0; k < 3; ++k) cubeMap.models[i].scale[k] = v->val[12 + k];
    }
    for (uint64_t i = 0; i < point_counter; ++i) {
      const Var* v = graph.getLandmark('u', i);
      Pose T;
      pose_identity(T);  // updateEllipsoid: Pose3(Rot3(), point); absent key -> Point3() (graph.cpp:282-288)
      if (v) { T.t[0] = v->val[0]; T.t[1] = v->val[1]; T.t[2] = v->val[2]; }
      ellMap.models[i].pose = T;
    }
  }

  // sloamNode.cpp:830-869 + RunSloam: submap gate, projection, matching, map update.
  void associate(const Pose& poseEstimate, const Detections& body, bool allow_first_scan_shortcut,
                 std::vector<CylObj>& cylW, std::vector<BoxObj>& cubeW, std::vector<BoxObj>& ellW, FrameResult& res) {
    std::vector<CylObj> subCyl;
    std::vector<BoxObj> subCube, subEll;
    cylMap.getSubmap(poseEstimate, subCyl);
    cubeMap.getSubmap(poseEstimate, subCube);
    ellMap.getSubmap(poseEstimate, subEll);
    cylW = body.cyl; cubeW = body.cube; ellW = body.ell;
    res.cyl_match.assign(cylW.size(), -1);
    res.cube_match.assign(cubeW.size(), -1);
    res.ell_match.assign(ellW.size(), -1);
    for (auto& c : cylW) cyl_project(c, poseEstimate);
    for (auto& c : cubeW) box_project(c, poseEstimate);
    for (auto& c : ellW) box_project(c, poseEstimate);
    if (allow_first_scan_shortcut && firstScan) {
      firstScan = false;  // sloam.cpp:235-248: no matching on the very first scan
    } else {
      match_cylinders(cylW, subCyl, mp.cyl_thresh, res.cyl_match);
      match_cubes(cubeW, subCube, mp.cube_thresh, res.cube_match);
      match_ellipsoids(ellW, subEll, mp.ell_thresh, res.ell_match);
    }
    update_cyl_map(cylMap, cylW, res.cyl_match);
    update_box_map(cubeMap, cubeW, res.cube_match, false);
    update_box_map(ellMap, ellW, res.ell_match, true);
  }

  // runSLOAMNode host-robot branch (sloamNode.cpp:785-897, 1010-1014, 1028)
  // defer_map_update = true reproduces the reference order when foreign packets are pending:
  // host add+solve -> ingest loop (maps NOT yet refreshed) -> one solve -> updateFactorGraphMap.
  FrameResult process_frame(int robot, const Pose& relMotion, const Pose& prevKeyPose, const Detections& body,
                            bool defer_map_update = false) {
    FrameResult res;
    const Pose poseEstimate = pose_compose(prevKeyPose, relMotion);
    std::vector<CylObj> cylW;
    std::vector<BoxObj> cubeW, ellW;
    const double t0 = now_sec();
    associate(poseEstimate, body, true, cylW, cubeW, ellW, res);
    const double t1 = now_sec();
    res.optimized = addSLOAMObservation(res.cyl_match, cylW, res.cube_match, cubeW, res.ell_match, ellW, relMotion,
                                        poseEstimate, robot, true, &res);
    res.out_pose = poseEstimate;
    if (res.optimized && res.solve_status == 0 && !defer_map_update) {
      updateFactorGraphMap();
      graph.getPose(pose_counter[robot] - 1, robot, res.out_pose);
    }
    const double t2 = now_sec();
    res.t_assoc = t1 - t0;
    res.t_graph = t2 - t1;
    return res;
  }

  // One foreign packet (sloamNode.cpp:938-999): pose already in the host frame, no solve.
  FrameResult ingest_packet(int robot, const Pose& relMotion, const Pose& poseInHostFrame, const Detections& body) {
    FrameResult res;
    std::vector<CylObj> cylW;
    std::vector<BoxObj> cubeW, ellW;
    associate(poseInHostFrame, body, false, cylW, cubeW, ellW, res);
    addSLOAMObservation(res.cyl_match, cylW, res.cube_match, cubeW, res.ell_match, ellW, relMotion, poseInHostFrame,
                        robot, false, &res);
    res.out_pose = poseInHostFrame;
    return res;
  }
  // sloamNode.cpp:1000: one solve() per foreign robot after its batch of packets
  int ingest_solve() { return graph.solve(); }
  // sloamNode.cpp:1010-1014: refresh the map from the optimised landmarks, fetch the host's current pose
  bool end_frame(int robot, Pose& out) {
    updateFactorGraphMap();
    return graph.getPose(pose_counter[robot] - 1, robot, out);
  }
};

}  // namespace orc
