// slide_sloam_adaptor.hpp — source-compatible C++ adaptor over the C-ABI of slide_gpu.h (header only, no GTSAM / Sophus / ROS).
//
// The reference calls the hot path through two classes of backend/sloam:
//   S1  SemanticFactorGraph          include/factorgraph/graph.h:70-121        (src/factorgraph/graph.cpp)
//   S2  SemanticFactorGraphWrapper   include/factorgraph/graphWrapper.h:82-134 (src/factorgraph/graphWrapper.cpp)
// The two classes below carry the SAME method names, argument order and return conventions, so that the call sites in
// graphWrapper.cpp / sloamNode.cpp compile against them after a type alias — every pose argument is a template parameter that only
// has to provide what gtsam::Pose3 / Sophus::SE3d provide at those call sites:
//     pose.translation()   -> something indexable [0..2]
//     pose.rotation().toQuaternion() / pose.unit_quaternion()  -> x(), y(), z(), w()
// (slide::Pose7 below is the plain carrier used when neither library is present, e.g. in this repository's compile test).
// Errors: the reference's bool / throw behaviour is kept — getPose() returns false + identity for an absent key (graph.cpp:290-312),
// getCylinder / getCube throw std::out_of_range like gtsam::Values::at (graph.cpp:274-280), getCentroidLandmark returns the zero
// point (graph.cpp:282-288); a device / solver failure throws slide::Error carrying slide_last_error().
#ifndef SLIDE_SLOAM_ADAPTOR_HPP_
#define SLIDE_SLOAM_ADAPTOR_HPP_

#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "slide_gpu.h"

namespace slide {

struct Error : std::runtime_error {
  int code;
  Error(int c, const char* what) : std::runtime_error(std::string(what) + ": " + slide_last_error()), code(c) {}
};

// tx ty tz qx qy qz qw — geometry_msgs/Pose order, T_world<-sensor (graph.h:44)
struct Pose7 {
  double v[7] = {0, 0, 0, 0, 0, 0, 1};
  struct Q { double x_, y_, z_, w_; double x() const { return x_; } double y() const { return y_; } double z() const { return z_; } double w() const { return w_; } };
  std::array<double, 3> translation() const { return {v[0], v[1], v[2]}; }
  Q unit_quaternion() const { return {v[3], v[4], v[5], v[6]}; }
};

namespace detail {
template <class T>
auto quat_of(const T& p, int) -> decltype(p.unit_quaternion()) { return p.unit_quaternion(); }                 // Sophus::SE3d, slide::Pose7
template <class T>
auto quat_of(const T& p, long) -> decltype(p.rotation().toQuaternion()) { return p.rotation().toQuaternion(); }  // gtsam::Pose3
template <class T>
void to7(const T& p, double o[7]) {
  const auto t = p.translation();
  const auto q = quat_of(p, 0);
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
  o[3] = q.x(); o[4] = q.y(); o[5] = q.z(); o[6] = q.w();
}
template <class V>
void to3(const V& p, double o[3]) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
// point of the world frame in the frame of pose7 (R^T (p - t)): graphWrapper.cpp:165-173 computes curr_pose^-1 * ellipsoid_world
inline void to_body(const double T[7], const double p[3], double o[3]) {
  const double x = T[3], y = T[4], z = T[5], w = T[6];
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                       2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                       2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
  const double d[3] = {p[0] - T[0], p[1] - T[1], p[2] - T[2]};
  for (int c = 0; c < 3; ++c) o[c] = R[c] * d[0] + R[3 + c] * d[1] + R[6 + c] * d[2];
}
inline void check(int rc, const char* what) { if (rc < 0) throw Error(rc, what); }
}  // namespace detail

// Customisation point for the read-back methods that hand poses to the CALLER's types (getCurrPose, updateFactorGraphMap, ...):
// slide_assign_pose(dst, pose7) is found by argument-dependent lookup; the overload for slide::Pose7 is below, a maintainer adds
//     inline void slide_assign_pose(Sophus::SE3d& d, const double p[7]) { d = Sophus::SE3d(Eigen::Quaterniond(p[6], p[3], p[4], p[5]), {p[0], p[1], p[2]}); }
// next to the type.
inline void slide_assign_pose(Pose7& dst, const double p[7]) { for (int i = 0; i < 7; ++i) dst.v[i] = p[i]; }

// gtsam_cylinder::CylinderMeasurement / gtsam_cube::CubeMeasurement as the reference's call sites fill them
// (cylinderFactor.h:22-40, cubeFactor.h:25-44): plain aggregates here.
struct CylinderMeasurement { double root[3], ray[3], radius; };
struct CubeMeasurement { Pose7 pose; double scale[3]; };

// ---- S1 ----------------------------------------------------------------------------------------------------------------------
class SemanticFactorGraph {
 public:
  explicit SemanticFactorGraph(const slide_params_t* p = nullptr) : g_(slide_graph_create(p)), own_(true) {
    if (!g_) throw Error(SLIDE_ERR_HIP, "slide_graph_create");
  }
  explicit SemanticFactorGraph(slide_graph_t* borrowed) : g_(borrowed), own_(false) {}
  SemanticFactorGraph(const SemanticFactorGraph&) = delete;             // (the reference copy-assigns the wrapper and shares the raw ISAM2*, sloamNode.cpp:150: not reproduced)
  SemanticFactorGraph& operator=(const SemanticFactorGraph&) = delete;
  ~SemanticFactorGraph() { if (own_ && g_) slide_graph_destroy(g_); }

  template <class Pose>
  void setPriors(const Pose& pose_prior, const int& robotID) {                                          // graph.cpp:24-42
    double p[7]; detail::to7(pose_prior, p);
    detail::check(slide_graph_set_prior(g_, robotID, p), "setPriors");
  }
  template <class Pose>
  void addKeyPoseAndBetween(const size_t fromIdx, const size_t toIdx, const Pose& relativeMotion, const Pose& poseEstimate,
                            const int& robotID) {                                                      // graph.cpp:44-151
    double r[7], e[7]; detail::to7(relativeMotion, r); detail::to7(poseEstimate, e);
    detail::check(slide_graph_add_keypose_between(g_, robotID, fromIdx, toIdx, r, e), "addKeyPoseAndBetween");
  }
  template <class Pose>
  void addCylinderFactor(const size_t poseIdx, const size_t cylIdx, const Pose& pose, const CylinderMeasurement& cylinder,
                         bool alreadyExists, const int& robotID) {                                     // graph.cpp:182-196
    double p[7]; detail::to7(pose, p);
    detail::check(slide_graph_add_cylinder(g_, robotID, poseIdx, cylIdx, p, cylinder.root, cylinder.ray, cylinder.radius, alreadyExists ? 1 : 0),
                  "addCylinderFactor");
  }
  template <class Pose>
  void addCubeFactor(const size_t poseIdx, const size_t cubeIdx, const Pose& pose, const CubeMeasurement& cube_global_meas,
                     bool alreadyExists, const int& robotID) {                                         // graph.cpp:198-231
    double p[7], c[7]; detail::to7(pose, p); detail::to7(cube_global_meas.pose, c);
    detail::check(slide_graph_add_cube(g_, robotID, poseIdx, cubeIdx, p, c, cube_global_meas.scale, alreadyExists ? 1 : 0), "addCubeFactor");
  }
  template <class Point>
  void addPointLandmarkKey(const size_t ugvIdx, const Point& landmark_position) {                       // graph.cpp:153-156
    double x[3]; detail::to3(landmark_position, x);
    detail::check(slide_graph_add_point_landmark(g_, ugvIdx, x), "addPointLandmarkKey");
  }
  template <class Point>
  void addRangeBearingFactor(const size_t poseIdx, const size_t ugvIdx, const Point& bearing_measurement, const double& range_measurement,
                             const int& robotID) {                                                     // graph.cpp:158-180
    double b[3]; detail::to3(bearing_measurement, b);
    detail::check(slide_graph_add_range_bearing(g_, robotID, poseIdx, ugvIdx, b, range_measurement), "addRangeBearingFactor");
  }
  template <class Pose>
  void addLoopClosureFactor(const Pose& poseRelative, const size_t fromIdx, const size_t fromRobot, const size_t toIdx, const size_t toRobot) {
    double r[7]; detail::to7(poseRelative, r);                                                          // graph.cpp:233-245
    detail::check(slide_graph_add_loop_closure(g_, r, fromIdx, (int)fromRobot, toIdx, (int)toRobot), "addLoopClosureFactor");
  }
  template <class Pose>
  void addRelativeMeasFactor(const Pose& poseRelative, const size_t fromIdx, const size_t fromRobot, const size_t toIdx, const size_t toRobot) {
    double r[7]; detail::to7(poseRelative, r);                                                          // graph.cpp:247-258
    detail::check(slide_graph_add_relative_meas(g_, r, fromIdx, (int)fromRobot, toIdx, (int)toRobot), "addRelativeMeasFactor");
  }
  void solve() { detail::check(slide_graph_solve(g_), "solve"); }                                       // graph.cpp:260-272

  // getPose graph.cpp:290-312: false + identity when the key is absent
  bool getPose(const size_t idx, const int& robotID, Pose7& out) const {
    const int rc = slide_graph_get_pose(g_, robotID, idx, out.v);
    detail::check(rc, "getPose");
    return rc == SLIDE_OK;
  }
  CylinderMeasurement getCylinder(const int idx) const {                                                // graph.cpp:274-276 (Values::at throws)
    double o[15];
    const int rc = slide_graph_get_landmark(g_, SLIDE_CLS_CYLINDER, (uint64_t)idx, o);
    detail::check(rc, "getCylinder");
    if (rc == SLIDE_MISSING) throw std::out_of_range("getCylinder: key not in the graph");
    return CylinderMeasurement{{o[0], o[1], o[2]}, {o[3], o[4], o[5]}, o[6]};
  }
  // getCube graph.cpp:278-280: out15 = R row-major (9), t (3), scale (3)
  std::array<double, 15> getCube(const int idx) const {
    std::array<double, 15> o{};
    const int rc = slide_graph_get_landmark(g_, SLIDE_CLS_CUBE, (uint64_t)idx, o.data());
    detail::check(rc, "getCube");
    if (rc == SLIDE_MISSING) throw std::out_of_range("getCube: key not in the graph");
    return o;
  }
  std::array<double, 3> getCentroidLandmark(const int idx) const {                                      // graph.cpp:282-288 (zero point when absent)
    double o[15] = {0};
    detail::check(slide_graph_get_landmark(g_, SLIDE_CLS_ELLIPSOID, (uint64_t)idx, o), "getCentroidLandmark");
    return {o[0], o[1], o[2]};
  }
  std::array<double, 36> getPoseCovariance(const int idx, const int& robotID) const {                   // graph.cpp:314-323, row-major 6x6 [rot, trans]
    std::array<double, 36> c{};
    detail::check(slide_graph_get_pose_covariance(g_, robotID, (uint64_t)idx, c.data()), "getPoseCovariance");
    return c;
  }
  slide_graph_t* handle() const { return g_; }

 protected:
  slide_graph_t* g_;
  bool own_;
};

// ---- S2 ----------------------------------------------------------------------------------------------------------------------
// addSLOAMObservation (graphWrapper.cpp:99-237) consumes map managers' match tables; on the MI355X the whole per-key-frame body
// (submap gate -> match -> updateMap -> addSLOAMObservation -> solve -> updateFactorGraphMap) is one call, so the wrapper owns a
// slide_backend_t and exposes the S2 read-back methods over it.
class SemanticFactorGraphWrapper : public SemanticFactorGraph {
 public:
  explicit SemanticFactorGraphWrapper(const slide_params_t* p = nullptr) : SemanticFactorGraphWrapper(make(p)) {}
  ~SemanticFactorGraphWrapper() { if (b_) slide_backend_destroy(b_); }

  // runSLOAMNode body for one key frame (sloamNode.cpp:819-1014 incl. addSLOAMObservation :889); returns "optimized"
  template <class Pose>
  bool addSLOAMObservation(const slide_detections_t& detections, const Pose& relativeMotion, const Pose& prevKeyPose, const int& robotID,
                           Pose7* outPose = nullptr, slide_frame_result_t* matches = nullptr, int mode = SLIDE_FRAME_HOST) {
    double r[7], p[7]; detail::to7(relativeMotion, r); detail::to7(prevKeyPose, p);
    slide_frame_result_t local{};
    slide_frame_result_t* res = matches ? matches : &local;
    detail::check(slide_backend_process_frame(b_, mode, robotID, r, p, &detections, res), "addSLOAMObservation");
    if (outPose) for (int i = 0; i < 7; ++i) outPose->v[i] = res->out_pose7[i];
    return res->optimized != 0;
  }
  // getCurrPose graphWrapper.cpp:277-297
  bool getCurrPose(Pose7& curr_pose, const int& robotID) const {
    const size_t n = getPoseCounterById(robotID);
    return n > 0 && getPose(n - 1, robotID, curr_pose);
  }
  // getAllPoses graphWrapper.cpp:313-338
  bool getAllPoses(std::vector<Pose7>& optimized_poses, std::vector<size_t>& pose_inds, const int& robotID) const {
    const size_t n = getPoseCounterById(robotID);
    std::vector<double> flat(7 * (n ? n : 1));
    uint64_t got = 0;
    detail::check(slide_graph_get_all_poses(g_, robotID, flat.data(), n, &got), "getAllPoses");
    optimized_poses.resize(got);
    pose_inds.resize(got);
    for (uint64_t i = 0; i < got; ++i) { for (int k = 0; k < 7; ++k) optimized_poses[i].v[k] = flat[7 * i + k]; pose_inds[i] = i; }
    return got > 0;
  }
  size_t getPoseCounterById(const int& robotID) const {                                                // graphWrapper.h:127
    uint64_t out4[4], pc[SLIDE_MAX_ROBOTS];
    detail::check(slide_backend_counts(b_, out4, pc, SLIDE_MAX_ROBOTS), "getPoseCounterById");
    return robotID >= 0 && robotID < SLIDE_MAX_ROBOTS ? (size_t)pc[robotID] : 0;
  }
  slide_backend_t* backend() const { return b_; }

  // ---- the reference's own S2 signatures (graphWrapper.h:82-122) over the S1 entry points --------------------------------------
  // For a caller that keeps the reference's map managers and association (RunSloam, *MapManager::updateMap) on the host and hands
  // the match vectors over, exactly as sloamNode.cpp:889 / :993 do.  The types are template parameters that only have to provide
  // what graphWrapper.cpp:99-237 uses: map.getMatchesMap().at(int) -> int, map.getRawMap()[i].model, cylinder.model.{root, ray,
  // radius}, cube.model.{pose, scale}, ellipsoid.model.{pose, scale, semantic_label}.  These methods keep the reference's counters
  // (graphWrapper.h:128-134) themselves; do not mix them with the whole-frame addSLOAMObservation above on one object.
  std::vector<size_t> pose_counter_robot_ = std::vector<size_t>(SLIDE_MAX_ROBOTS, 0);                    // graphWrapper.h:128 (public there too)

  template <class CylMap, class CubeMap, class EllMap, class Cyl, class Cube, class Ell, class SE3>
  bool addSLOAMObservation(const CylMap& semanticMap, const CubeMap& cubeSemanticMap, const EllMap& ellipsoidSemanticMap,
                           const std::vector<int>& cyl_matches, const std::vector<Cyl>& cylinders, const std::vector<int>& cube_matches,
                           const std::vector<Cube>& cubes, const std::vector<int>& ellipsoid_matches, const std::vector<Ell>& ellipsoids,
                           const SE3& relativeMotion, const SE3& poseEstimate, const int& robotID, bool opt = true) {   // graphWrapper.cpp:99-237
    if (robotID < 0 || robotID >= SLIDE_MAX_ROBOTS) throw Error(SLIDE_ERR_INVALID, "addSLOAMObservation: robotID");
    Pose7 curr; detail::to7(poseEstimate, curr.v);
    const size_t pose_counter = pose_counter_robot_[robotID];
    if (pose_counter == 0) setPriors(poseEstimate, robotID);                                             // :114-119
    else addKeyPoseAndBetween(pose_counter - 1, pose_counter, relativeMotion, poseEstimate, robotID);    // :121
    const auto matchesMap = semanticMap.getMatchesMap();
    const auto cubeMatchesMap = cubeSemanticMap.getMatchesMap();
    for (size_t i = 0; i < cyl_matches.size(); ++i) {                                                    // :127-140
      CylinderMeasurement m;
      detail::to3(cylinders[i].model.root, m.root); detail::to3(cylinders[i].model.ray, m.ray); m.radius = cylinders[i].model.radius;
      if (cyl_matches[i] == -1) { addCylinderFactor(pose_counter, cyl_counter_, curr, m, false, robotID); ++cyl_counter_; }
      else addCylinderFactor(pose_counter, (size_t)matchesMap.at(cyl_matches[i]), curr, m, true, robotID);
    }
    for (size_t i = 0; i < cube_matches.size(); ++i) {                                                   // :143-156
      CubeMeasurement m;
      detail::to7(cubes[i].model.pose, m.pose.v); detail::to3(cubes[i].model.scale, m.scale);
      if (cube_matches[i] == -1) { addCubeFactor(pose_counter, cube_counter_, curr, m, false, robotID); ++cube_counter_; }
      else addCubeFactor(pose_counter, (size_t)cubeMatchesMap.at(cube_matches[i]), curr, m, true, robotID);
    }
    const auto ellipMatchesMap = ellipsoidSemanticMap.getMatchesMap();
    for (size_t i = 0; i < ellipsoid_matches.size(); ++i) {                                              // :159-203: body-frame bearing + range
      double w[3], b[3];
      detail::to3(ellipsoids[i].model.pose.translation(), w);
      detail::to_body(curr.v, w, b);
      const double range = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
      // (an ellipsoid AT the sensor: Eigen's .normalized() of the reference, graphWrapper.cpp:176-184, leaves the zero vector as it is —
      // dividing would hand NaNs to the solve)
      const double inv = range > 0.0 ? 1.0 / range : 0.0;
      const double bearing[3] = {b[0] * inv, b[1] * inv, b[2] * inv};
      if (ellipsoid_matches[i] == -1) {
        addPointLandmarkKey(point_landmark_counter_, w);
        addRangeBearingFactor(pose_counter, point_landmark_counter_, bearing, range, robotID);
        point_landmark_labels_.push_back(ellipsoids[i].model.semantic_label);
        ++point_landmark_counter_;
      } else {
        addRangeBearingFactor(pose_counter, (size_t)ellipMatchesMap.at(ellipsoid_matches[i]), bearing, range, robotID);
      }
    }
    pose_counter_robot_[robotID] = pose_counter + 1;                                                     // :206-207
    if (opt) { solve(); return true; }                                                                   // :212-230
    return false;
  }
  // updateFactorGraphMap graphWrapper.cpp:259-275 (+ updateCylinder / updateCube / updateEllipsoid :239-256): every optimised landmark
  // back into the caller's map models; an ellipsoid's pose becomes identity rotation + the optimised point (:252-255)
  template <class CylMap, class CubeMap, class EllMap>
  void updateFactorGraphMap(CylMap& semanticMap, CubeMap& cubeSemanticMap, EllMap& ellipsoidSemanticMap) {
    auto& map = semanticMap.getRawMap();
    auto& cube_map = cubeSemanticMap.getRawMap();
    auto& ellipsoid_map = ellipsoidSemanticMap.getRawMap();
    for (size_t i = 0; i < cyl_counter_; ++i) {
      const CylinderMeasurement m = getCylinder((int)i);
      for (int k = 0; k < 3; ++k) { map[i].model.root[k] = m.root[k]; map[i].model.ray[k] = m.ray[k]; }
      map[i].model.radius = m.radius;
    }
    for (size_t i = 0; i < cube_counter_; ++i) {
      const std::array<double, 15> c = getCube((int)i);
      double p7[7];
      rt_to7(c.data(), c.data() + 9, p7);
      slide_assign_pose(cube_map[i].model.pose, p7);
      for (int k = 0; k < 3; ++k) cube_map[i].model.scale[k] = c[12 + k];
    }
    for (size_t i = 0; i < point_landmark_counter_; ++i) {
      const std::array<double, 3> x = getCentroidLandmark((int)i);
      const double p7[7] = {x[0], x[1], x[2], 0, 0, 0, 1};
      slide_assign_pose(ellipsoid_map[i].model.pose, p7);
    }
  }
  // getCurrPose graphWrapper.cpp:277-297 with the reference's argument list; cov (optional, boost::optional<MatrixXd&> there) = the
  // 6x6 marginal covariance of that pose, row-major [rot, trans]
  template <class SE3>
  void getCurrPose(SE3& curr_pose, const int& robotID, std::array<double, 36>* cov) {
    const size_t n = robotID >= 0 && robotID < SLIDE_MAX_ROBOTS ? pose_counter_robot_[robotID] : 0;
    Pose7 p;
    if (n > 0) (void)getPose(n - 1, robotID, p);                  // (absent: identity, as the reference after its ROS_ERROR)
    slide_assign_pose(curr_pose, p.v);
    if (cov && n > 0) *cov = getPoseCovariance((int)(n - 1), robotID);
  }
  // getAllCentroidLandmarks / ...AndLabels graphWrapper.cpp:340-400: a landmark whose optimised point is exactly (0, 0, 0) counts as
  // missing (:345-347)
  template <class SE3>
  void getAllCentroidLandmarks(std::vector<SE3>& optimized_landmark_pos, std::vector<size_t>& landmark_inds) {
    for (size_t i = 0; i < point_landmark_counter_; ++i) {
      const std::array<double, 3> x = getCentroidLandmark((int)i);
      if (x[0] == 0.0 && x[1] == 0.0 && x[2] == 0.0) continue;
      const double p7[7] = {x[0], x[1], x[2], 0, 0, 0, 1};
      SE3 pose; slide_assign_pose(pose, p7);
      optimized_landmark_pos.push_back(pose);
      landmark_inds.push_back(i);
    }
  }
  template <class SE3>
  void getAllCentroidLandmarksAndLabels(std::vector<SE3>& optimized_landmark_pos, std::vector<int>& landmark_labels) {
    for (size_t i = 0; i < point_landmark_counter_; ++i) {
      const std::array<double, 3> x = getCentroidLandmark((int)i);
      if (x[0] == 0.0 && x[1] == 0.0 && x[2] == 0.0) continue;
      const double p7[7] = {x[0], x[1], x[2], 0, 0, 0, 1};
      SE3 pose; slide_assign_pose(pose, p7);
      optimized_landmark_pos.push_back(pose);
      landmark_labels.push_back(i < point_landmark_labels_.size() ? point_landmark_labels_[i] : -1);     // (:392-398)
    }
  }

 private:
  size_t cyl_counter_ = 0, cube_counter_ = 0, point_landmark_counter_ = 0;                                // graphWrapper.h:131-133
  std::vector<int> point_landmark_labels_;
  static void rt_to7(const double R[9], const double t[3], double o[7]) {                                  // row-major R -> unit quaternion
    const double tr = R[0] + R[4] + R[8];
    double w, x, y, z;
    if (tr > 0) { const double s = std::sqrt(tr + 1.0) * 2; w = 0.25 * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s; }
    else if (R[0] > R[4] && R[0] > R[8]) { const double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2; w = (R[7] - R[5]) / s; x = 0.25 * s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s; }
    else if (R[4] > R[8]) { const double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2; w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = 0.25 * s; z = (R[5] + R[7]) / s; }
    else { const double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2; w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = 0.25 * s; }
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = x; o[4] = y; o[5] = z; o[6] = w;
  }
  static slide_backend_t* make(const slide_params_t* p) {
    slide_backend_t* b = slide_backend_create(p);
    if (!b) throw Error(SLIDE_ERR_HIP, "slide_backend_create");
    return b;
  }
  explicit SemanticFactorGraphWrapper(slide_backend_t* b) : SemanticFactorGraph(slide_backend_graph(b)), b_(b) {}
  slide_backend_t* b_ = nullptr;
};

}  // namespace slide
#endif  // SLIDE_SLOAM_ADAPTOR_HPP_
