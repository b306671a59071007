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
inline void check(int rc, const char* what) { if (rc < 0) throw Error(rc, what); }
}  // namespace detail

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

 private:
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
