// Compile-and-link check of include/slide_sloam_adaptor.hpp against libslide_gpu.so (tests/test_abi.py): exercises every method of the
// S1 / S2 adaptor classes the way graphWrapper.cpp / sloamNode.cpp call the reference classes.  Run on a GPU box it also executes.
#include <cmath>
#include <cstdio>
#include <vector>

#include <map>

#include "slide_sloam_adaptor.hpp"

// stand-ins with the member names graphWrapper.cpp:99-275 uses on the reference's map managers and objects
namespace ref {
struct CylModel { double root[3], ray[3], radius; };
struct Cylinder { CylModel model; };
struct CubeModel { slide::Pose7 pose; double scale[3]; };
struct Cube { CubeModel model; };
struct EllModel { slide::Pose7 pose; double scale[3]; int semantic_label; };
struct Ellipsoid { EllModel model; };
template <class T>
struct MapManager {
  std::vector<T> raw;
  std::map<int, int> matches;
  std::map<int, int> getMatchesMap() const { return matches; }      // (returned by value in the reference too, graphWrapper.cpp:123-124)
  std::vector<T>& getRawMap() { return raw; }
};
}  // namespace ref

int main(int argc, char** argv) {
  if (argc < 2) return 0;      // link check only (no device needed)
  try {
    slide::SemanticFactorGraph g;
    slide::Pose7 a, b;
    b.v[0] = 1.0;
    g.setPriors(a, 0);
    g.addKeyPoseAndBetween(0, 1, b, b, 0);
    double xyz[3] = {2.0, 1.0, 0.0}, bearing[3] = {0.894427190999916, 0.447213595499958, 0.0};
    g.addPointLandmarkKey(0, xyz);
    g.addRangeBearingFactor(0, 0, bearing, 2.23606797749979, 0);
    double b2[3] = {0.707106781186548, 0.707106781186548, 0.0};
    g.addRangeBearingFactor(1, 0, b2, 1.4142135623731, 0);
    slide::CubeMeasurement cm; cm.pose.v[0] = 5.0; cm.scale[0] = cm.scale[1] = cm.scale[2] = 1.0;
    g.addCubeFactor(0, 0, a, cm, false, 0);
    slide::CylinderMeasurement cy{{3, 4, 0}, {0, 0, 1}, 0.3};
    g.addCylinderFactor(1, 0, b, cy, false, 0);
    g.addLoopClosureFactor(b, 0, 0, 1, 0);
    g.solve();
    slide::Pose7 out;
    const bool ok = g.getPose(1, 0, out), missing = !g.getPose(7, 0, out);
    const auto lm = g.getCentroidLandmark(0);
    const auto cube = g.getCube(0);
    const auto cyl = g.getCylinder(0);
    const auto cov = g.getPoseCovariance(1, 0);
    bool threw = false;
    try { g.getCube(9); } catch (const std::out_of_range&) { threw = true; }
    std::printf("adaptor ok=%d missing=%d threw=%d lm=%.3f cube_t=%.3f cyl_r=%.3f cov00=%.3e\n", ok, missing, threw, lm[0], cube[9], cyl.radius, cov[0]);
    slide::SemanticFactorGraphWrapper w;
    slide_detections_t det{};
    slide::Pose7 est;
    w.addSLOAMObservation(det, a, a, 0, &est);
    std::vector<slide::Pose7> poses; std::vector<size_t> idx;
    w.getAllPoses(poses, idx, 0);
    std::printf("wrapper poses=%zu counter=%zu\n", poses.size(), w.getPoseCounterById(0));
    // the reference's own S2 argument lists (graphWrapper.h:82-122): two key frames, one object of each class seen from both
    slide::SemanticFactorGraphWrapper w2;
    ref::MapManager<ref::Cylinder> cylMap; ref::MapManager<ref::Cube> cubeMap; ref::MapManager<ref::Ellipsoid> ellMap;
    ref::Cylinder cyl0{{{3, 4, 0}, {0, 0, 1}, 0.3}};
    ref::Cube cube0{}; cube0.model.pose.v[0] = 5.0; cube0.model.scale[0] = cube0.model.scale[1] = cube0.model.scale[2] = 1.0;
    ref::Ellipsoid ell0{}; ell0.model.pose.v[0] = 2.0; ell0.model.pose.v[1] = 1.0; ell0.model.semantic_label = 4;
    cylMap.raw.push_back(cyl0); cubeMap.raw.push_back(cube0); ellMap.raw.push_back(ell0);
    const std::vector<ref::Cylinder> cyls{cyl0}; const std::vector<ref::Cube> cubes{cube0}; const std::vector<ref::Ellipsoid> ells{ell0};
    const bool opt0 = w2.addSLOAMObservation(cylMap, cubeMap, ellMap, {-1}, cyls, {-1}, cubes, {-1}, ells, a, a, 0, false);
    cylMap.matches[0] = 0; cubeMap.matches[0] = 0; ellMap.matches[0] = 0;
    const bool opt1 = w2.addSLOAMObservation(cylMap, cubeMap, ellMap, {0}, cyls, {0}, cubes, {0}, ells, b, b, 0);
    w2.updateFactorGraphMap(cylMap, cubeMap, ellMap);
    slide::Pose7 cur; std::array<double, 36> cov2{};
    w2.getCurrPose(cur, 0, &cov2);
    std::vector<slide::Pose7> lms; std::vector<int> labels;
    w2.getAllCentroidLandmarksAndLabels(lms, labels);
    std::vector<size_t> linds; std::vector<slide::Pose7> lms2;
    w2.getAllCentroidLandmarks(lms2, linds);
    std::printf("S2 opt=%d/%d counter=%zu cur_x=%.3f cov00=%.3e cyl_r=%.3f cube_x=%.3f ell=(%.3f, %.3f) label=%d\n", opt0, opt1, w2.pose_counter_robot_[0],
                cur.v[0], cov2[0], cylMap.raw[0].model.radius, cubeMap.raw[0].model.pose.v[0], ellMap.raw[0].model.pose.v[0], ellMap.raw[0].model.pose.v[1],
                labels.empty() ? -9 : labels[0]);
    const bool s2 = !opt0 && opt1 && w2.pose_counter_robot_[0] == 2 && lms.size() == 1 && labels[0] == 4 && linds.size() == 1 &&
                    std::fabs(cur.v[0] - 1.0) < 1e-3 && std::fabs(ellMap.raw[0].model.pose.v[0] - 2.0) < 1e-3 && cov2[0] > 0.0;
    return ok && missing && threw && poses.size() == 1 && s2 ? 0 : 1;
  } catch (const slide::Error& e) {
    std::printf("slide::Error %d: %s\n", e.code, e.what());
    return 2;
  }
}
