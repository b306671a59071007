// Compile-and-link check of include/slide_sloam_adaptor.hpp against libslide_gpu.so (tests/test_abi.py): exercises every method of the
// S1 / S2 adaptor classes the way graphWrapper.cpp / sloamNode.cpp call the reference classes.  Run on a GPU box it also executes.
#include <cstdio>
#include <vector>

#include "slide_sloam_adaptor.hpp"

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
    return ok && missing && threw && poses.size() == 1 ? 0 : 1;
  } catch (const slide::Error& e) {
    std::printf("slide::Error %d: %s\n", e.code, e.what());
    return 2;
  }
}
