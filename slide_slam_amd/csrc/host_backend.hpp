// Host side of the per-key-frame update (see host_backend.hip).
#pragma once
#include <vector>

#include "host_graph.hpp"

namespace sl {

// HBM-resident semantic map of one landmark class (the *MapManager state of the reference:
// models + hit counts + float32 cloud of first-seen positions, cubeMapManager.cpp:116-120).
struct ClassMap {
  int cls = 0, K = 0, stride = 3;
  std::vector<float> h_cx, h_cy, h_cz;     // first-seen positions, SoA (three coalesced streams for the K-NN scan)
  std::vector<double> h_model;   // values at insertion; the live models are refreshed in HBM
  std::vector<int> h_label, hits, lid;
  std::vector<double> scale;     // boxes only (ellipsoid EMA lives here)
  std::vector<int> matchesMap;   // submap index -> map index of the latest getSubmap
  DevArr<float> d_cx, d_cy, d_cz;
  DevArr<double> d_model;
  DevArr<int> d_label, d_lid;
  size_t up_n = 0, up_lid = 0;
  int n() const { return (int)h_label.size(); }
  int sync_device(hipStream_t s);
};

struct FrameAssoc {
  std::vector<double> det_body[3], det_world[3];
  std::vector<int> match_sub[3], match_map[3];
};

class HostBackend {
 public:
  explicit HostBackend(const slide_params_t& p);
  int init();
  int process_frame(int mode, int robot, const double* rel7, const double* prev7, const slide_detections_t& det,
                    slide_frame_result_t* res);
  int ingest_solve();
  int end_frame(int robot, double* out7);
  int map_model(int cls, int idx, double* out, int* hits, int* label);

  HostGraph g;
  slide_params_t P;
  ClassMap maps[3];
  bool firstScan = true;
  uint64_t cyl_counter = 0, cube_counter = 0, point_counter = 0;   // graphWrapper.h:131-134
  uint64_t pose_counter[SLIDE_MAX_ROBOTS];
  std::vector<int> point_labels;

 private:
  int associate(const SE3& poseEstimate, const slide_detections_t& det, bool first_scan_shortcut, FrameAssoc& A);
  int add_observation(const FrameAssoc& A, const slide_detections_t& det, const SE3& rel, const SE3& pose, int robot, bool opt,
                      slide_frame_result_t* res, bool* optimized);
  int refresh_maps();
  DevArr<double> d_det[3], d_det_world[3], d_pose12;
  DevArr<int> d_det_label[3], d_match_sub[3], d_match_map[3], d_submap[3], d_nsub, d_status;
  UploadBatch ub;          // the ~17 uploads and ~14 downloads of one association step as one copy each
  DownloadBatch db;
  DevArr<AssocFrameDev> d_cls3;
};

}  // namespace sl
