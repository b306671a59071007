// Per-key-frame update on one MI355X: the body of SLOAMNode::runSLOAMNode
// (backend/sloam/src/core/sloamNode.cpp:762-1036) without ROS: submap gate + projectModels + match
// (one kernel launch, one workgroup per landmark class), updateMap (append to the HBM-resident maps),
// SemanticFactorGraphWrapper::addSLOAMObservation (graphWrapper.cpp:99-237), solve, updateFactorGraphMap.
#include "host_backend.hpp"

#include <chrono>
#include <cmath>

namespace sl {

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

HostBackend::HostBackend(const slide_params_t& p) : g(p), P(p) {
  maps[0].cls = SLIDE_CLS_CYLINDER; maps[0].K = p.knn_cylinder; maps[0].stride = 7;
  maps[1].cls = SLIDE_CLS_CUBE; maps[1].K = p.knn_cube; maps[1].stride = 3;
  maps[2].cls = SLIDE_CLS_ELLIPSOID; maps[2].K = p.knn_ellipsoid; maps[2].stride = 3;
  for (int i = 0; i < SLIDE_MAX_ROBOTS; ++i) pose_counter[i] = 0;
}

int HostBackend::init() {
  const int rc = g.init();
  if (rc != SLIDE_OK) return rc;
  if (d_pose12.ensure(12, 0, g.stream) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_cls3.ensure(3, 0, g.stream) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_nsub.ensure(4, 0, g.stream, true) != SLIDE_OK) return SLIDE_ERR_HIP;
  return SLIDE_OK;
}

// getSubmap + projectModels + match*Models for the three classes of one key frame
int HostBackend::associate(const SE3& poseEstimate, const slide_detections_t& det, bool first_scan_shortcut, FrameAssoc& A) {
  hipStream_t s = g.stream;
  const int nd[3] = {det.n_cyl, det.n_cube, det.n_ell};
  const int dstride[3] = {7, 12, 12};
  const int32_t* labels[3] = {det.cyl_label, det.cube_label, det.ell_label};
  // pack body-frame detections: cylinders [root ray radius], boxes pose12
  for (int c = 0; c < 3; ++c) {
    A.det_body[c].resize((size_t)nd[c] * dstride[c]);
    A.det_world[c].resize((size_t)nd[c] * dstride[c]);
    A.match_sub[c].assign(nd[c], -1);
    A.match_map[c].assign(nd[c], -1);
  }
  for (int i = 0; i < det.n_cyl; ++i) {
    double* o = &A.det_body[0][7 * (size_t)i];
    for (int k = 0; k < 3; ++k) { o[k] = det.cyl_root[3 * i + k]; o[3 + k] = det.cyl_ray[3 * i + k]; }
    o[6] = det.cyl_radius[i];
  }
  for (int i = 0; i < det.n_cube; ++i) to12(from7(det.cube_pose7 + 7 * (size_t)i), &A.det_body[1][12 * (size_t)i]);
  for (int i = 0; i < det.n_ell; ++i) to12(from7(det.ell_pose7 + 7 * (size_t)i), &A.det_body[2][12 * (size_t)i]);

  AssocFrameDev F[3];
  if (ub.begin() != SLIDE_OK) return SLIDE_ERR_HIP;
  struct BatchGuard { ~BatchGuard() { UploadBatch::current = nullptr; } } batch_guard;      // an error return abandons the batch
  size_t lds_bytes = 0;
  for (int c = 0; c < 3; ++c) {
    ClassMap& M = maps[c];
    size_t lb = 0;
    if (!assoc_plan(M.n(), M.K, 1, M.stride, &F[c].Kp, &F[c].cached, &F[c].staged, &lb)) {
      g_last_error = "K-NN gate: K exceeds the on-chip sort buffer (16384 neighbours)";
      return SLIDE_ERR_CAPACITY;
    }
    lds_bytes = std::max(lds_bytes, lb);
    F[c].gate = 1;
    const size_t n1 = std::max<size_t>(nd[c], 1);
    if (d_det[c].ensure(n1 * dstride[c], 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_det_world[c].ensure(n1 * dstride[c], 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_det_label[c].ensure(n1, 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_match_sub[c].ensure(n1, 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_match_map[c].ensure(n1, 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_submap[c].ensure(std::max(M.K, 1), 0, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_det[c].upload(A.det_body[c].data(), 0, A.det_body[c].size(), s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (nd[c] && d_det_label[c].upload(labels[c], 0, nd[c], s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (M.sync_device(s) != SLIDE_OK) return SLIDE_ERR_HIP;
    F[c].cx = M.d_cx.d; F[c].cy = M.d_cy.d; F[c].cz = M.d_cz.d; F[c].model = M.d_model.d; F[c].label = M.d_label.d; F[c].n = M.n(); F[c].K = M.K;
    F[c].is_cyl = (c == 0);
    // thresholds / initial bestDist of sloam.cpp:90,104 / :128-136,150 / :174-180,198
    if (c == 0) { F[c].thresh = P.cylinder_match_thresh; F[c].best_init = P.cylinder_match_thresh + 100; F[c].label_gate = 2; }
    else if (c == 1) { F[c].thresh = P.cuboid_match_thresh; F[c].best_init = 30; F[c].label_gate = 0; }
    else { F[c].thresh = P.ellipsoid_match_thresh; F[c].best_init = 1000; F[c].label_gate = 1; }
    F[c].det = d_det[c].d; F[c].det_label = d_det_label[c].d; F[c].n_det = nd[c];
    F[c].det_world = d_det_world[c].d; F[c].match_sub = d_match_sub[c].d; F[c].match_map = d_match_map[c].d;
    // (the nearest-first submap list itself is not asked for: the matches' positions in it are counted on the device, assoc_core rank_mode)
    F[c].submap = nullptr; F[c].n_sub = nullptr;
  }
  double pose12[12];
  to12(poseEstimate, pose12);
  if (d_pose12.upload(pose12, 0, 12, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_cls3.upload(F, 0, 3, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (ub.flush(s) != SLIDE_OK) return SLIDE_ERR_HIP;
  launch_assoc_frame(d_cls3.d, lds_bytes, d_pose12.d, s);
  for (int c = 0; c < 3; ++c) {
    if (!nd[c]) continue;
    db.add(A.det_world[c].data(), d_det_world[c].d, A.det_world[c].size() * sizeof(double));
    db.add(A.match_sub[c].data(), d_match_sub[c].d, nd[c] * sizeof(int));
    db.add(A.match_map[c].data(), d_match_map[c].d, nd[c] * sizeof(int));
  }
  if (db.run(s) != SLIDE_OK) return SLIDE_ERR_HIP;
  SL_HIP(hipGetLastError());
  if (first_scan_shortcut && firstScan) {   // sloam.cpp:235-248: the first scan is never matched
    firstScan = false;
    for (int c = 0; c < 3; ++c) { A.match_sub[c].assign(nd[c], -1); A.match_map[c].assign(nd[c], -1); }
  }
  // updateMap (cylinderMapManager.cpp:35-68, cubeMapManager.cpp:104-130, ellipsoidMapManager.cpp:111-145)
  for (int c = 0; c < 3; ++c) {
    ClassMap& M = maps[c];
    const double* scales = c == 1 ? det.cube_scale : det.ell_scale;
    for (int i = 0; i < nd[c]; ++i) {
      const double* w = &A.det_world[c][(size_t)i * dstride[c]];
      if (A.match_sub[c][i] == -1) {
        const double* pos = c == 0 ? w : w + 9;
        M.h_cx.push_back((float)pos[0]); M.h_cy.push_back((float)pos[1]); M.h_cz.push_back((float)pos[2]);
        if (c == 0) M.h_model.insert(M.h_model.end(), w, w + 7);
        else {
          M.h_model.insert(M.h_model.end(), pos, pos + 3);
          M.scale.insert(M.scale.end(), scales + 3 * (size_t)i, scales + 3 * (size_t)i + 3);
        }
        M.h_label.push_back(labels[c][i]);
        M.hits.push_back(1);
      } else {
        const int mi = A.match_map[c][i];
        M.hits[mi] += 1;
        if (c == 2) {   // ellipsoid scale moving average, alpha = 0.2
          for (int k = 0; k < 3; ++k) M.scale[3 * (size_t)mi + k] = (1. - 0.2) * M.scale[3 * (size_t)mi + k] + 0.2 * scales[3 * (size_t)i + k];
        }
      }
    }
  }
  return SLIDE_OK;
}

// graphWrapper.cpp:99-237 (detections in the world frame)
int HostBackend::add_observation(const FrameAssoc& A, const slide_detections_t& det, const SE3& rel, const SE3& pose, int robot,
                                 bool opt, slide_frame_result_t* res, bool* optimized) {
  const uint64_t pc = pose_counter[robot];
  double p7[7], r7[7];
  to7(pose, p7);
  to7(rel, r7);
  if (pc == 0) g.set_prior(robot, p7);
  else g.add_keypose_between(robot, pc - 1, pc, r7, p7);
  // NB the graph stores the pose it was GIVEN (Pose3 from the same matrix), so re-derive it from pose7
  const SE3 gpose = from7(p7);
  (void)gpose;
  for (int i = 0; i < det.n_cyl; ++i) {
    const double* w = &A.det_world[0][7 * (size_t)i];
    uint64_t id;
    const bool isnew = A.match_sub[0][i] == -1;
    id = isnew ? cyl_counter++ : (uint64_t)A.match_map[0][i];
    g.add_cylinder(robot, pc, id, pose, w, w + 3, w[6], !isnew);
    if (res && res->cyl_id) res->cyl_id[i] = (int32_t)id;
  }
  for (int i = 0; i < det.n_cube; ++i) {
    const SE3 cw = from12(&A.det_world[1][12 * (size_t)i]);
    const bool isnew = A.match_sub[1][i] == -1;
    const uint64_t id = isnew ? cube_counter++ : (uint64_t)A.match_map[1][i];
    g.add_cube(robot, pc, id, pose, cw, det.cube_scale + 3 * (size_t)i, !isnew);
    if (res && res->cube_id) res->cube_id[i] = (int32_t)id;
  }
  for (int i = 0; i < det.n_ell; ++i) {
    const double* w = &A.det_world[2][12 * (size_t)i];
    const V3 pw{w[9], w[10], w[11]};
    const V3 body = transform_to(pose, pw);            // (pose^-1 * ellipsoid_world).translation()
    const double range = norm(body);
    const double bearing[3] = {body.x / range, body.y / range, body.z / range};
    const bool isnew = A.match_sub[2][i] == -1;
    uint64_t id;
    if (isnew) {
      id = point_counter++;
      const double xyz[3] = {pw.x, pw.y, pw.z};
      g.add_point_landmark(id, xyz);
      point_labels.push_back(det.ell_label[i]);
    } else {
      id = (uint64_t)A.match_map[2][i];
    }
    g.add_range_bearing(robot, pc, id, bearing, range);
    if (res && res->ell_id) res->ell_id[i] = (int32_t)id;
  }
  pose_counter[robot] = pc + 1;
  *optimized = false;
  if (opt) {
    const int rc = g.solve();
    *optimized = true;
    return rc;
  }
  return SLIDE_OK;
}

int ClassMap::sync_device(hipStream_t s) {
  const size_t nn = h_label.size();
  if (d_cx.ensure(std::max<size_t>(nn, 1), up_n, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_cy.ensure(std::max<size_t>(nn, 1), up_n, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_cz.ensure(std::max<size_t>(nn, 1), up_n, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_model.ensure(std::max<size_t>(stride * nn, 1), stride * up_n, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (d_label.ensure(std::max<size_t>(nn, 1), up_n, s) != SLIDE_OK) return SLIDE_ERR_HIP;
  if (nn > up_n) {
    if (d_cx.upload(h_cx.data() + up_n, up_n, nn - up_n, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_cy.upload(h_cy.data() + up_n, up_n, nn - up_n, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_cz.upload(h_cz.data() + up_n, up_n, nn - up_n, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_model.upload(h_model.data() + stride * up_n, stride * up_n, stride * (nn - up_n), s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (d_label.upload(h_label.data() + up_n, up_n, nn - up_n, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    up_n = nn;
  }
  return SLIDE_OK;
}

// updateFactorGraphMap (graphWrapper.cpp:259-275): every optimised landmark back into the map models
int HostBackend::refresh_maps() {
  hipStream_t s = g.stream;
  const uint64_t counters[3] = {cyl_counter, cube_counter, point_counter};
  // (the new landmarks' cloud points, models, labels and graph ids of all three classes travel as ONE copy: up to 18 small ones otherwise)
  if (ub.begin() != SLIDE_OK) return SLIDE_ERR_HIP;
  struct BatchGuard { ~BatchGuard() { UploadBatch::current = nullptr; } } batch_guard;
  for (int c = 0; c < 3; ++c) {
    ClassMap& M = maps[c];
    if (M.sync_device(s) != SLIDE_OK) return SLIDE_ERR_HIP;
    while (M.lid.size() < counters[c]) {
      const int lid = g.lm_lid(M.cls, M.lid.size());
      if (lid < 0) { g_last_error = "landmark missing from the graph during map refresh"; return SLIDE_ERR_INVALID; }
      M.lid.push_back(lid);
    }
    if (M.d_lid.ensure(std::max<size_t>(M.lid.size(), 1), M.up_lid, s) != SLIDE_OK) return SLIDE_ERR_HIP;
    if (M.lid.size() > M.up_lid) {
      if (M.d_lid.upload(M.lid.data() + M.up_lid, M.up_lid, M.lid.size() - M.up_lid, s) != SLIDE_OK) return SLIDE_ERR_HIP;
      M.up_lid = M.lid.size();
    }
  }
  if (ub.flush(s) != SLIDE_OK) return SLIDE_ERR_HIP;
  launch_map_refresh(maps[0].d_model.d, (int)maps[0].lid.size(), maps[0].d_lid.d, maps[1].d_model.d, (int)maps[1].lid.size(),
                     maps[1].d_lid.d, maps[2].d_model.d, (int)maps[2].lid.size(), maps[2].d_lid.d, g.G.lm_est, s);
  SL_HIP(hipGetLastError());
  return SLIDE_OK;
}

int HostBackend::process_frame(int mode, int robot, const double* rel7, const double* prev7, const slide_detections_t& det,
                               slide_frame_result_t* res) {
  if (robot < 0 || robot >= SLIDE_MAX_ROBOTS) return SLIDE_ERR_INVALID;
  const SE3 rel = from7(rel7);
  const SE3 prev = from7(prev7);
  const bool foreign = mode == SLIDE_FRAME_FOREIGN;
  const SE3 poseEstimate = foreign ? prev : compose(prev, rel);    // sloamNode.cpp:785 / :939-940
  FrameAssoc A;
  const double t0 = now_ms();
  int rc = associate(poseEstimate, det, !foreign, A);
  if (rc != SLIDE_OK) return rc;
  const double t1 = now_ms();
  bool optimized = false;
  rc = add_observation(A, det, rel, poseEstimate, robot, !foreign, res, &optimized);
  if (res) {
    for (int i = 0; i < det.n_cyl; ++i) if (res->cyl_match) res->cyl_match[i] = A.match_sub[0][i];
    for (int i = 0; i < det.n_cube; ++i) if (res->cube_match) res->cube_match[i] = A.match_sub[1][i];
    for (int i = 0; i < det.n_ell; ++i) if (res->ell_match) res->ell_match[i] = A.match_sub[2][i];
    res->optimized = optimized ? 1 : 0;
    to7(poseEstimate, res->out_pose7);
  }
  if (rc != SLIDE_OK) return rc;
  if (optimized && mode == SLIDE_FRAME_HOST) {
    rc = refresh_maps();
    if (rc != SLIDE_OK) return rc;
    double p12[12];
    const int st = g.get_pose12(robot, pose_counter[robot] - 1, p12);
    if (st == SLIDE_OK && res) to7(from12(p12), res->out_pose7);
  }
  if (res) {
    res->ms_association = t1 - t0;
    res->ms_graph = now_ms() - t1;
  }
  return SLIDE_OK;
}

int HostBackend::ingest_solve() { return g.solve(); }

int HostBackend::end_frame(int robot, double* out7) {
  int rc = refresh_maps();
  if (rc != SLIDE_OK) return rc;
  double p12[12];
  rc = g.get_pose12(robot, pose_counter[robot] - 1, p12);
  to7(from12(p12), out7);
  return rc;
}

int HostBackend::map_model(int cls, int idx, double* out, int* hits, int* label) {
  if (cls < 0 || cls > 2) return SLIDE_ERR_INVALID;
  ClassMap& M = maps[cls];
  if (idx < 0 || idx >= M.n()) return SLIDE_MISSING;
  if (M.sync_device(g.stream) != SLIDE_OK) return SLIDE_ERR_HIP;
  SL_HIP(hipMemcpyAsync(out, M.d_model.d + (size_t)M.stride * idx, M.stride * sizeof(double), hipMemcpyDeviceToHost, g.stream));
  SL_HIP(hipStreamSynchronize(g.stream));
  if (cls != 0) for (int k = 0; k < 3; ++k) out[3 + k] = M.scale[3 * (size_t)idx + k];
  if (cls == SLIDE_CLS_CUBE && (size_t)idx < M.up_lid) {
    // updateCube (graphWrapper.cpp:245-249) also copies the optimised scale into the map model
    double c15[15];
    if (g.get_landmark(SLIDE_CLS_CUBE, (uint64_t)idx, c15) == SLIDE_OK) for (int k = 0; k < 3; ++k) out[3 + k] = c15[12 + k];
  }
  *hits = M.hits[idx];
  *label = M.h_label[idx];
  return SLIDE_OK;
}

}  // namespace sl
