// Wire codec of the reference's sloam_msgs (ROS-1 serialisation) and a rosbag v2.0 reader — host code behind
// include/slide_wire.h.  Field order follows backend/sloam_msgs/msg/*.msg; the ROS-1 rules are:
// little-endian scalars, T[N] inline, T[] = uint32 count + elements, string = uint32 length + bytes, time = 2 x uint32.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/slide_gpu.h"
#include "../../include/slide_wire.h"

namespace sl {
extern thread_local std::string g_last_error;
}

struct slide_wire_arena {
  std::vector<void*> blocks;
  ~slide_wire_arena() { for (void* b : blocks) free(b); }
  template <class T>
  T* make(size_t n) {
    void* p = calloc(n ? n : 1, sizeof(T));
    if (!p) return nullptr;
    blocks.push_back(p);
    return static_cast<T*>(p);
  }
};

namespace {

// ---- writer: counts when buf == nullptr ------------------------------------------------------------------------------
struct W {
  uint8_t* buf;
  size_t cap, n = 0;
  bool overflow = false;
  void raw(const void* p, size_t k) {
    if (buf) {
      if (n + k <= cap) memcpy(buf + n, p, k); else overflow = true;
    }
    n += k;
  }
  // this library only runs on little-endian hosts (x86-64 beside gfx950): scalars go out as they lie in memory
  void u32(uint32_t v) { raw(&v, 4); }
  void i8(int8_t v) { raw(&v, 1); }
  void i64(int64_t v) { raw(&v, 8); }
  void f32(float v) { raw(&v, 4); }
  void f64(double v) { raw(&v, 8); }
  void str(const char* s, uint32_t len) { u32(len); if (len) raw(s, len); }
};

struct R {
  const uint8_t* buf;
  size_t len, n = 0;
  bool bad = false;
  bool need(size_t k) {
    if (bad || k > len - n) { bad = true; return false; }
    return true;
  }
  void raw(void* p, size_t k) { if (need(k)) { memcpy(p, buf + n, k); n += k; } else memset(p, 0, k); }
  uint32_t u32() { uint32_t v; raw(&v, 4); return v; }
  int8_t i8() { int8_t v; raw(&v, 1); return v; }
  int64_t i64() { int64_t v; raw(&v, 8); return v; }
  float f32() { float v; raw(&v, 4); return v; }
  double f64() { double v; raw(&v, 8); return v; }
  // element count of a T[] whose elements take at least `min_elem` bytes each: rejects counts the buffer cannot hold
  uint32_t count(size_t min_elem) {
    const uint32_t c = u32();
    if (!bad && (size_t)c * min_elem > len - n) bad = true;
    return bad ? 0 : c;
  }
};

void put_pose(W& w, const slide_wire_pose_t& p) { for (int i = 0; i < 3; ++i) w.f64(p.p[i]); for (int i = 0; i < 4; ++i) w.f64(p.q[i]); }
void get_pose(R& r, slide_wire_pose_t& p) { for (int i = 0; i < 3; ++i) p.p[i] = r.f64(); for (int i = 0; i < 4; ++i) p.q[i] = r.f64(); }

void put_box(W& w, const slide_wire_box_t& b) {            // ROSCube.msg / ROSEllipsoid.msg
  for (int i = 0; i < 3; ++i) w.f32(b.dim[i]);
  w.i8(b.semantic_label);
  put_pose(w, b.pose);
}
void get_box(R& r, slide_wire_box_t& b) {
  for (int i = 0; i < 3; ++i) b.dim[i] = r.f32();
  b.semantic_label = r.i8();
  get_pose(r, b.pose);
}
void put_cyl(W& w, const slide_wire_cylinder_t& c) {       // ROSCylinder.msg
  for (int i = 0; i < 3; ++i) w.f32(c.root[i]);
  for (int i = 0; i < 3; ++i) w.f32(c.ray[i]);
  w.u32(c.n_radii);
  for (uint32_t i = 0; i < c.n_radii; ++i) w.f64(c.radii[i]);
  w.f32(c.radius);
  w.i64(c.id);
  w.i8(c.semantic_label);
}
bool get_cyl(R& r, slide_wire_arena& A, slide_wire_cylinder_t& c) {
  for (int i = 0; i < 3; ++i) c.root[i] = r.f32();
  for (int i = 0; i < 3; ++i) c.ray[i] = r.f32();
  c.n_radii = r.count(8);
  double* rad = A.make<double>(c.n_radii);
  if (!rad) return false;
  for (uint32_t i = 0; i < c.n_radii; ++i) rad[i] = r.f64();
  c.radii = rad;
  c.radius = r.f32();
  c.id = r.i64();
  c.semantic_label = r.i8();
  return true;
}
void put_boxes(W& w, uint32_t n, const slide_wire_box_t* b) { w.u32(n); for (uint32_t i = 0; i < n; ++i) put_box(w, b[i]); }
void put_cyls(W& w, uint32_t n, const slide_wire_cylinder_t* c) { w.u32(n); for (uint32_t i = 0; i < n; ++i) put_cyl(w, c[i]); }
bool get_boxes(R& r, slide_wire_arena& A, uint32_t& n, const slide_wire_box_t*& out) {
  n = r.count(69);
  slide_wire_box_t* b = A.make<slide_wire_box_t>(n);
  if (!b) return false;
  for (uint32_t i = 0; i < n; ++i) get_box(r, b[i]);
  out = b;
  return true;
}
bool get_cyls(R& r, slide_wire_arena& A, uint32_t& n, const slide_wire_cylinder_t*& out) {
  n = r.count(41);
  slide_wire_cylinder_t* c = A.make<slide_wire_cylinder_t>(n);
  if (!c) return false;
  for (uint32_t i = 0; i < n; ++i)
    if (!get_cyl(r, A, c[i])) return false;
  out = c;
  return true;
}
void put_header(W& w, const slide_wire_header_t& h) { w.u32(h.seq); w.u32(h.stamp_sec); w.u32(h.stamp_nsec); w.str(h.frame_id, h.frame_id_len); }
bool get_str(R& r, slide_wire_arena& A, uint32_t& len, const char*& s) {
  len = r.count(1);
  char* c = A.make<char>((size_t)len + 1);
  if (!c) return false;
  r.raw(c, len);
  c[len] = 0;
  s = c;
  return true;
}
bool get_header(R& r, slide_wire_arena& A, slide_wire_header_t& h) {
  h.seq = r.u32(); h.stamp_sec = r.u32(); h.stamp_nsec = r.u32();
  return get_str(r, A, h.frame_id_len, h.frame_id);
}
void put_odom(W& w, const slide_wire_odometry_t& o) {      // nav_msgs/Odometry
  put_header(w, o.header);
  w.str(o.child_frame_id, o.child_frame_id_len);
  put_pose(w, o.pose);
  for (int i = 0; i < 36; ++i) w.f64(o.pose_covariance[i]);
  for (int i = 0; i < 6; ++i) w.f64(o.twist[i]);
  for (int i = 0; i < 36; ++i) w.f64(o.twist_covariance[i]);
}
bool get_odom(R& r, slide_wire_arena& A, slide_wire_odometry_t& o) {
  if (!get_header(r, A, o.header)) return false;
  if (!get_str(r, A, o.child_frame_id_len, o.child_frame_id)) return false;
  get_pose(r, o.pose);
  for (int i = 0; i < 36; ++i) o.pose_covariance[i] = r.f64();
  for (int i = 0; i < 6; ++i) o.twist[i] = r.f64();
  for (int i = 0; i < 36; ++i) o.twist_covariance[i] = r.f64();
  return true;
}

int finish_encode(const W& w, size_t* len) {
  if (len) *len = w.n;
  if (w.overflow) { sl::g_last_error = "slide_wire: output buffer too small"; return SLIDE_ERR_CAPACITY; }
  return SLIDE_OK;
}
template <class M>
int finish_decode(R& r, slide_wire_arena* A, bool ok, slide_wire_arena_t** arena, const M** msg, const M* m) {
  if (!ok) { delete A; sl::g_last_error = "slide_wire: out of memory"; return SLIDE_ERR_RUNTIME; }
  if (r.bad || r.n != r.len) {
    delete A;
    sl::g_last_error = r.bad ? "slide_wire: truncated message" : "slide_wire: trailing bytes after the message";
    return SLIDE_ERR_INVALID;
  }
  *arena = A;
  *msg = m;
  return SLIDE_OK;
}

}  // namespace

extern "C" {

void slide_wire_free(slide_wire_arena_t* arena) { delete arena; }

int slide_wire_encode_bundle(const slide_wire_bundle_t* m, uint8_t* out, size_t cap, size_t* len) {
  if (!m) return SLIDE_ERR_INVALID;
  W w{out, cap};
  w.i8(m->robot_id);                                       // PoseMstBundle.msg
  w.u32(m->n_pose_mst);
  for (uint32_t i = 0; i < m->n_pose_mst; ++i) {           // PoseMst.msg
    const slide_wire_pose_mst_t& p = m->pose_mst[i];
    put_pose(w, p.pose);
    put_pose(w, p.relative_raw_odom);
    w.u32(p.stamp_sec); w.u32(p.stamp_nsec);
    put_boxes(w, p.n_cubes, p.cubes);
    put_cyls(w, p.n_cylinders, p.cylinders);
    put_boxes(w, p.n_ellipsoids, p.ellipsoids);
  }
  w.u32(m->n_map);                                         // vector7d[]
  for (uint32_t i = 0; i < 7 * m->n_map; ++i) w.f64(m->map_label_xyz[i]);
  w.u32(m->n_tfs);                                         // interRobotTF[]
  for (uint32_t i = 0; i < m->n_tfs; ++i) {
    w.i8(m->tfs[i].host_robot_id);
    w.i8(m->tfs[i].target_robot_id);
    put_pose(w, m->tfs[i].tf_target_to_host);
  }
  return finish_encode(w, len);
}

int slide_wire_decode_bundle(const uint8_t* buf, size_t len, slide_wire_arena_t** arena, const slide_wire_bundle_t** msg) {
  if (!buf || !arena || !msg) return SLIDE_ERR_INVALID;
  slide_wire_arena* A = new slide_wire_arena;
  R r{buf, len};
  slide_wire_bundle_t* m = A->make<slide_wire_bundle_t>(1);
  bool ok = m != nullptr;
  if (ok) {
    m->robot_id = r.i8();
    m->n_pose_mst = r.count(56 + 56 + 8 + 12);
    slide_wire_pose_mst_t* pm = A->make<slide_wire_pose_mst_t>(m->n_pose_mst);
    ok = pm != nullptr;
    for (uint32_t i = 0; ok && i < m->n_pose_mst; ++i) {
      get_pose(r, pm[i].pose);
      get_pose(r, pm[i].relative_raw_odom);
      pm[i].stamp_sec = r.u32(); pm[i].stamp_nsec = r.u32();
      ok = get_boxes(r, *A, pm[i].n_cubes, pm[i].cubes) && get_cyls(r, *A, pm[i].n_cylinders, pm[i].cylinders) &&
           get_boxes(r, *A, pm[i].n_ellipsoids, pm[i].ellipsoids);
    }
    m->pose_mst = pm;
    if (ok) {
      m->n_map = r.count(56);
      double* mp = A->make<double>((size_t)7 * m->n_map);
      ok = mp != nullptr;
      for (uint32_t i = 0; ok && i < 7 * m->n_map; ++i) mp[i] = r.f64();
      m->map_label_xyz = mp;
    }
    if (ok) {
      m->n_tfs = r.count(58);
      slide_wire_inter_robot_tf_t* tf = A->make<slide_wire_inter_robot_tf_t>(m->n_tfs);
      ok = tf != nullptr;
      for (uint32_t i = 0; ok && i < m->n_tfs; ++i) {
        tf[i].host_robot_id = r.i8();
        tf[i].target_robot_id = r.i8();
        get_pose(r, tf[i].tf_target_to_host);
      }
      m->tfs = tf;
    }
  }
  return finish_decode(r, A, ok, arena, msg, m);
}

int slide_wire_encode_sync_odom(const slide_wire_sync_odom_t* m, uint8_t* out, size_t cap, size_t* len) {
  if (!m) return SLIDE_ERR_INVALID;
  W w{out, cap};
  put_header(w, m->header);                                // SemanticMeasSyncOdom.msg ("new version")
  put_boxes(w, m->n_ellipsoids, m->ellipsoids);
  put_cyls(w, m->n_cylinders, m->cylinders);
  put_boxes(w, m->n_cubes, m->cubes);
  put_odom(w, m->odometry);
  return finish_encode(w, len);
}

int slide_wire_decode_sync_odom(const uint8_t* buf, size_t len, slide_wire_arena_t** arena, const slide_wire_sync_odom_t** msg) {
  if (!buf || !arena || !msg) return SLIDE_ERR_INVALID;
  slide_wire_arena* A = new slide_wire_arena;
  R r{buf, len};
  slide_wire_sync_odom_t* m = A->make<slide_wire_sync_odom_t>(1);
  bool ok = m != nullptr;
  ok = ok && get_header(r, *A, m->header) && get_boxes(r, *A, m->n_ellipsoids, m->ellipsoids) &&
       get_cyls(r, *A, m->n_cylinders, m->cylinders) && get_boxes(r, *A, m->n_cubes, m->cubes) && get_odom(r, *A, m->odometry);
  return finish_decode(r, A, ok, arena, msg, m);
}

int slide_wire_encode_relative_meas(const slide_wire_relative_meas_t* m, uint8_t* out, size_t cap, size_t* len) {
  if (!m) return SLIDE_ERR_INVALID;
  W w{out, cap};
  put_header(w, m->header);                                // RelativeInterRobotMeasurementOdom.msg
  put_pose(w, m->relative_pose);
  w.i8(m->robot_id_observer);
  w.i8(m->robot_id_observed);
  put_odom(w, m->odometry_observer);
  put_odom(w, m->odometry_observed);
  return finish_encode(w, len);
}

int slide_wire_decode_relative_meas(const uint8_t* buf, size_t len, slide_wire_arena_t** arena,
                                    const slide_wire_relative_meas_t** msg) {
  if (!buf || !arena || !msg) return SLIDE_ERR_INVALID;
  slide_wire_arena* A = new slide_wire_arena;
  R r{buf, len};
  slide_wire_relative_meas_t* m = A->make<slide_wire_relative_meas_t>(1);
  bool ok = m != nullptr;
  if (ok) {
    ok = get_header(r, *A, m->header);
    get_pose(r, m->relative_pose);
    m->robot_id_observer = r.i8();
    m->robot_id_observed = r.i8();
    ok = ok && get_odom(r, *A, m->odometry_observer) && get_odom(r, *A, m->odometry_observed);
  }
  return finish_decode(r, A, ok, arena, msg, m);
}

// Robot::RobotObservationCb (robot.cpp:100-137), rosCylinder2CylinderObj (:170-181), rosEllipsoid2EllipObj (:183-203):
// float32 fields widen to double, poses keep the quaternion as sent (gtsam::Rot3(w, x, y, z) normalises on use).
int slide_wire_sync_odom_to_frame(const slide_wire_sync_odom_t* m, double odom_pose7[7], double* cyl_root, double* cyl_ray,
                                  double* cyl_radius, int32_t* cyl_label, double* cube_pose7, double* cube_scale,
                                  int32_t* cube_label, double* ell_pose7, double* ell_scale, int32_t* ell_label) {
  if (!m || !odom_pose7) return SLIDE_ERR_INVALID;
  if ((m->n_cylinders && !(cyl_root && cyl_ray && cyl_radius && cyl_label)) || (m->n_cubes && !(cube_pose7 && cube_scale && cube_label)) ||
      (m->n_ellipsoids && !(ell_pose7 && ell_scale && ell_label)))
    return SLIDE_ERR_INVALID;
  for (int i = 0; i < 3; ++i) odom_pose7[i] = m->odometry.pose.p[i];
  for (int i = 0; i < 4; ++i) odom_pose7[3 + i] = m->odometry.pose.q[i];
  for (uint32_t i = 0; i < m->n_cylinders; ++i) {
    const slide_wire_cylinder_t& c = m->cylinders[i];
    for (int k = 0; k < 3; ++k) { cyl_root[3 * i + k] = (double)c.root[k]; cyl_ray[3 * i + k] = (double)c.ray[k]; }
    cyl_radius[i] = (double)c.radius;
    cyl_label[i] = c.semantic_label;
  }
  auto boxes = [](uint32_t n, const slide_wire_box_t* b, double* pose7, double* scale, int32_t* label) {
    for (uint32_t i = 0; i < n; ++i) {
      for (int k = 0; k < 3; ++k) pose7[7 * i + k] = b[i].pose.p[k];
      for (int k = 0; k < 4; ++k) pose7[7 * i + 3 + k] = b[i].pose.q[k];
      for (int k = 0; k < 3; ++k) scale[3 * i + k] = (double)b[i].dim[k];
      label[i] = b[i].semantic_label;
    }
  };
  boxes(m->n_cubes, m->cubes, cube_pose7, cube_scale, cube_label);
  boxes(m->n_ellipsoids, m->ellipsoids, ell_pose7, ell_scale, ell_label);
  return SLIDE_OK;
}

}  // extern "C"

// ---- rosbag v2.0 -------------------------------------------------------------------------------------------------------------
// File = "#ROSBAG V2.0\n" + records; record = uint32 header_len, header, uint32 data_len, data; header = fields
// "uint32 field_len, name=value".  op (1 byte): 0x03 bag header, 0x05 chunk (compression, size; data = records), 0x07
// connection (conn, topic; data = connection header with type / md5sum / message_definition), 0x02 message data (conn, time),
// 0x04 index data, 0x06 chunk info.  The reader walks every record in file order (the index records are not needed).
struct slide_bag {
  std::vector<uint8_t> file;
  struct Conn { uint32_t id; std::string topic, type, md5; };
  struct Msg { uint32_t conn, sec, nsec; uint64_t off, len; uint64_t order; };
  std::vector<Conn> conns;
  std::vector<Msg> msgs;
};

namespace {

struct Fields {
  const uint8_t* op = nullptr; size_t op_len = 0;
  const uint8_t* conn = nullptr; size_t conn_len = 0;
  const uint8_t* time = nullptr; size_t time_len = 0;
  std::string topic, compression, type, md5;
};
bool parse_fields(const uint8_t* h, size_t hl, Fields& f) {
  size_t n = 0;
  while (n < hl) {
    if (hl - n < 4) return false;
    uint32_t fl;
    memcpy(&fl, h + n, 4);
    n += 4;
    if (fl > hl - n) return false;
    const uint8_t* fe = static_cast<const uint8_t*>(memchr(h + n, '=', fl));
    if (!fe) return false;
    const std::string name(reinterpret_cast<const char*>(h + n), fe - (h + n));
    const uint8_t* v = fe + 1;
    const size_t vl = fl - (size_t)(v - (h + n));
    if (name == "op") { f.op = v; f.op_len = vl; }
    else if (name == "conn") { f.conn = v; f.conn_len = vl; }
    else if (name == "time") { f.time = v; f.time_len = vl; }
    else if (name == "topic") f.topic.assign(reinterpret_cast<const char*>(v), vl);
    else if (name == "compression") f.compression.assign(reinterpret_cast<const char*>(v), vl);
    else if (name == "type") f.type.assign(reinterpret_cast<const char*>(v), vl);
    else if (name == "md5sum") f.md5.assign(reinterpret_cast<const char*>(v), vl);
    n += fl;
  }
  return true;
}

// records in [beg, end) of bag.file; depth 1 = inside a chunk
int walk(slide_bag& b, size_t beg, size_t end, int depth) {
  const uint8_t* F = b.file.data();
  size_t n = beg;
  while (n < end) {
    if (end - n < 4) return SLIDE_ERR_INVALID;
    uint32_t hl;
    memcpy(&hl, F + n, 4);
    n += 4;
    if (hl > end - n) return SLIDE_ERR_INVALID;
    Fields f;
    if (!parse_fields(F + n, hl, f)) return SLIDE_ERR_INVALID;
    n += hl;
    if (end - n < 4) return SLIDE_ERR_INVALID;
    uint32_t dl;
    memcpy(&dl, F + n, 4);
    n += 4;
    if (dl > end - n) return SLIDE_ERR_INVALID;
    if (!f.op || f.op_len != 1) return SLIDE_ERR_INVALID;
    switch (*f.op) {
      case 0x05: {                                         // chunk
        if (depth != 0) return SLIDE_ERR_INVALID;
        if (f.compression != "none") {
          sl::g_last_error = "slide_bag: chunk compression '" + f.compression + "' is not supported (rosbag decompress first)";
          return SLIDE_ERR_RUNTIME;
        }
        const int rc = walk(b, n, n + dl, 1);
        if (rc != SLIDE_OK) return rc;
        break;
      }
      case 0x07: {                                         // connection: the data is the connection header
        if (!f.conn || f.conn_len != 4) return SLIDE_ERR_INVALID;
        uint32_t id;
        memcpy(&id, f.conn, 4);
        Fields ch;
        if (!parse_fields(F + n, dl, ch)) return SLIDE_ERR_INVALID;
        bool known = false;
        for (const auto& c : b.conns) known = known || c.id == id;
        if (!known) b.conns.push_back({id, f.topic.empty() ? ch.topic : f.topic, ch.type, ch.md5});
        break;
      }
      case 0x02: {                                         // message data
        if (!f.conn || f.conn_len != 4 || !f.time || f.time_len != 8) return SLIDE_ERR_INVALID;
        slide_bag::Msg m;
        memcpy(&m.conn, f.conn, 4);
        memcpy(&m.sec, f.time, 4);
        memcpy(&m.nsec, f.time + 4, 4);
        m.off = n;
        m.len = dl;
        m.order = b.msgs.size();
        b.msgs.push_back(m);
        break;
      }
      default: break;                                      // bag header, index data, chunk info: not needed
    }
    n += dl;
  }
  return SLIDE_OK;
}

}  // namespace

extern "C" {

int slide_bag_open(const char* path, slide_bag_t** bag) {
  if (!path || !bag) return SLIDE_ERR_INVALID;
  FILE* fh = fopen(path, "rb");
  if (!fh) { sl::g_last_error = std::string("slide_bag: cannot open ") + path; return SLIDE_ERR_INVALID; }
  slide_bag* b = new slide_bag;
  fseek(fh, 0, SEEK_END);
  const long sz = ftell(fh);
  fseek(fh, 0, SEEK_SET);
  b->file.resize(sz > 0 ? (size_t)sz : 0);
  const size_t got = b->file.empty() ? 0 : fread(b->file.data(), 1, b->file.size(), fh);
  fclose(fh);
  static const char magic[] = "#ROSBAG V2.0\n";
  if (got != b->file.size() || b->file.size() < 13 || memcmp(b->file.data(), magic, 13) != 0) {
    delete b;
    sl::g_last_error = "slide_bag: not a rosbag v2.0 file";
    return SLIDE_ERR_INVALID;
  }
  const int rc = walk(*b, 13, b->file.size(), 0);
  if (rc != SLIDE_OK) {
    if (rc == SLIDE_ERR_INVALID) sl::g_last_error = "slide_bag: malformed record";
    delete b;
    return rc;
  }
  std::stable_sort(b->msgs.begin(), b->msgs.end(), [](const slide_bag::Msg& x, const slide_bag::Msg& y) {
    return x.sec != y.sec ? x.sec < y.sec : x.nsec < y.nsec;
  });
  *bag = b;
  return SLIDE_OK;
}

void slide_bag_close(slide_bag_t* bag) { delete bag; }

int slide_bag_num_connections(const slide_bag_t* bag, int32_t* n) {
  if (!bag || !n) return SLIDE_ERR_INVALID;
  *n = (int32_t)bag->conns.size();
  return SLIDE_OK;
}

int slide_bag_connection(const slide_bag_t* bag, int32_t i, uint32_t* conn_id, const char** topic, const char** datatype,
                         const char** md5sum) {
  if (!bag || i < 0 || (size_t)i >= bag->conns.size()) return SLIDE_ERR_INVALID;
  const auto& c = bag->conns[i];
  if (conn_id) *conn_id = c.id;
  if (topic) *topic = c.topic.c_str();
  if (datatype) *datatype = c.type.c_str();
  if (md5sum) *md5sum = c.md5.c_str();
  return SLIDE_OK;
}

int slide_bag_num_messages(const slide_bag_t* bag, int64_t* n) {
  if (!bag || !n) return SLIDE_ERR_INVALID;
  *n = (int64_t)bag->msgs.size();
  return SLIDE_OK;
}

int slide_bag_message(const slide_bag_t* bag, int64_t i, uint32_t* conn_id, uint32_t* sec, uint32_t* nsec, const uint8_t** data,
                      uint64_t* len) {
  if (!bag || i < 0 || (size_t)i >= bag->msgs.size()) return SLIDE_ERR_INVALID;
  const auto& m = bag->msgs[i];
  if (conn_id) *conn_id = m.conn;
  if (sec) *sec = m.sec;
  if (nsec) *nsec = m.nsec;
  if (data) *data = bag->file.data() + m.off;
  if (len) *len = m.len;
  return SLIDE_OK;
}

}  // extern "C"
