/* slide_wire.h — C-ABI of the wire codec and bag reader: the reference's `sloam_msgs` on the
 * wire (ROS-1 serialisation, little-endian) and `rosbag` v2.0 files, without ROS.  Host code only (no device work).
 *
 * Reference interfaces replaced:
 *   backend/sloam_msgs/msg/SemanticMeasSyncOdom.msg, PoseMst.msg, PoseMstBundle.msg, interRobotTF.msg, vector7d.msg,
 *   ROSCube.msg, ROSCylinder.msg, ROSEllipsoid.msg, RelativeInterRobotMeasurementOdom.msg (generated roscpp serialisers);
 *   the converters next to them: Robot::RobotObservationCb / rosCylinder2CylinderObj / rosEllipsoid2EllipObj
 *   (backend/sloam/src/core/robot.cpp:100-199), databaseManager::poseMstCb_ / runCommunication_
 *   (databaseManager.cpp:98-160, 219-279), obj2RosObjMsg / gtsamPoseToRosPose / toSE3Pose (databaseManager.h:233-341).
 *
 * Encoding rules (ROS-1): little-endian scalars; T[N] = N elements; T[] = uint32 count + elements; string = uint32 length +
 * bytes; time = uint32 sec + uint32 nsec; nested messages inline.  geometry_msgs/Pose = position x y z, orientation x y z w
 * (7 float64 = 56 bytes) — the same order as this library's pose7.  float32 fields (dim / scale / root / ray / radius) are
 * where the reference quantises: encode rounds the given float, decode widens exactly as the reference's converters do.
 *
 * Decoded messages live in an arena owned by the caller (slide_wire_free); encoders write into a caller buffer and report the
 * length (call with out = NULL to size).  Every function returns SLIDE_OK or a negative SLIDE_ERR_* of slide_gpu.h;
 * SLIDE_ERR_INVALID on a truncated or over-long buffer.
 */
#ifndef SLIDE_WIRE_H
#define SLIDE_WIRE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double p[3]; double q[4]; } slide_wire_pose_t;            /* geometry_msgs/Pose: x y z, qx qy qz qw */

/* ROSCube.msg / ROSEllipsoid.msg (same layout: float32[3] dim|scale, int8 semantic_label, Pose): 69 bytes on the wire */
typedef struct {
  float dim[3];
  int8_t semantic_label;
  slide_wire_pose_t pose;
} slide_wire_box_t;

/* ROSCylinder.msg: float32[3] root, float32[3] ray, float64[] radii, float32 radius, int64 id, int8 semantic_label */
typedef struct {
  float root[3];
  float ray[3];
  uint32_t n_radii;
  const double* radii;
  float radius;
  int64_t id;
  int8_t semantic_label;
} slide_wire_cylinder_t;

/* PoseMst.msg */
typedef struct {
  slide_wire_pose_t pose;
  slide_wire_pose_t relative_raw_odom;
  uint32_t stamp_sec, stamp_nsec;
  uint32_t n_cubes;      const slide_wire_box_t* cubes;
  uint32_t n_cylinders;  const slide_wire_cylinder_t* cylinders;
  uint32_t n_ellipsoids; const slide_wire_box_t* ellipsoids;
} slide_wire_pose_mst_t;

/* interRobotTF.msg: 58 bytes on the wire */
typedef struct {
  int8_t host_robot_id, target_robot_id;
  slide_wire_pose_t tf_target_to_host;
} slide_wire_inter_robot_tf_t;

/* PoseMstBundle.msg */
typedef struct {
  int8_t robot_id;
  uint32_t n_pose_mst; const slide_wire_pose_mst_t* pose_mst;
  uint32_t n_map;      const double* map_label_xyz;          /* vector7d[]: 7 doubles per entry */
  uint32_t n_tfs;      const slide_wire_inter_robot_tf_t* tfs;
} slide_wire_bundle_t;

/* std_msgs/Header */
typedef struct {
  uint32_t seq, stamp_sec, stamp_nsec;
  uint32_t frame_id_len; const char* frame_id;               /* not NUL-terminated on the wire; decode adds a NUL */
} slide_wire_header_t;

/* nav_msgs/Odometry */
typedef struct {
  slide_wire_header_t header;
  uint32_t child_frame_id_len; const char* child_frame_id;
  slide_wire_pose_t pose;
  double pose_covariance[36];
  double twist[6];                                            /* linear x y z, angular x y z */
  double twist_covariance[36];
} slide_wire_odometry_t;

/* SemanticMeasSyncOdom.msg */
typedef struct {
  slide_wire_header_t header;
  uint32_t n_ellipsoids; const slide_wire_box_t* ellipsoids;
  uint32_t n_cylinders;  const slide_wire_cylinder_t* cylinders;
  uint32_t n_cubes;      const slide_wire_box_t* cubes;
  slide_wire_odometry_t odometry;
} slide_wire_sync_odom_t;

/* RelativeInterRobotMeasurementOdom.msg */
typedef struct {
  slide_wire_header_t header;
  slide_wire_pose_t relative_pose;
  int8_t robot_id_observer, robot_id_observed;
  slide_wire_odometry_t odometry_observer, odometry_observed;
} slide_wire_relative_meas_t;

typedef struct slide_wire_arena slide_wire_arena_t;           /* owns everything a decoded message points to */
void slide_wire_free(slide_wire_arena_t* arena);

/* encoders: *len = bytes needed; writes when out != NULL and cap >= *len (else SLIDE_ERR_CAPACITY) */
int slide_wire_encode_bundle(const slide_wire_bundle_t* msg, uint8_t* out, size_t cap, size_t* len);
int slide_wire_encode_sync_odom(const slide_wire_sync_odom_t* msg, uint8_t* out, size_t cap, size_t* len);
int slide_wire_encode_relative_meas(const slide_wire_relative_meas_t* msg, uint8_t* out, size_t cap, size_t* len);
/* decoders: the whole buffer must be consumed */
int slide_wire_decode_bundle(const uint8_t* buf, size_t len, slide_wire_arena_t** arena, const slide_wire_bundle_t** msg);
int slide_wire_decode_sync_odom(const uint8_t* buf, size_t len, slide_wire_arena_t** arena, const slide_wire_sync_odom_t** msg);
int slide_wire_decode_relative_meas(const uint8_t* buf, size_t len, slide_wire_arena_t** arena,
                                    const slide_wire_relative_meas_t** msg);

/* SemanticMeasSyncOdom -> the arrays slide_backend_process_frame takes (slide_detections_t of slide_gpu.h), exactly as
 * Robot::RobotObservationCb builds its Observation (robot.cpp:100-137): body-frame objects, float32 fields widened to double,
 * quaternions kept as sent.  Arrays must hold n_cylinders / n_cubes / n_ellipsoids entries (3 | 3 | 1 | 1 and 7 | 3 | 1). */
int slide_wire_sync_odom_to_frame(const slide_wire_sync_odom_t* msg, double odom_pose7[7], double* cyl_root, double* cyl_ray,
                                  double* cyl_radius, int32_t* cyl_label, double* cube_pose7, double* cube_scale,
                                  int32_t* cube_label, double* ell_pose7, double* ell_scale, int32_t* ell_label);

/* ---- rosbag v2.0 reader (uncompressed chunks; "bz2" / "lz4" chunks give SLIDE_ERR_RUNTIME) ------------------------------- */
typedef struct slide_bag slide_bag_t;
int slide_bag_open(const char* path, slide_bag_t** bag);
void slide_bag_close(slide_bag_t* bag);
int slide_bag_num_connections(const slide_bag_t* bag, int32_t* n);
/* topic / datatype / md5sum are NUL-terminated and live as long as the bag */
int slide_bag_connection(const slide_bag_t* bag, int32_t i, uint32_t* conn_id, const char** topic, const char** datatype,
                         const char** md5sum);
int slide_bag_num_messages(const slide_bag_t* bag, int64_t* n);
/* message i in play order (receive time, then file order): connection id, receive time, serialised payload (points into the
 * bag's memory) */
int slide_bag_message(const slide_bag_t* bag, int64_t i, uint32_t* conn_id, uint32_t* sec, uint32_t* nsec, const uint8_t** data,
                      uint64_t* len);

#ifdef __cplusplus
}
#endif
#endif
