"""Host mirror of include/slide_wire.h: the reference's `sloam_msgs` on the wire (ROS-1 serialisation) and rosbag v2.0
files, without ROS.

Messages are plain dicts with the field names of backend/sloam_msgs/msg/*.msg:
  pose            = 7 floats  x y z qx qy qz qw                       (geometry_msgs/Pose)
  cube/ellipsoid  = dict(dim|scale=3 float32, semantic_label, pose)   (ROSCube.msg / ROSEllipsoid.msg)
  cylinder        = dict(root, ray, radii, radius, id, semantic_label) (ROSCylinder.msg)
  PoseMst         = dict(pose, relativeRawOdom, stamp=(sec, nsec), cubes, cylinders, ellipsoids)
  PoseMstBundle   = dict(robotID, poseMstPair, map_of_labelXYZ (n x 7), interRobotTFs)
  SemanticMeasSyncOdom = dict(header, ellipsoid_factors, cylinder_factors, cuboid_factors, odometry)
Encoding / decoding is done by the C++ library; nothing here touches the oracle.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api

SLIDE_OK = 0


class Pose(C.Structure):
    _fields_ = [("p", C.c_double * 3), ("q", C.c_double * 4)]


class Box(C.Structure):
    _fields_ = [("dim", C.c_float * 3), ("semantic_label", C.c_int8), ("pose", Pose)]


class Cylinder(C.Structure):
    _fields_ = [("root", C.c_float * 3), ("ray", C.c_float * 3), ("n_radii", C.c_uint32), ("radii", C.POINTER(C.c_double)),
                ("radius", C.c_float), ("id", C.c_int64), ("semantic_label", C.c_int8)]


class PoseMst(C.Structure):
    _fields_ = [("pose", Pose), ("relative_raw_odom", Pose), ("stamp_sec", C.c_uint32), ("stamp_nsec", C.c_uint32),
                ("n_cubes", C.c_uint32), ("cubes", C.POINTER(Box)), ("n_cylinders", C.c_uint32), ("cylinders", C.POINTER(Cylinder)),
                ("n_ellipsoids", C.c_uint32), ("ellipsoids", C.POINTER(Box))]


class InterRobotTF(C.Structure):
    _fields_ = [("host_robot_id", C.c_int8), ("target_robot_id", C.c_int8), ("tf_target_to_host", Pose)]


class Bundle(C.Structure):
    _fields_ = [("robot_id", C.c_int8), ("n_pose_mst", C.c_uint32), ("pose_mst", C.POINTER(PoseMst)), ("n_map", C.c_uint32),
                ("map_label_xyz", C.POINTER(C.c_double)), ("n_tfs", C.c_uint32), ("tfs", C.POINTER(InterRobotTF))]


class Header(C.Structure):
    _fields_ = [("seq", C.c_uint32), ("stamp_sec", C.c_uint32), ("stamp_nsec", C.c_uint32), ("frame_id_len", C.c_uint32),
                ("frame_id", C.c_char_p)]


class Odometry(C.Structure):
    _fields_ = [("header", Header), ("child_frame_id_len", C.c_uint32), ("child_frame_id", C.c_char_p), ("pose", Pose),
                ("pose_covariance", C.c_double * 36), ("twist", C.c_double * 6), ("twist_covariance", C.c_double * 36)]


class SyncOdom(C.Structure):
    _fields_ = [("header", Header), ("n_ellipsoids", C.c_uint32), ("ellipsoids", C.POINTER(Box)), ("n_cylinders", C.c_uint32),
                ("cylinders", C.POINTER(Cylinder)), ("n_cubes", C.c_uint32), ("cubes", C.POINTER(Box)), ("odometry", Odometry)]


class RelativeMeas(C.Structure):
    _fields_ = [("header", Header), ("relative_pose", Pose), ("robot_id_observer", C.c_int8), ("robot_id_observed", C.c_int8),
                ("odometry_observer", Odometry), ("odometry_observed", Odometry)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = api.lib()
        L.slide_wire_free.argtypes = [C.c_void_p]
        L.slide_wire_free.restype = None
        for n, T in (("bundle", Bundle), ("sync_odom", SyncOdom), ("relative_meas", RelativeMeas)):
            f = getattr(L, "slide_wire_encode_" + n)
            f.argtypes = [C.POINTER(T), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
            f.restype = C.c_int
            f = getattr(L, "slide_wire_decode_" + n)
            f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.POINTER(T))]
            f.restype = C.c_int
        L.slide_wire_sync_odom_to_frame.argtypes = [C.POINTER(SyncOdom)] + [C.c_void_p] * 11
        L.slide_wire_sync_odom_to_frame.restype = C.c_int
        L.slide_bag_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.slide_bag_close.argtypes = [C.c_void_p]
        L.slide_bag_close.restype = None
        L.slide_bag_num_connections.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.slide_bag_connection.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                           C.POINTER(C.c_char_p)]
        L.slide_bag_num_messages.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.slide_bag_message.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


class WireError(RuntimeError):
    def __init__(self, code):
        self.code = code
        msg = api.lib().slide_last_error()
        super().__init__(f"slide_wire error {code}: {msg.decode() if msg else ''}")


def _check(rc):
    if rc != SLIDE_OK:
        raise WireError(rc)


# ---- dict -> ctypes (keeps the backing arrays alive in `keep`) --------------------------------------------------------------
def _pose(v):
    v = np.asarray(v, np.float64).reshape(7)
    return Pose((C.c_double * 3)(*v[:3]), (C.c_double * 4)(*v[3:]))


def _boxes(items, key, keep):
    arr = (Box * max(len(items), 1))()
    for i, b in enumerate(items):
        arr[i] = Box((C.c_float * 3)(*np.asarray(b[key], np.float32)), int(b["semantic_label"]), _pose(b["pose"]))
    keep.append(arr)
    return len(items), C.cast(arr, C.POINTER(Box))


def _cyls(items, keep):
    arr = (Cylinder * max(len(items), 1))()
    for i, c in enumerate(items):
        rad = np.ascontiguousarray(c.get("radii", []), np.float64)
        keep.append(rad)
        arr[i] = Cylinder((C.c_float * 3)(*np.asarray(c["root"], np.float32)), (C.c_float * 3)(*np.asarray(c["ray"], np.float32)),
                          len(rad), rad.ctypes.data_as(C.POINTER(C.c_double)), float(np.float32(c["radius"])), int(c.get("id", 0)),
                          int(c["semantic_label"]))
    keep.append(arr)
    return len(items), C.cast(arr, C.POINTER(Cylinder))


def _header(h, keep):
    fid = h.get("frame_id", "").encode()
    keep.append(fid)
    sec, nsec = h.get("stamp", (0, 0))
    return Header(int(h.get("seq", 0)), int(sec), int(nsec), len(fid), fid)


def _odom(o, keep):
    cid = o.get("child_frame_id", "").encode()
    keep.append(cid)
    return Odometry(_header(o.get("header", {}), keep), len(cid), cid, _pose(o["pose"]),
                    (C.c_double * 36)(*np.asarray(o.get("pose_covariance", np.zeros(36)), np.float64).reshape(36)),
                    (C.c_double * 6)(*np.asarray(o.get("twist", np.zeros(6)), np.float64).reshape(6)),
                    (C.c_double * 36)(*np.asarray(o.get("twist_covariance", np.zeros(36)), np.float64).reshape(36)))


def _encode(fn, msg):
    n = C.c_size_t(0)
    _check(fn(C.byref(msg), None, 0, C.byref(n)))
    buf = (C.c_uint8 * max(n.value, 1))()
    _check(fn(C.byref(msg), buf, n.value, C.byref(n)))
    return bytes(buf[: n.value])


def encode_bundle(b) -> bytes:
    keep = []
    pm = (PoseMst * max(len(b["poseMstPair"]), 1))()
    for i, p in enumerate(b["poseMstPair"]):
        nc, cubes = _boxes(p.get("cubes", []), "dim", keep)
        ny, cyls = _cyls(p.get("cylinders", []), keep)
        ne, ells = _boxes(p.get("ellipsoids", []), "scale", keep)
        sec, nsec = p.get("stamp", (0, 0))
        pm[i] = PoseMst(_pose(p["pose"]), _pose(p["relativeRawOdom"]), int(sec), int(nsec), nc, cubes, ny, cyls, ne, ells)
    mp = np.ascontiguousarray(b.get("map_of_labelXYZ", np.zeros((0, 7))), np.float64).reshape(-1, 7)
    tfs = (InterRobotTF * max(len(b.get("interRobotTFs", [])), 1))()
    for i, t in enumerate(b.get("interRobotTFs", [])):
        tfs[i] = InterRobotTF(int(t["hostRobotID"]), int(t["targetRobotID"]), _pose(t["TFfromTarget2Host"]))
    msg = Bundle(int(b["robotID"]), len(b["poseMstPair"]), C.cast(pm, C.POINTER(PoseMst)), len(mp),
                 mp.ctypes.data_as(C.POINTER(C.c_double)), len(b.get("interRobotTFs", [])), C.cast(tfs, C.POINTER(InterRobotTF)))
    return _encode(lib().slide_wire_encode_bundle, msg)


def _sync_odom_struct(m, keep):
    ne, ells = _boxes(m.get("ellipsoid_factors", []), "scale", keep)
    ny, cyls = _cyls(m.get("cylinder_factors", []), keep)
    nc, cubes = _boxes(m.get("cuboid_factors", []), "dim", keep)
    return SyncOdom(_header(m.get("header", {}), keep), ne, ells, ny, cyls, nc, cubes, _odom(m["odometry"], keep))


def encode_sync_odom(m) -> bytes:
    keep = []
    return _encode(lib().slide_wire_encode_sync_odom, _sync_odom_struct(m, keep))


def encode_relative_meas(m) -> bytes:
    keep = []
    msg = RelativeMeas(_header(m.get("header", {}), keep), _pose(m["relativePose"]), int(m["robotIdObserver"]),
                       int(m["robotIdObserved"]), _odom(m["odometryObserver"], keep), _odom(m["odometryObserved"], keep))
    return _encode(lib().slide_wire_encode_relative_meas, msg)


# ---- ctypes -> dict ----------------------------------------------------------------------------------------------------------
def _pose_out(p):
    return np.array(list(p.p) + list(p.q))


def _boxes_out(n, ptr, key):
    return [{key: np.array(list(ptr[i].dim), np.float32), "semantic_label": int(ptr[i].semantic_label), "pose": _pose_out(ptr[i].pose)}
            for i in range(n)]


def _cyls_out(n, ptr):
    return [dict(root=np.array(list(ptr[i].root), np.float32), ray=np.array(list(ptr[i].ray), np.float32),
                 radii=np.array([ptr[i].radii[k] for k in range(ptr[i].n_radii)], np.float64), radius=np.float32(ptr[i].radius),
                 id=int(ptr[i].id), semantic_label=int(ptr[i].semantic_label)) for i in range(n)]


def _header_out(h):
    return dict(seq=int(h.seq), stamp=(int(h.stamp_sec), int(h.stamp_nsec)), frame_id=(h.frame_id or b"")[: h.frame_id_len].decode())


def _odom_out(o):
    return dict(header=_header_out(o.header), child_frame_id=(o.child_frame_id or b"")[: o.child_frame_id_len].decode(),
                pose=_pose_out(o.pose), pose_covariance=np.array(list(o.pose_covariance)), twist=np.array(list(o.twist)),
                twist_covariance=np.array(list(o.twist_covariance)))


def _decode(fn, T, data):
    data = bytes(data)
    arena = C.c_void_p()
    msg = C.POINTER(T)()
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    _check(fn(buf, len(data), C.byref(arena), C.byref(msg)))
    return arena, msg


def decode_bundle(data):
    arena, mp = _decode(lib().slide_wire_decode_bundle, Bundle, data)
    try:
        m = mp.contents
        out = dict(robotID=int(m.robot_id), poseMstPair=[], interRobotTFs=[],
                   map_of_labelXYZ=np.array([m.map_label_xyz[i] for i in range(7 * m.n_map)]).reshape(-1, 7))
        for i in range(m.n_pose_mst):
            p = m.pose_mst[i]
            out["poseMstPair"].append(dict(pose=_pose_out(p.pose), relativeRawOdom=_pose_out(p.relative_raw_odom),
                                           stamp=(int(p.stamp_sec), int(p.stamp_nsec)), cubes=_boxes_out(p.n_cubes, p.cubes, "dim"),
                                           cylinders=_cyls_out(p.n_cylinders, p.cylinders),
                                           ellipsoids=_boxes_out(p.n_ellipsoids, p.ellipsoids, "scale")))
        for i in range(m.n_tfs):
            t = m.tfs[i]
            out["interRobotTFs"].append(dict(hostRobotID=int(t.host_robot_id), targetRobotID=int(t.target_robot_id),
                                             TFfromTarget2Host=_pose_out(t.tf_target_to_host)))
        return out
    finally:
        lib().slide_wire_free(arena)


def _sync_odom_out(m):
    return dict(header=_header_out(m.header), ellipsoid_factors=_boxes_out(m.n_ellipsoids, m.ellipsoids, "scale"),
                cylinder_factors=_cyls_out(m.n_cylinders, m.cylinders), cuboid_factors=_boxes_out(m.n_cubes, m.cubes, "dim"),
                odometry=_odom_out(m.odometry))


def decode_sync_odom(data):
    arena, mp = _decode(lib().slide_wire_decode_sync_odom, SyncOdom, data)
    try:
        return _sync_odom_out(mp.contents)
    finally:
        lib().slide_wire_free(arena)


def decode_relative_meas(data):
    arena, mp = _decode(lib().slide_wire_decode_relative_meas, RelativeMeas, data)
    try:
        m = mp.contents
        return dict(header=_header_out(m.header), relativePose=_pose_out(m.relative_pose), robotIdObserver=int(m.robot_id_observer),
                    robotIdObserved=int(m.robot_id_observed), odometryObserver=_odom_out(m.odometry_observer),
                    odometryObserved=_odom_out(m.odometry_observed))
    finally:
        lib().slide_wire_free(arena)


def sync_odom_to_frame(data):
    """Serialised SemanticMeasSyncOdom -> (odometry pose7, detections dict for SlideBackend.process_frame), the conversion of
    Robot::RobotObservationCb (robot.cpp:100-137)."""
    arena, mp = _decode(lib().slide_wire_decode_sync_odom, SyncOdom, data)
    try:
        m = mp.contents
        ny, nc, ne = m.n_cylinders, m.n_cubes, m.n_ellipsoids
        pose7 = np.zeros(7)
        det = dict(cyl_root=np.zeros((ny, 3)), cyl_ray=np.zeros((ny, 3)), cyl_radius=np.zeros(ny), cyl_label=np.zeros(ny, np.int32),
                   cube_pose7=np.zeros((nc, 7)), cube_scale=np.zeros((nc, 3)), cube_label=np.zeros(nc, np.int32),
                   ell_pose7=np.zeros((ne, 7)), ell_scale=np.zeros((ne, 3)), ell_label=np.zeros(ne, np.int32))
        order = ("cyl_root", "cyl_ray", "cyl_radius", "cyl_label", "cube_pose7", "cube_scale", "cube_label", "ell_pose7", "ell_scale",
                 "ell_label")
        _check(lib().slide_wire_sync_odom_to_frame(mp, pose7.ctypes.data, *[det[k].ctypes.data for k in order]))
        return pose7, det, _header_out(m.header)
    finally:
        lib().slide_wire_free(arena)


class Bag:
    """rosbag v2.0 reader (uncompressed chunks).  Iterating yields (topic, datatype, (sec, nsec), payload bytes) in play order."""

    def __init__(self, path):
        self.h = C.c_void_p()
        _check(lib().slide_bag_open(str(path).encode(), C.byref(self.h)))
        n = C.c_int32()
        _check(lib().slide_bag_num_connections(self.h, C.byref(n)))
        self.connections = {}
        for i in range(n.value):
            cid, t, d, m = C.c_uint32(), C.c_char_p(), C.c_char_p(), C.c_char_p()
            _check(lib().slide_bag_connection(self.h, i, C.byref(cid), C.byref(t), C.byref(d), C.byref(m)))
            self.connections[cid.value] = dict(topic=t.value.decode(), datatype=d.value.decode(), md5sum=m.value.decode())

    def __len__(self):
        n = C.c_int64()
        _check(lib().slide_bag_num_messages(self.h, C.byref(n)))
        return n.value

    def __iter__(self):
        for i in range(len(self)):
            cid, sec, nsec, data, ln = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_void_p(), C.c_uint64()
            _check(lib().slide_bag_message(self.h, i, C.byref(cid), C.byref(sec), C.byref(nsec), C.byref(data), C.byref(ln)))
            c = self.connections.get(cid.value, dict(topic="", datatype=""))
            yield c["topic"], c["datatype"], (sec.value, nsec.value), C.string_at(data, ln.value)

    def close(self):
        if self.h:
            lib().slide_bag_close(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def inflate_bag(src, dst):
    """Rewrite a rosbag v2.0 file with bz2-compressed chunks as one with uncompressed chunks (what `rosbag decompress` does), so
    that `Bag` can read it.  Host-side convenience in Python (the C++ reader links no decompressor); lz4 chunks are refused.
    The index / chunk-info records are carried over unchanged — their file offsets are stale, this library's reader walks the
    records and never uses them."""
    import bz2
    import struct
    data = open(src, "rb").read()
    magic = b"#ROSBAG V2.0\n"
    if not data.startswith(magic):
        raise ValueError("not a rosbag v2.0 file")
    out = [magic]
    n = len(magic)
    while n < len(data):
        hl, = struct.unpack_from("<I", data, n)
        hdr = data[n + 4:n + 4 + hl]
        dl, = struct.unpack_from("<I", data, n + 4 + hl)
        body = data[n + 8 + hl:n + 8 + hl + dl]
        if len(hdr) != hl or len(body) != dl:
            raise ValueError("truncated bag")
        n += 8 + hl + dl
        fields, m = {}, 0
        while m < hl:
            fl, = struct.unpack_from("<I", hdr, m)
            k, _, v = hdr[m + 4:m + 4 + fl].partition(b"=")
            fields[k] = v
            m += 4 + fl
        if fields.get(b"op") == b"\x05" and fields.get(b"compression", b"none") != b"none":
            if fields[b"compression"] != b"bz2":
                raise ValueError("unsupported chunk compression " + fields[b"compression"].decode())
            body = bz2.decompress(body)
            fields[b"compression"] = b"none"
            fields[b"size"] = struct.pack("<I", len(body))
            hdr = b"".join(struct.pack("<I", len(k) + 1 + len(v)) + k + b"=" + v for k, v in fields.items())
        out.append(struct.pack("<I", len(hdr)) + hdr + struct.pack("<I", len(body)) + body)
    with open(dst, "wb") as fh:
        fh.write(b"".join(out))
