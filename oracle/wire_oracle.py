"""ORACLE (test infrastructure only — never imported by the product path).

Pure-Python restatement of how roscpp serialises the reference's messages (backend/sloam_msgs/msg/*.msg) and of the
rosbag v2.0 container, written from the public format rules with `struct`:
  little-endian scalars; T[N] inline; T[] = uint32 count + elements; string = uint32 length + bytes; time = 2 x uint32;
  geometry_msgs/Pose = position x y z + orientation x y z w (float64); nav_msgs/Odometry = Header, child_frame_id,
  PoseWithCovariance (Pose + float64[36]), TwistWithCovariance (Vector3 linear + Vector3 angular + float64[36]).
Parity pins: the byte counts the reference itself states for these messages — ROSCube / ROSEllipsoid 69, Pose 56,
vector7d 56, interRobotTF 58 (PoseMst.msg comments, databaseManager.cpp:240-272).  No serialised fixture of these
messages exists in the reference (it ships no bag and no test vectors): "parity unpinned" beyond those sizes and
the published ROS-1 rules.
"""
from __future__ import annotations

import struct

import numpy as np


def pose(p):
    return struct.pack("<7d", *np.asarray(p, np.float64).reshape(7))


def box(b, key):                                         # ROSCube.msg / ROSEllipsoid.msg
    return struct.pack("<3f", *np.asarray(b[key], np.float32)) + struct.pack("<b", int(b["semantic_label"])) + pose(b["pose"])


def cylinder(c):                                         # ROSCylinder.msg
    rad = np.asarray(c.get("radii", []), np.float64)
    return (struct.pack("<3f", *np.asarray(c["root"], np.float32)) + struct.pack("<3f", *np.asarray(c["ray"], np.float32)) +
            struct.pack("<I", len(rad)) + struct.pack("<%dd" % len(rad), *rad) + struct.pack("<f", np.float32(c["radius"])) +
            struct.pack("<q", int(c.get("id", 0))) + struct.pack("<b", int(c["semantic_label"])))


def array(items, enc):
    return struct.pack("<I", len(items)) + b"".join(enc(i) for i in items)


def string(s):
    s = s.encode() if isinstance(s, str) else bytes(s)
    return struct.pack("<I", len(s)) + s


def header(h):                                           # std_msgs/Header
    sec, nsec = h.get("stamp", (0, 0))
    return struct.pack("<3I", int(h.get("seq", 0)), int(sec), int(nsec)) + string(h.get("frame_id", ""))


def odometry(o):                                         # nav_msgs/Odometry
    return (header(o.get("header", {})) + string(o.get("child_frame_id", "")) + pose(o["pose"]) +
            struct.pack("<36d", *np.asarray(o.get("pose_covariance", np.zeros(36)), np.float64).reshape(36)) +
            struct.pack("<6d", *np.asarray(o.get("twist", np.zeros(6)), np.float64).reshape(6)) +
            struct.pack("<36d", *np.asarray(o.get("twist_covariance", np.zeros(36)), np.float64).reshape(36)))


def pose_mst(p):                                         # PoseMst.msg
    sec, nsec = p.get("stamp", (0, 0))
    return (pose(p["pose"]) + pose(p["relativeRawOdom"]) + struct.pack("<2I", int(sec), int(nsec)) +
            array(p.get("cubes", []), lambda b: box(b, "dim")) + array(p.get("cylinders", []), cylinder) +
            array(p.get("ellipsoids", []), lambda b: box(b, "scale")))


def bundle(b):                                           # PoseMstBundle.msg
    mp = np.asarray(b.get("map_of_labelXYZ", np.zeros((0, 7))), np.float64).reshape(-1, 7)
    return (struct.pack("<b", int(b["robotID"])) + array(b["poseMstPair"], pose_mst) +
            array(list(mp), lambda v: struct.pack("<7d", *v)) +
            array(b.get("interRobotTFs", []), lambda t: struct.pack("<2b", int(t["hostRobotID"]), int(t["targetRobotID"])) +
                  pose(t["TFfromTarget2Host"])))


def sync_odom(m):                                        # SemanticMeasSyncOdom.msg
    return (header(m.get("header", {})) + array(m.get("ellipsoid_factors", []), lambda b: box(b, "scale")) +
            array(m.get("cylinder_factors", []), cylinder) + array(m.get("cuboid_factors", []), lambda b: box(b, "dim")) +
            odometry(m["odometry"]))


def relative_meas(m):                                    # RelativeInterRobotMeasurementOdom.msg
    return (header(m.get("header", {})) + pose(m["relativePose"]) +
            struct.pack("<2b", int(m["robotIdObserver"]), int(m["robotIdObserved"])) + odometry(m["odometryObserver"]) +
            odometry(m["odometryObserved"]))


# ---- rosbag v2.0 writer (for reader tests) ------------------------------------------------------------------------------------
def _fields(d):
    out = b""
    for k, v in d.items():
        f = k.encode() + b"=" + v
        out += struct.pack("<I", len(f)) + f
    return out


def _record(hdr, data):
    h = _fields(hdr)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def write_bag(path, connections, messages, chunk_messages=3, compression="none"):
    """connections: {conn_id: (topic, datatype, md5sum)}; messages: [(conn_id, (sec, nsec), payload)] in file order.
    Layout as rosbag writes it: bag header record padded to 4096 bytes, chunks (connection record before a connection's first
    message, message records), one index-data record per connection after each chunk, then connection + chunk-info records."""
    body = b""
    seen = set()
    chunk_infos = []
    for c0 in range(0, max(len(messages), 1), chunk_messages):
        part = messages[c0:c0 + chunk_messages]
        chunk = b""
        index = {}
        for conn, (sec, nsec), payload in part:
            if conn not in seen:
                seen.add(conn)
                topic, dtype, md5 = connections[conn]
                chunk += _record({"op": b"\x07", "conn": struct.pack("<I", conn), "topic": topic.encode()},
                                 _fields({"topic": topic.encode(), "type": dtype.encode(), "md5sum": md5.encode(),
                                          "message_definition": b"# definition omitted"}))
            index.setdefault(conn, []).append((sec, nsec, len(chunk)))
            chunk += _record({"op": b"\x02", "conn": struct.pack("<I", conn), "time": struct.pack("<2I", sec, nsec)}, payload)
        pos = 13 + 4096 + len(body)
        stored = chunk
        if compression == "bz2":
            import bz2
            stored = bz2.compress(chunk)
        body += _record({"op": b"\x05", "compression": compression.encode(), "size": struct.pack("<I", len(chunk))}, stored)
        for conn, ent in index.items():
            body += _record({"op": b"\x04", "ver": struct.pack("<I", 1), "conn": struct.pack("<I", conn),
                             "count": struct.pack("<I", len(ent))}, b"".join(struct.pack("<3I", *e) for e in ent))
        if part:
            t0, t1 = min(p[1] for p in part), max(p[1] for p in part)
            chunk_infos.append((pos, t0, t1, {c: len(e) for c, e in index.items()}))
    index_pos = 13 + 4096 + len(body)
    tail = b""
    for conn, (topic, dtype, md5) in connections.items():
        tail += _record({"op": b"\x07", "conn": struct.pack("<I", conn), "topic": topic.encode()},
                        _fields({"topic": topic.encode(), "type": dtype.encode(), "md5sum": md5.encode(),
                                 "message_definition": b"# definition omitted"}))
    for pos, t0, t1, counts in chunk_infos:
        tail += _record({"op": b"\x06", "ver": struct.pack("<I", 1), "chunk_pos": struct.pack("<Q", pos),
                         "start_time": struct.pack("<2I", *t0), "end_time": struct.pack("<2I", *t1),
                         "count": struct.pack("<I", len(counts))}, b"".join(struct.pack("<2I", c, n) for c, n in counts.items()))
    bh = _fields({"op": b"\x03", "index_pos": struct.pack("<Q", index_pos), "conn_count": struct.pack("<I", len(connections)),
                  "chunk_count": struct.pack("<I", len(chunk_infos))})
    pad = 4096 - 4 - len(bh) - 4
    head = struct.pack("<I", len(bh)) + bh + struct.pack("<I", pad) + b" " * pad
    with open(path, "wb") as fh:
        fh.write(b"#ROSBAG V2.0\n" + head + body + tail)
