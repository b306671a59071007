"""Wire codec + rosbag reader (include/slide_wire.h) against the struct-based oracle and hand-written bytes.
Host code only: runs without a GPU."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import wire_oracle as wo          # noqa: E402  (the checker)
from slide_slam_amd import wire               # noqa: E402  (the product: ctypes over libslide_gpu.so)


def rpose(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return np.concatenate([rng.normal(size=3) * 10, q])


def rbox(rng, key):
    return {key: rng.uniform(0.2, 3, 3), "semantic_label": int(rng.integers(-3, 9)), "pose": rpose(rng)}


def rcyl(rng, nrad=0):
    return dict(root=rng.normal(size=3) * 5, ray=rng.normal(size=3), radii=rng.uniform(0.1, 0.4, nrad), radius=rng.uniform(0.1, 0.5),
                id=int(rng.integers(-2**40, 2**40)), semantic_label=int(rng.integers(0, 9)))


def rodom(rng):
    return dict(header=dict(seq=int(rng.integers(0, 1000)), stamp=(int(rng.integers(0, 2**31)), int(rng.integers(0, 10**9))),
                            frame_id="quadrotor/odom"), child_frame_id="quadrotor/base_link", pose=rpose(rng),
                pose_covariance=rng.normal(size=36), twist=rng.normal(size=6), twist_covariance=rng.normal(size=36))


def rbundle(rng, n_pm=3):
    return dict(robotID=int(rng.integers(0, 8)),
                poseMstPair=[dict(pose=rpose(rng), relativeRawOdom=rpose(rng), stamp=(int(rng.integers(0, 2**31)), int(rng.integers(0, 10**9))),
                                  cubes=[rbox(rng, "dim") for _ in range(rng.integers(0, 4))],
                                  cylinders=[rcyl(rng, int(rng.integers(0, 3))) for _ in range(rng.integers(0, 5))],
                                  ellipsoids=[rbox(rng, "scale") for _ in range(rng.integers(0, 4))]) for _ in range(n_pm)],
                map_of_labelXYZ=rng.normal(size=(int(rng.integers(0, 6)), 7)),
                interRobotTFs=[dict(hostRobotID=0, targetRobotID=int(t), TFfromTarget2Host=rpose(rng)) for t in range(1, int(rng.integers(1, 4)))])


def rsync(rng):
    return dict(header=dict(seq=7, stamp=(1700000000, 123456789), frame_id="quadrotor/base_link"),
                ellipsoid_factors=[rbox(rng, "scale") for _ in range(rng.integers(0, 6))],
                cylinder_factors=[rcyl(rng) for _ in range(rng.integers(0, 6))],
                cuboid_factors=[rbox(rng, "dim") for _ in range(rng.integers(0, 6))], odometry=rodom(rng))


def test_reference_stated_sizes():
    """Byte counts the reference itself states (PoseMst.msg comments; databaseManager.cpp:240-272): Pose 56, ROSCube / ROSEllipsoid
    69, vector7d 56, interRobotTF 58, stamp 8."""
    rng = np.random.default_rng(0)
    empty = dict(robotID=1, poseMstPair=[], map_of_labelXYZ=np.zeros((0, 7)), interRobotTFs=[])
    base = len(wire.encode_bundle(empty))
    assert base == 1 + 4 + 4 + 4
    one_pm = dict(empty, poseMstPair=[dict(pose=rpose(rng), relativeRawOdom=rpose(rng), stamp=(1, 2), cubes=[], cylinders=[], ellipsoids=[])])
    pm0 = len(wire.encode_bundle(one_pm)) - base
    assert pm0 == 56 + 56 + 8 + 3 * 4
    for key, fld in (("cubes", "dim"), ("ellipsoids", "scale")):
        m = dict(one_pm, poseMstPair=[dict(one_pm["poseMstPair"][0], **{key: [rbox(rng, fld)]})])
        assert len(wire.encode_bundle(m)) - base - pm0 == 69
    assert len(wire.encode_bundle(dict(empty, map_of_labelXYZ=np.ones((1, 7))))) - base == 56
    assert len(wire.encode_bundle(dict(empty, interRobotTFs=[dict(hostRobotID=0, targetRobotID=1, TFfromTarget2Host=rpose(rng))]))) - base == 58
    m = dict(one_pm, poseMstPair=[dict(one_pm["poseMstPair"][0], cylinders=[rcyl(rng, 0)])])
    assert len(wire.encode_bundle(m)) - base - pm0 == 37 + 4       # the 37 bytes of the .msg comment + the radii[] count


def test_known_answer_bytes():
    """Hand-written little-endian bytes of a PoseMstBundle with one interRobotTF."""
    msg = dict(robotID=2, poseMstPair=[], map_of_labelXYZ=np.zeros((0, 7)),
               interRobotTFs=[dict(hostRobotID=0, targetRobotID=1, TFfromTarget2Host=[1.0, 2.0, 3.0, 0.0, 0.0, 0.0, 1.0])])
    expect = bytes.fromhex("02" "00000000" "00000000" "01000000" "00" "01"
                           "000000000000f03f" "0000000000000040" "0000000000000840"
                           "0000000000000000" "0000000000000000" "0000000000000000" "000000000000f03f")
    assert wire.encode_bundle(msg) == expect
    back = wire.decode_bundle(expect)
    assert back["robotID"] == 2 and back["interRobotTFs"][0]["targetRobotID"] == 1
    assert np.array_equal(back["interRobotTFs"][0]["TFfromTarget2Host"], [1, 2, 3, 0, 0, 0, 1])
    # float32 quantisation + negative int8 label + int64 id of a cylinder inside a SemanticMeasSyncOdom
    cyl = dict(root=[0.1, 0.2, 0.3], ray=[0.0, 0.0, 1.0], radii=[], radius=0.25, id=-2, semantic_label=-1)
    m = dict(header=dict(seq=1, stamp=(2, 3), frame_id="ab"), ellipsoid_factors=[], cylinder_factors=[cyl], cuboid_factors=[],
             odometry=dict(header=dict(seq=0, stamp=(0, 0), frame_id=""), child_frame_id="", pose=[0, 0, 0, 0, 0, 0, 1]))
    b = wire.encode_sync_odom(m)
    head = bytes.fromhex("01000000" "02000000" "03000000" "02000000" "6162" "00000000" "01000000")
    assert b[:len(head)] == head
    body = b[len(head):len(head) + 41]
    assert body == (struct.pack("<3f", np.float32(0.1), np.float32(0.2), np.float32(0.3)) + bytes.fromhex("00000000" "00000000" "0000803f")
                    + bytes.fromhex("00000000") + bytes.fromhex("0000803e") + bytes.fromhex("feffffffffffffff") + b"\xff")
    assert struct.pack("<f", np.float32(0.1)) == bytes.fromhex("cdcccc3d")


@pytest.mark.parametrize("seed", range(6))
def test_encode_matches_oracle_and_round_trips(seed):
    rng = np.random.default_rng(seed)
    b = rbundle(rng, n_pm=int(rng.integers(0, 5)))
    enc = wire.encode_bundle(b)
    assert enc == wo.bundle(b)
    d = wire.decode_bundle(enc)
    assert wo.bundle(d) == enc                           # decode loses nothing that is on the wire
    assert d["robotID"] == b["robotID"] and len(d["poseMstPair"]) == len(b["poseMstPair"])
    for p, q in zip(b["poseMstPair"], d["poseMstPair"]):
        assert np.array_equal(q["pose"], p["pose"]) and q["stamp"] == p["stamp"]
        for x, y in zip(p["cubes"], q["cubes"]):
            assert np.array_equal(y["dim"], np.asarray(x["dim"], np.float32)) and y["semantic_label"] == x["semantic_label"]
        for x, y in zip(p["cylinders"], q["cylinders"]):
            assert y["radius"] == np.float32(x["radius"]) and y["id"] == x["id"] and np.array_equal(y["radii"], x["radii"])
    s = rsync(rng)
    enc = wire.encode_sync_odom(s)
    assert enc == wo.sync_odom(s)
    assert wo.sync_odom(wire.decode_sync_odom(enc)) == enc
    r = dict(header=s["header"], relativePose=rpose(rng), robotIdObserver=0, robotIdObserved=3, odometryObserver=rodom(rng),
             odometryObserved=rodom(rng))
    enc = wire.encode_relative_meas(r)
    assert enc == wo.relative_meas(r)
    d = wire.decode_relative_meas(enc)
    assert wo.relative_meas(d) == enc and d["robotIdObserved"] == 3 and d["odometryObserver"]["header"]["frame_id"] == "quadrotor/odom"


def test_malformed_buffers_are_rejected():
    rng = np.random.default_rng(3)
    enc = wire.encode_sync_odom(rsync(rng))
    for bad in (enc[:-1], enc[:10], enc + b"\0", b""):
        with pytest.raises(wire.WireError) as e:
            wire.decode_sync_odom(bad)
        assert e.value.code == -1
    # an element count the buffer cannot hold must not allocate or read past the end
    huge = struct.pack("<b", 0) + struct.pack("<I", 0xFFFFFFF0) + b"\0" * 64
    with pytest.raises(wire.WireError):
        wire.decode_bundle(huge)


def test_sync_odom_to_frame_matches_the_reference_converter():
    """Robot::RobotObservationCb / rosCylinder2CylinderObj / rosEllipsoid2EllipObj (robot.cpp:100-203): float32 fields widen to
    double, poses and labels pass through, odometry pose = odometry.pose.pose."""
    rng = np.random.default_rng(5)
    s = rsync(rng)
    while not (s["cylinder_factors"] and s["cuboid_factors"] and s["ellipsoid_factors"]):
        s = rsync(rng)
    pose7, det, hdr = wire.sync_odom_to_frame(wo.sync_odom(s))
    assert np.array_equal(pose7, s["odometry"]["pose"]) and hdr["stamp"] == (1700000000, 123456789)
    for i, c in enumerate(s["cylinder_factors"]):
        assert np.array_equal(det["cyl_root"][i], np.asarray(c["root"], np.float32).astype(np.float64))
        assert np.array_equal(det["cyl_ray"][i], np.asarray(c["ray"], np.float32).astype(np.float64))
        assert det["cyl_radius"][i] == float(np.float32(c["radius"])) and det["cyl_label"][i] == c["semantic_label"]
    for key, pre, fld in (("cuboid_factors", "cube", "dim"), ("ellipsoid_factors", "ell", "scale")):
        for i, b in enumerate(s[key]):
            assert np.array_equal(det[pre + "_pose7"][i], b["pose"])
            assert np.array_equal(det[pre + "_scale"][i], np.asarray(b[fld], np.float32).astype(np.float64))
            assert det[pre + "_label"][i] == b["semantic_label"]


def test_bag_reader(tmp_path):
    rng = np.random.default_rng(9)
    conns = {0: ("/robot0/semantic_meas_sync_odom", "sloam_msgs/SemanticMeasSyncOdom", "a" * 32),
             1: ("/robot0/pose_mst_bundle", "sloam_msgs/PoseMstBundle", "b" * 32)}
    msgs = []
    for k in range(8):
        s = rsync(rng)
        msgs.append((0, (100 + k, 5 * k), wo.sync_odom(s)))
        if k % 3 == 0:
            msgs.append((1, (100 + k, 5 * k + 1), wo.bundle(rbundle(rng, 2))))
    # file order differs from time order: the reader returns play order (receive time, then file order)
    shuffled = [msgs[i] for i in (2, 0, 1, 3, 5, 4, 6, 7, 9, 8, 10)]
    path = tmp_path / "t.bag"
    wo.write_bag(str(path), conns, shuffled, chunk_messages=4)
    with wire.Bag(path) as bag:
        assert {c["topic"] for c in bag.connections.values()} == {c[0] for c in conns.values()}
        assert bag.connections[0]["datatype"] == "sloam_msgs/SemanticMeasSyncOdom" and bag.connections[1]["md5sum"] == "b" * 32
        got = list(bag)
    assert len(got) == len(msgs)
    want = sorted(shuffled, key=lambda m: m[1])
    for (topic, dtype, stamp, payload), (conn, t, data) in zip(got, want):
        assert topic == conns[conn][0] and dtype == conns[conn][1] and stamp == t and payload == data
    n_frames = 0
    for topic, dtype, stamp, payload in got:
        if dtype == "sloam_msgs/SemanticMeasSyncOdom":
            pose7, det, hdr = wire.sync_odom_to_frame(payload)
            n_frames += 1
            assert det["cyl_root"].shape[1] == 3 and pose7.shape == (7,)
    assert n_frames == 8
    # compressed chunks and foreign files are refused, not misread
    wo.write_bag(str(tmp_path / "c.bag"), conns, shuffled, compression="bz2")
    with pytest.raises(wire.WireError) as e:
        wire.Bag(tmp_path / "c.bag")
    assert e.value.code == -5
    # ... and readable after inflate_bag (the `rosbag decompress` step, host side)
    wire.inflate_bag(tmp_path / "c.bag", tmp_path / "ci.bag")
    with wire.Bag(tmp_path / "ci.bag") as bag:
        again = list(bag)
    assert [(t, s_, p) for t, _, s_, p in again] == [(t, s_, p) for t, _, s_, p in got]
    (tmp_path / "x.bag").write_bytes(b"#ROSBAG V1.2\n" + b"\0" * 100)
    with pytest.raises(wire.WireError):
        wire.Bag(tmp_path / "x.bag")
    trunc = path.read_bytes()[:-7]
    (tmp_path / "t2.bag").write_bytes(trunc)
    with pytest.raises(wire.WireError):
        wire.Bag(tmp_path / "t2.bag")
