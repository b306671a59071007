"""ROS-free replay harness: feeds a synthetic frame log through a backend object in the order
``SLOAMNode::runSLOAMNode`` (reference sloamNode.cpp:762-1036) and its caller
``InputManager::RunInputNode`` (inputNode.cpp:158-181) would.

``backend`` is any object exposing ``process_frame(robot, rel7, prev7, det, mode)``,
``ingest_solve()``, ``end_frame(robot)`` and ``graph`` (the product's ``SlideBackend`` or, in
tests and the CPU-baseline leg only, the oracle wrapper).
"""
from __future__ import annotations

import time

import numpy as np

from .synth import frame_detections, pose7_to_Rt, pose7

IDENT7 = np.array([0, 0, 0, 0, 0, 0, 1.0])


def _compose7(a, b):
    Ra, ta = pose7_to_Rt(a)
    Rb, tb = pose7_to_Rt(b)
    return pose7(Ra @ Rb, Ra @ tb + ta)


def replay_single(backend, log, robot=0, n_frames=None, collect=True):
    """Single-robot replay: every frame = associate + add factors + solve (one pose-graph update)."""
    P = len(log["rel7"]) if n_frames is None else n_frames
    prev = IDENT7.copy()
    out = dict(pose7=[], cyl_id=[], cube_id=[], ell_id=[], t_frame=[])
    for k in range(P):
        det = frame_detections(log, k)
        t0 = time.perf_counter()
        r = backend.process_frame(robot, log["rel7"][k], prev, det, 0)
        t1 = time.perf_counter()
        if r["status"] != 0:
            raise RuntimeError(f"solve failed at frame {k}: status {r['status']}")
        prev = r["pose7"].copy()
        if collect:
            out["pose7"].append(prev.copy())
            out["cyl_id"].append(r["cyl_id"].copy()); out["cube_id"].append(r["cube_id"].copy())
            out["ell_id"].append(r["ell_id"].copy())
        out["t_frame"].append(t1 - t0)
    return out


def foreign_key_poses(own_backend, log):
    """What a robot's OWN sloam node publishes for each key frame (PoseMstPair.keyPose, sloamNode.cpp:793-800):
    poseEstimate = previous optimised key pose * relative raw odometry."""
    out = replay_single(own_backend, log, robot=0)
    P = len(log["rel7"])
    kp = [np.asarray(log["rel7"][0], dtype=np.float64).copy()]
    for k in range(1, P):
        kp.append(_compose7(out["pose7"][k - 1], log["rel7"][k]))
    return kp


def replay_multi(backend, data, host=0, n_frames=None, own_node_factory=None):
    """Multi-robot replay on ONE host graph (the reference's per-host replica): per time step the host
    robot's own frame (add + solve), then every other robot's packet of that step is ingested
    (sloamNode.cpp:912-1002: associate against the host's maps, add, one solve per robot), then the
    map refresh + current-pose fetch (:1010-1014).  Inter-robot TFs are known a priori (identity:
    all odometry is expressed in the common world frame, databaseManager.cpp:22-45 priorTFKnown).
    The foreign key poses are the ones each robot's own node would publish; `own_node_factory()` builds such a
    node (a fresh backend of the same kind)."""
    cfg = data["cfg"]
    logs = data["logs"]
    R = cfg.robots
    P = cfg.poses_per_robot if n_frames is None else n_frames
    prev = [IDENT7.copy() for _ in range(R)]
    rel_by_step = {}
    for (k, a, b, rel) in data["relmeas"]:
        rel_by_step.setdefault(k, []).append((a, b, rel))
    kposes = {}
    for o in range(R):
        if o != host:
            if own_node_factory is None:
                raise ValueError("replay_multi needs own_node_factory to produce the foreign robots' key poses")
            kposes[o] = foreign_key_poses(own_node_factory(), logs[o])
    out = dict(host_pose7=[], ids=[])
    for k in range(P):
        det = frame_detections(logs[host], k)
        r = backend.process_frame(host, logs[host]["rel7"][k], prev[host], det, 1)
        if r["status"] != 0:
            raise RuntimeError(f"host solve failed at step {k}")
        ids = [(r["cyl_id"].copy(), r["cube_id"].copy(), r["ell_id"].copy())]
        for o in range(R):
            if o == host:
                continue
            ro = backend.process_frame(o, logs[o]["rel7"][k], kposes[o][k], frame_detections(logs[o], k), 2)
            ids.append((ro["cyl_id"].copy(), ro["cube_id"].copy(), ro["ell_id"].copy()))
            st = backend.ingest_solve()
            if st != 0:
                raise RuntimeError(f"ingest solve failed at step {k} robot {o}")
        for (a, b, rel) in rel_by_step.get(k, []):
            backend.graph.add_relative_meas(rel, k, a, k, b)   # queued; consumed by the next solve()
        st, pose = backend.end_frame(host)
        prev[host] = pose.copy()
        out["host_pose7"].append(pose.copy())
        out["ids"].append(ids)
    return out


def _inverse7(a):
    Ra, ta = pose7_to_Rt(a)
    return pose7(Ra.T, -Ra.T @ ta)


def replay_bag(backend, bag_path, robot=0, topic=None, n_frames=None, collect=True):
    """ROS-free replay of a processed bag (rosbag v2.0, include/slide_wire.h): every `sloam_msgs/SemanticMeasSyncOdom` message
    (optionally only those on `topic`) becomes one frame, converted as Robot::RobotObservationCb does (robot.cpp:100-137); the
    relative raw odometry between consecutive observations is what SLOAMNode::runSLOAMNode feeds the graph
    (sloamNode.cpp:770-800: relativeMotion = prevOdom^-1 * currOdom)."""
    from . import wire
    prev_est = IDENT7.copy()
    prev_odom = None
    out = dict(pose7=[], cyl_id=[], cube_id=[], ell_id=[], t_frame=[], stamp=[])
    with wire.Bag(bag_path) as bag:
        for tpc, dtype, _, payload in bag:
            if dtype != "sloam_msgs/SemanticMeasSyncOdom" or (topic is not None and tpc != topic):
                continue
            if n_frames is not None and len(out["t_frame"]) >= n_frames:
                break
            odom7, det, hdr = wire.sync_odom_to_frame(payload)
            rel = odom7 if prev_odom is None else _compose7(_inverse7(prev_odom), odom7)
            t0 = time.perf_counter()
            r = backend.process_frame(robot, rel, prev_est, det, 0)
            t1 = time.perf_counter()
            if r["status"] != 0:
                raise RuntimeError(f"solve failed at frame {len(out['t_frame'])}: status {r['status']}")
            prev_est = r["pose7"].copy()
            prev_odom = odom7
            out["t_frame"].append(t1 - t0)
            out["stamp"].append(hdr["stamp"])
            if collect:
                out["pose7"].append(prev_est.copy())
                out["cyl_id"].append(r["cyl_id"].copy()); out["cube_id"].append(r["cube_id"].copy())
                out["ell_id"].append(r["ell_id"].copy())
    return out


def log_to_sync_odom_messages(log, n_frames=None, t0=1700000000, dt=0.5, frame_id="quadrotor/base_link"):
    """A synthetic robot log as the message stream its process node would publish: one SemanticMeasSyncOdom dict per frame with
    the integrated raw odometry (body-frame objects as they are in the log)."""
    P = len(log["rel7"]) if n_frames is None else n_frames
    odom = IDENT7.copy()
    msgs = []
    for k in range(P):
        odom = np.asarray(log["rel7"][k], np.float64).copy() if k == 0 else _compose7(odom, log["rel7"][k])
        d = frame_detections(log, k)
        t = t0 + k * dt
        msgs.append(dict(
            header=dict(seq=k, stamp=(int(t), int(round((t - int(t)) * 1e9))), frame_id=frame_id),
            ellipsoid_factors=[dict(scale=d["ell_scale"][i], semantic_label=int(d["ell_label"][i]), pose=d["ell_pose7"][i])
                               for i in range(len(d["ell_label"]))],
            cylinder_factors=[dict(root=d["cyl_root"][i], ray=d["cyl_ray"][i], radii=[], radius=d["cyl_radius"][i], id=0,
                                   semantic_label=int(d["cyl_label"][i])) for i in range(len(d["cyl_label"]))],
            cuboid_factors=[dict(dim=d["cube_scale"][i], semantic_label=int(d["cube_label"][i]), pose=d["cube_pose7"][i])
                            for i in range(len(d["cube_label"]))],
            odometry=dict(header=dict(seq=k, stamp=(int(t), int(round((t - int(t)) * 1e9))), frame_id="world"),
                          child_frame_id=frame_id, pose=odom)))
    return msgs
