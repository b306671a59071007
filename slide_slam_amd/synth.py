"""Seeded synthetic multi-robot semantic-SLAM frame logs.

The reference ships no processed bags (README.md:231-233 of the reference: Google-Drive
downloads), so every input of the hot path is generated here.  One *frame* is what
``SLOAMNode::runSLOAMNode`` (reference backend/sloam/src/core/sloamNode.cpp:762-768) receives:
the relative raw-odometry motion since the previous key pose and the body-frame detections
(cylinders / cuboids / ellipsoids), quantised to float32 exactly where the reference wire
types are float32 (backend/sloam_msgs/msg/ROSCube.msg:1, ROSCylinder.msg:1-4, ROSEllipsoid.msg:1).

Poses are ``pose7 = tx,ty,tz,qx,qy,qz,qw``.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, List

import numpy as np

MASTER_SEED = 0x51DE51A4

CLS_CYL, CLS_CUBE, CLS_ELL = 0, 1, 2
MULTI_ROBOT_NOISE = dict(sigma_odom=(0.0005, 0.0005, 0.0005, 0.001, 0.001, 0.001), sigma_det_pos=0.005, sigma_cube_yaw=0.0017,
                         sigma_scale=0.001)


@dataclasses.dataclass
class SynthConfig:
    name: str = "C2"
    robots: int = 1
    poses_per_robot: int = 500
    landmarks: int = 1000
    grid: tuple = (1, 1)            # robot cells (rows, cols)
    cell: float = 120.0             # cell edge [m]
    overlap: float = 0.0            # margin each robot sweeps into its neighbours [m]
    step: float = 0.8               # key-frame spacing (> min_odom_distance 0.5, params/sloam.yaml:11-12)
    sense_range: float = 15.0
    max_dets: int = 20
    min_spacing: float = 2.5        # > largest match threshold
    class_mix: tuple = (0.10, 0.20, 0.70)   # cylinder, cube, ellipsoid
    sigma_odom: tuple = (0.01, 0.01, 0.01, 0.02, 0.02, 0.02)  # [rot, trans] per metre
    sigma_det_pos: float = 0.05
    sigma_cube_yaw: float = 0.017
    sigma_scale: float = 0.01
    relmeas_every: int = 50
    seed: int = MASTER_SEED

    @staticmethod
    def preset(name: str) -> "SynthConfig":
        if name == "tiny":     # CPU-test size: oracle replays it in well under a second
            return SynthConfig(name="tiny", robots=1, poses_per_robot=40, landmarks=60, cell=40.0)
        if name == "small":
            return SynthConfig(name="small", robots=1, poses_per_robot=120, landmarks=220, cell=60.0)
        if name == "C2":       # BASELINE.json configs[1]
            return SynthConfig(name="C2", robots=1, poses_per_robot=500, landmarks=1000, cell=120.0)
        # Full-size multi-robot presets: odometry and detection noise of a LiDAR-inertial front end (0.03 deg/m, 0.1 % translation,
        # 5 mm object positions) instead of the yaml's noise-MODEL sigmas used as noise: with those, two robots' independently built
        # maps drift metres apart over 500 m (measured: 4.8 m), the cross-robot association (thresholds 0.75 - 2 m, TF known a
        # priori) finds 15 % of the landmarks both robots really observed, and a host replica ingesting a neighbour's packets
        # diverges.  With these values the merge recovers them (C3: 188 of 188; tools/assoc_recall.py).
        if name == "C3":       # configs[2]: 2 robots, 30 % shared landmarks
            return SynthConfig(name="C3", robots=2, poses_per_robot=500, landmarks=1700, grid=(1, 2), cell=102.0,
                               overlap=18.0, **MULTI_ROBOT_NOISE)
        if name == "C3spec":   # C3 with the noise SURVEY 8d writes down (the yaml's noise-model sigmas used as noise): 20 x / 10 x the above
            return SynthConfig(name="C3spec", robots=2, poses_per_robot=500, landmarks=1700, grid=(1, 2), cell=102.0, overlap=18.0)
        if name == "C3tiny":
            return SynthConfig(name="C3tiny", robots=2, poses_per_robot=40, landmarks=110, grid=(1, 2), cell=34.0,
                               overlap=8.0)
        if name == "C3rel":    # C3tiny with an inter-robot relative-pose measurement every 8 frames
            return SynthConfig(name="C3rel", robots=2, poses_per_robot=40, landmarks=110, grid=(1, 2), cell=34.0,
                               overlap=8.0, relmeas_every=8)
        if name == "C4tiny":   # four robots on a 2 x 2 grid: CPU-test size for several robots per process x several processes
            return SynthConfig(name="C4tiny", robots=4, poses_per_robot=30, landmarks=200, grid=(2, 2), cell=30.0,
                               overlap=8.0)
        if name == "C8tiny":   # eight robots on the 2 x 4 grid of C4 at CPU-test size, with inter-robot relative-pose measurements
            return SynthConfig(name="C8tiny", robots=8, poses_per_robot=30, landmarks=400, grid=(2, 4), cell=30.0,
                               overlap=8.0, relmeas_every=6)
        if name == "C4":       # configs[3]: 8 robots, 10 k landmarks, 5 k poses
            return SynthConfig(name="C4", robots=8, poses_per_robot=625, landmarks=10000, grid=(2, 4), cell=110.0,
                               overlap=15.0, **MULTI_ROBOT_NOISE)
        if name == "C4shard":  # one robot's share of C4 (bench N=1 workload)
            return SynthConfig(name="C4shard", robots=1, poses_per_robot=625, landmarks=1250, cell=110.0)
        raise ValueError(name)


# ----------------------------------------------------------------------------------------------
# small SO(3)/SE(3) helpers (numpy, generator side only)
# ----------------------------------------------------------------------------------------------
def rpy_to_R(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw), math.sin(yaw)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1.0, 0], [-sp, 0, cp]])
    Rx = np.array([[1.0, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def expmap_so3(w):
    th = float(np.linalg.norm(w))
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-12:
        return np.eye(3) + W
    return np.eye(3) + math.sin(th) / th * W + (1 - math.cos(th)) / (th * th) * (W @ W)


def R_to_quat(R):
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        w, x, y, z = 0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        w, x, y, z = (R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        w, x, y, z = (R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        w, x, y, z = (R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s
    if w < 0:
        w, x, y, z = -w, -x, -y, -z
    return np.array([x, y, z, w])


def quat_to_R(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def pose7(R, t):
    return np.concatenate([np.asarray(t, dtype=np.float64), R_to_quat(R)])


def pose7_to_Rt(p):
    return quat_to_R(np.asarray(p[3:7], dtype=np.float64)), np.asarray(p[0:3], dtype=np.float64)


def f32(x):
    return np.asarray(x, dtype=np.float32).astype(np.float64)


# ----------------------------------------------------------------------------------------------
def _poisson_points(rng, n, x0, x1, y0, y1, min_d):
    """n points in the rectangle with pairwise spacing >= min_d (grid-hashed dart throwing)."""
    cell = min_d
    grid: Dict[tuple, List[int]] = {}
    pts = np.zeros((n, 2))
    k = 0
    tries = 0
    while k < n:
        tries += 1
        if tries > 400 * n:
            raise RuntimeError("landmark density too high for the requested spacing")
        p = np.array([rng.uniform(x0, x1), rng.uniform(y0, y1)])
        gi, gj = int(math.floor(p[0] / cell)), int(math.floor(p[1] / cell))
        ok = True
        for a in (gi - 1, gi, gi + 1):
            for b in (gj - 1, gj, gj + 1):
                for idx in grid.get((a, b), ()):
                    if (pts[idx, 0] - p[0]) ** 2 + (pts[idx, 1] - p[1]) ** 2 < min_d * min_d:
                        ok = False
        if ok:
            pts[k] = p
            grid.setdefault((gi, gj), []).append(k)
            k += 1
    return pts


def make_world(cfg: SynthConfig):
    rng = np.random.default_rng(cfg.seed)
    rows, cols = cfg.grid
    W, H = cols * cfg.cell, rows * cfg.cell
    n = cfg.landmarks
    xy = _poisson_points(rng, n, 0.0, W, 0.0, H, cfg.min_spacing)
    z = rng.normal(0.0, 0.3, n)
    u = rng.uniform(0, 1, n)
    cls = np.where(u < cfg.class_mix[0], CLS_CYL, np.where(u < cfg.class_mix[0] + cfg.class_mix[1], CLS_CUBE, CLS_ELL))
    label = rng.integers(1, 7, n)
    world = dict(xyz=np.column_stack([xy, z]), cls=cls.astype(np.int32), label=label.astype(np.int32),
                 yaw=rng.uniform(-math.pi, math.pi, n),
                 cube_scale=rng.uniform(0.5, 3.0, (n, 3)), ell_scale=rng.uniform(0.3, 1.5, (n, 3)),
                 ray=np.column_stack([rng.normal(0, 0.02, n), rng.normal(0, 0.02, n), np.ones(n)]),
                 radius=rng.uniform(0.1, 0.4, n), extent=(W, H))
    return world


def make_trajectory(cfg: SynthConfig, robot: int, rng):
    """Boustrophedon sweep of the robot's cell (expanded by the overlap margin), 0.8 m steps."""
    rows, cols = cfg.grid
    r, c = divmod(robot, cols)
    W, H = cols * cfg.cell, rows * cfg.cell
    x0, x1 = max(c * cfg.cell - cfg.overlap, 0.0) + 4.0, min((c + 1) * cfg.cell + cfg.overlap, W) - 4.0
    y0, y1 = max(r * cfg.cell - cfg.overlap, 0.0) + 4.0, min((r + 1) * cfg.cell + cfg.overlap, H) - 4.0
    P = cfg.poses_per_robot
    total = P * cfg.step
    lane_len = x1 - x0
    n_lanes = max(1, int(math.ceil(total / lane_len)))
    lane_gap = (y1 - y0) / max(n_lanes, 1)
    pts = []
    s_left = 0.0
    lane, direction = 0, 1
    x, y = x0, y0 + 0.5 * lane_gap
    heading = 0.0
    turning = 0.0
    for _ in range(P):
        pts.append((x, y, heading))
        adv = cfg.step
        while adv > 1e-12:
            if turning > 0:          # moving up to the next lane
                d = min(adv, turning)
                y += d
                turning -= d
                adv -= d
                heading = math.pi / 2
                if turning <= 1e-12:
                    direction = -direction
            else:
                edge = x1 if direction > 0 else x0
                room = abs(edge - x)
                d = min(adv, room)
                x += direction * d
                adv -= d
                heading = 0.0 if direction > 0 else math.pi
                if room - d <= 1e-12:
                    lane += 1
                    turning = lane_gap
        s_left += cfg.step
    poses = []
    for (px, py, hd) in pts:
        R = rpy_to_R(rng.normal(0, 0.01), rng.normal(0, 0.01), hd + rng.normal(0, 0.01))
        poses.append((R, np.array([px, py, 2.0 + rng.normal(0, 0.02)])))
    return poses


def make_robot_log(cfg: SynthConfig, world, robot: int):
    rng = np.random.default_rng(cfg.seed + 1 + robot)
    traj = make_trajectory(cfg, robot, rng)
    P = len(traj)
    rel7 = np.zeros((P, 7))
    gt7 = np.zeros((P, 7))
    sig = np.asarray(cfg.sigma_odom)
    out = dict(cyl_off=[0], cube_off=[0], ell_off=[0], cyl_root=[], cyl_ray=[], cyl_radius=[], cyl_label=[], cyl_gt=[],
               cube_pose7=[], cube_scale=[], cube_label=[], cube_gt=[], ell_pose7=[], ell_scale=[], ell_label=[],
               ell_gt=[])
    xyz = world["xyz"]
    for k, (R, t) in enumerate(traj):
        gt7[k] = pose7(R, t)
        if k == 0:
            rel7[k] = pose7(R, t)           # prevKeyPose = identity for the first key frame (inputNode.cpp:163-169)
        else:
            Rp, tp = traj[k - 1]
            dR, dt = Rp.T @ R, Rp.T @ (t - tp)
            dist = max(float(np.linalg.norm(dt)), 1e-3)
            nz = rng.normal(0, 1, 6) * sig * dist
            rel7[k] = pose7(dR @ expmap_so3(nz[0:3]), dt + nz[3:6])
        # detections: all landmarks within range, nearest max_dets kept
        d = np.linalg.norm(xyz - t, axis=1)
        near = np.nonzero(d < cfg.sense_range)[0]
        near = near[np.argsort(d[near], kind="stable")][: cfg.max_dets]
        near = np.sort(near)
        for lm in near:
            pw = xyz[lm]
            pb = R.T @ (pw - t) + rng.normal(0, cfg.sigma_det_pos, 3)
            c = int(world["cls"][lm])
            if c == CLS_CYL:
                ray_w = world["ray"][lm]
                root_b = pb
                ray_b = R.T @ ray_w + rng.normal(0, 0.002, 3)
                out["cyl_root"].append(f32(root_b)); out["cyl_ray"].append(f32(ray_b))
                out["cyl_radius"].append(float(f32(world["radius"][lm] + rng.normal(0, 0.01))))
                out["cyl_label"].append(int(world["label"][lm])); out["cyl_gt"].append(int(lm))
            elif c == CLS_CUBE:
                Rw = rpy_to_R(0, 0, world["yaw"][lm] + rng.normal(0, cfg.sigma_cube_yaw))
                out["cube_pose7"].append(pose7(R.T @ Rw, pb))        # geometry_msgs/Pose: float64
                out["cube_scale"].append(f32(world["cube_scale"][lm] + rng.normal(0, cfg.sigma_scale, 3)))
                out["cube_label"].append(int(world["label"][lm])); out["cube_gt"].append(int(lm))
            else:
                out["ell_pose7"].append(pose7(R.T, pb))             # upright in the world frame
                out["ell_scale"].append(f32(world["ell_scale"][lm] + rng.normal(0, cfg.sigma_scale, 3)))
                out["ell_label"].append(int(world["label"][lm])); out["ell_gt"].append(int(lm))
        out["cyl_off"].append(len(out["cyl_gt"])); out["cube_off"].append(len(out["cube_gt"]))
        out["ell_off"].append(len(out["ell_gt"]))

    def arr(key, shape, dtype=np.float64):
        a = np.asarray(out[key], dtype=dtype)
        return a.reshape(shape) if a.size else np.zeros(shape if shape[0] != -1 else (0,) + tuple(shape[1:]), dtype=dtype)

    log = dict(rel7=rel7, gt7=gt7,
               cyl_off=np.asarray(out["cyl_off"], np.int64), cube_off=np.asarray(out["cube_off"], np.int64),
               ell_off=np.asarray(out["ell_off"], np.int64),
               cyl_root=arr("cyl_root", (-1, 3)), cyl_ray=arr("cyl_ray", (-1, 3)), cyl_radius=arr("cyl_radius", (-1,)),
               cyl_label=arr("cyl_label", (-1,), np.int32), cyl_gt=arr("cyl_gt", (-1,), np.int32),
               cube_pose7=arr("cube_pose7", (-1, 7)), cube_scale=arr("cube_scale", (-1, 3)),
               cube_label=arr("cube_label", (-1,), np.int32), cube_gt=arr("cube_gt", (-1,), np.int32),
               ell_pose7=arr("ell_pose7", (-1, 7)), ell_scale=arr("ell_scale", (-1, 3)),
               ell_label=arr("ell_label", (-1,), np.int32), ell_gt=arr("ell_gt", (-1,), np.int32))
    return log


def make_relmeas(cfg: SynthConfig, logs):
    """One relative-pose measurement per robot pair every ``relmeas_every`` frames
    (true relative pose + noise); consumed by addRelativeMeasFactor (graph.cpp:247-258)."""
    rng = np.random.default_rng(cfg.seed + 7777)
    out = []
    R_n = cfg.robots
    for k in range(cfg.relmeas_every, cfg.poses_per_robot, cfg.relmeas_every):
        for a in range(R_n):
            for b in range(a + 1, R_n):
                Ra, ta = pose7_to_Rt(logs[a]["gt7"][k]); Rb, tb = pose7_to_Rt(logs[b]["gt7"][k])
                dR, dt = Ra.T @ Rb, Ra.T @ (tb - ta)
                if np.linalg.norm(dt) > 60.0:
                    continue
                nz = rng.normal(0, 1, 6) * np.array([0.005, 0.005, 0.005, 0.02, 0.02, 0.02])
                out.append((k, a, b, pose7(dR @ expmap_so3(nz[0:3]), dt + nz[3:6])))
    return out


def make_relmeas_dense(cfg: SynthConfig, logs, every=50, max_range=40.0):
    """SURVEY 8d's density — one relative-pose measurement per ADJACENT robot pair per `every` frames while the observing robot is
    near the other's trajectory ("in the overlap") — regardless of simultaneity: pose ka of robot a is paired with the pose kb of robot b
    that is closest to it in space (the reference pairs by time stamp, sloam.cpp:321-412, so the two indices differ in general; the
    synthetic robots all sweep in lock step, which is why make_relmeas, pairing equal indices, finds only two on C4).
    Entries (ka, a, b, rel7 a@ka -> b@kb, kb); consumed by PassDriver.setup_ghosts / addRelativeMeasFactor (graph.cpp:247-258)."""
    rng = np.random.default_rng(cfg.seed + 8888)
    rows, cols = cfg.grid
    out = []
    pos = [np.array([pose7_to_Rt(g)[1] for g in lg["gt7"]]) for lg in logs]
    for a in range(cfg.robots):
        ra, ca = divmod(a, cols)
        for b in range(a + 1, cfg.robots):
            rb, cb = divmod(b, cols)
            if abs(ra - rb) + abs(ca - cb) != 1:
                continue
            for ka in range(every, cfg.poses_per_robot, every):
                d = np.linalg.norm(pos[b] - pos[a][ka], axis=1)
                kb = int(np.argmin(d))
                if d[kb] > max_range:
                    continue
                Ra, ta = pose7_to_Rt(logs[a]["gt7"][ka]); Rb, tb = pose7_to_Rt(logs[b]["gt7"][kb])
                dR, dt = Ra.T @ Rb, Ra.T @ (tb - ta)
                nz = rng.normal(0, 1, 6) * np.array([0.005, 0.005, 0.005, 0.02, 0.02, 0.02])
                out.append((ka, a, b, pose7(dR @ expmap_so3(nz[0:3]), dt + nz[3:6]), kb))
    return out


def relmeas_keys(e):
    """(ka, a, kb, b, rel7) of a relative-pose measurement entry: (k, a, b, rel7) pairs equal indices, (ka, a, b, rel7, kb) any two."""
    return (e[0], e[1], e[4] if len(e) > 4 else e[0], e[2], e[3])


def make_dataset(cfg: SynthConfig):
    world = make_world(cfg)
    logs = [make_robot_log(cfg, world, r) for r in range(cfg.robots)]
    rel = make_relmeas(cfg, logs) if cfg.robots > 1 else []
    return dict(cfg=cfg, world=world, logs=logs, relmeas=rel)


def frame_detections(log, k):
    """Slice frame k's detections out of a robot log."""
    a, b = int(log["cyl_off"][k]), int(log["cyl_off"][k + 1])
    c, d = int(log["cube_off"][k]), int(log["cube_off"][k + 1])
    e, f = int(log["ell_off"][k]), int(log["ell_off"][k + 1])
    return dict(cyl_root=log["cyl_root"][a:b], cyl_ray=log["cyl_ray"][a:b], cyl_radius=log["cyl_radius"][a:b],
                cyl_label=log["cyl_label"][a:b], cube_pose7=log["cube_pose7"][c:d], cube_scale=log["cube_scale"][c:d],
                cube_label=log["cube_label"][c:d], ell_pose7=log["ell_pose7"][e:f], ell_scale=log["ell_scale"][e:f],
                ell_label=log["ell_label"][e:f])


def assoc_sweep_case(seed: int, n_map: int = 10000, n_obs: int = 20, n_query: int = 5000):
    """Inputs of the batched association sweep at the headline sizes (BASELINE configs[3]): one resident map of `n_map` point landmarks
    on the C4 world (440 m x 220 m), `n_query` key frames, each at a uniformly drawn robot position with the `n_obs` NEAREST landmarks
    detected (5 cm ... 10 cm noise, SURVEY 8d's "all landmarks within reach, nearest 20 kept"), every seventh detection carrying a wrong
    label and one detection per frame displaced beyond the match threshold.  The parity test (tests/test_gpu_kernels.py::
    test_assoc_sweep_batch_matches_oracle) and bench.py's association leg draw from this ONE generator: the timed data is the tested data.
    Returns (cloud f32 [n_map, 3], model f64 [n_map, 3], label i32, qpos [n_query, 3], obs [n_query, n_obs, 3], olab i32 [n_query, n_obs])."""
    rng = np.random.default_rng(seed)
    model = np.column_stack([rng.uniform(0, 440, n_map), rng.uniform(0, 220, n_map), rng.normal(0, 0.3, n_map)])
    cloud = (model + rng.normal(0, 0.05, model.shape)).astype(np.float32)        # first-seen positions differ from the refined models
    label = rng.integers(1, 7, n_map).astype(np.int32)
    qpos = np.column_stack([rng.uniform(0, 440, n_query), rng.uniform(0, 220, n_query), np.full(n_query, 2.0)])
    obs = np.zeros((n_query, n_obs, 3))
    olab = np.zeros((n_query, n_obs), np.int32)
    for i in range(n_query):
        d2 = ((model[:, :2] - qpos[i, :2]) ** 2).sum(1)
        near = np.argpartition(d2, n_obs)[:n_obs]
        obs[i] = model[near] + rng.normal(0, 0.1, (n_obs, 3))
        olab[i] = label[near]
        olab[i, ::7] = (olab[i, ::7] % 6) + 1                                    # some detections carry the wrong label
        obs[i, 3] += 5.0                                                         # and one is off by more than the threshold
    return cloud, model, label, qpos, obs, olab
