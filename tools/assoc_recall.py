# Cross-robot association recall of a multi-robot preset as a function of the synthetic odometry noise: how many of the landmarks
# that two robots really both observed (ground-truth ids) end up as shared slots.  usage: assoc_recall.py <preset> <odom scale>:<detection noise scale> [..]
import dataclasses, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
import slide_slam_amd as s
from slide_slam_amd.distributed import gpu_matcher, setup_local_shards
from slide_slam_amd.replay import replay_single
from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world
preset = sys.argv[1]
for arg in sys.argv[2:]:
    sc, dsc = [float(v) for v in arg.split(":")]
    cfg = SynthConfig.preset(preset)
    cfg = dataclasses.replace(cfg, sigma_odom=tuple(v * sc for v in cfg.sigma_odom), sigma_det_pos=cfg.sigma_det_pos * dsc,
                              sigma_cube_yaw=cfg.sigma_cube_yaw * dsc, sigma_scale=cfg.sigma_scale * dsc)
    wm = make_world(cfg)
    logs = [make_robot_log(cfg, wm, r) for r in range(cfg.robots)]
    sets = [set(l["cyl_gt"]) | set(l["cube_gt"]) | set(l["ell_gt"]) for l in logs]
    cnt = {}
    for st in sets:
        for v in st: cnt[v] = cnt.get(v, 0) + 1
    true_shared = sum(1 for v in cnt.values() if v >= 2)
    shards, drift = [], []
    t0 = time.time()
    for lg in logs:
        gb = s.SlideBackend(s.default_params(), 1)
        out = replay_single(gb, lg)
        p = np.array(out["pose7"])
        drift.append(float(np.linalg.norm(p[:, :3] - lg["gt7"][:, :3], axis=1).max()))
        shards.append(gb)
    bufs, info = setup_local_shards(shards, gpu_matcher, device=torch.device("cuda", 0))
    c = [sh.counts() for sh in shards]
    print(f"{preset} odom x{sc} det x{dsc}: slots {info['n_slots']} / true shared {true_shared}; global inventory {sum(info['n_global'])} (gt union {len(cnt)}); "
          f"per-robot maps {[x['cyl'] + x['cube'] + x['point'] for x in c]} (gt {[len(x) for x in sets]}); max drift vs GT {max(drift):.2f} m; {time.time()-t0:.1f}s", flush=True)
