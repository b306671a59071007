import sys, numpy as np
sys.path.insert(0, '.')
import slide_slam_amd as s
from slide_slam_amd.replay import replay_single
from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world
cfg = SynthConfig.preset(sys.argv[1] if len(sys.argv) > 1 else "C4")
log = make_robot_log(cfg, make_world(cfg), 0)
gb = s.SlideBackend(s.default_params(), 1)
replay_single(gb, log, robot=0, collect=False)
g = gb.graph
P = cfg.poses_per_robot
poses = lambda: np.array([g.get_pose12(0, k)[1] for k in range(0, P, 5)])
p = poses()
for it in range(14):
    g.gauss_newton(1)
    q = poses()
    print(it, "max pose change %.3e" % np.abs(q - p).max())
    p = q
