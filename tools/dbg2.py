import sys, numpy as np
sys.path.insert(0, '.')
import slide_slam_amd as s
from oracle import pyoracle as po
from tests.test_gpu_graph import _small_graph
chart = int(sys.argv[1])
og = po.OracleGraph(po.OrcParams.default(pose_chart=chart))
gg = s.SlideGraph(s.default_params(pose_chart=chart))
_small_graph(og); _small_graph(gg)
for it in range(4):
    print("it", it, flush=True)
    print(" oracle", og.solve(), flush=True)
    print(" gpu", gg.solve(), flush=True)
    for k in range(4):
        a = og.get_pose12(0, k)[1]; b = gg.get_pose12(0, k)[1]
        print("  pose", k, np.abs(a-b).max(), flush=True)
