import sys, numpy as np, faulthandler
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import slide_slam_amd as s
from tests.test_gpu_graph import _small_graph
which = sys.argv[1]
if which == "c1":
    g = s.SlideGraph(s.default_params(pose_chart=1)); _small_graph(g); print("solve", g.solve()); print(g.get_pose(0,3))
elif which == "two0":
    g = s.SlideGraph(s.default_params(pose_chart=0)); _small_graph(g); print("solve", g.solve())
    g2 = s.SlideGraph(s.default_params(pose_chart=0)); _small_graph(g2); print("solve2", g2.solve()); print(g2.get_pose(0,3))
elif which == "del0":
    g = s.SlideGraph(s.default_params(pose_chart=0)); _small_graph(g); print("solve", g.solve()); del g
    g2 = s.SlideGraph(s.default_params(pose_chart=0)); _small_graph(g2); print("solve2", g2.solve()); print(g2.get_pose(0,3))
elif which == "loop1":
    g = s.SlideGraph(s.default_params(pose_chart=1)); _small_graph(g)
    for it in range(4):
        print("solve", it, g.solve(), g.stats(), flush=True)
