# The SlideMatch / SlideGraph / CLIPPER legs of bench.py by themselves (for rocprofv3 --kernel-trace --stats):
#   python3 tools/place_prof.py [place|graph|all]
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slide_slam_amd as s
import bench
s.device_check()
what = sys.argv[1] if len(sys.argv) > 1 else "all"
out = {}
if what in ("place", "all"):
    out["place"] = bench.place_roofline(s, with_cpu=False)
if what in ("graph", "all"):
    out.update(bench.slidegraph_roofline(s, with_cpu=False))
print(json.dumps(out))
