# The association sweep of bench.py's roofline.assoc leg by itself (for rocprofv3 --kernel-trace --stats / --pmc runs):
#   python3 tools/assoc_sweep_prof.py [n_query] [repeats]
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slide_slam_amd as s
import bench
s.device_check()
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 5
print(json.dumps(bench.assoc_roofline(s, n_query=nq, repeats=rep)))
