"""Association sweep timing (the bench's roofline.assoc leg alone): python tools/assoc_time.py  [SLIDE_ASSOC_THREADS=512]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slide_slam_amd as s
import bench
s.device_check()
r = bench.assoc_roofline(s)
r = bench.assoc_roofline(s)
print(json.dumps({k: r[k] for k in ("achieved", "frac", "avg_launch_ms", "frames_per_s", "matched_fraction")}), os.environ.get("SLIDE_ASSOC_THREADS"))
