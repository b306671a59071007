"""A/B of the association sweep kernels at the headline sizes (10 k map, K = 1000, 20 detections, 8192 frames): ms per launch, matched
fraction, and a checksum of the matches (the kernels must agree id for id).  SLIDE_ASSOC_REG / SLIDE_ASSOC_THREADS select the kernel."""
import json, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slide_slam_amd as s
from slide_slam_amd.synth import assoc_sweep_case

cloud, model, label, qpos, obs, olab = assoc_sweep_case(2024, 10000, 20, 8192)
out, ms = s.assoc_sweep_batch(cloud, model, label, qpos, obs, olab, 1000, 0.75, repeats=10)
bpf = 12 * 10000 + 28 * 1000 + 36 * 20
per = ms / 10
print(json.dumps(dict(env={k: v for k, v in os.environ.items() if k.startswith("SLIDE_ASSOC")}, ms_per_launch=per,
                      frac=bpf * 8192 / (per * 1e-3) / 8e12, matched=float((out >= 0).mean()),
                      crc=zlib.crc32(np.ascontiguousarray(out).tobytes()))))
