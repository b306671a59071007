import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import slide_slam_amd as s
from slide_slam_amd.synth import SynthConfig, make_dataset
from slide_slam_amd.replay import replay_single
data = make_dataset(SynthConfig.preset("tiny"))
log = data["logs"][0]
def free(): 
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
f0 = None
for it in range(60):
    gb = s.SlideBackend(s.default_params(), 1)
    replay_single(gb, log, collect=False)
    del gb
    if it == 5: f0 = free()
f1 = free()
print("free after 5 / 60 create-replay-destroy cycles (MiB):", f0 >> 20, f1 >> 20, "delta", (f0 - f1) >> 20)
import resource; print("host RSS MiB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024)
