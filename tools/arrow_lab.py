"""CPU lab: oracle shards (ingest-only build, as bench.py's workload) driven by PassDriver's exact joint step (arrow solve).
usage: arrow_lab.py <preset> <passes>"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle as po                                           # noqa: E402
from slide_slam_amd.distributed import PassDriver, setup_local_shards         # noqa: E402
from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world      # noqa: E402
from dist_worker import oracle_matcher                                        # noqa: E402
from pcg_lab import ingest                                                    # noqa: E402


def main():
    preset, passes = sys.argv[1], int(sys.argv[2])
    cfg = SynthConfig.preset(preset)
    wm = make_world(cfg)
    R, P = cfg.robots, cfg.poses_per_robot
    L = po.lib(native=True)
    shards = [po.OracleBackend(po.OrcParams.default(num_threads=8), 1, L=L) for _ in range(R)]
    for r, sh in enumerate(shards):
        ingest(sh, make_robot_log(cfg, wm, r))
    bufs, info = setup_local_shards(shards, oracle_matcher)
    print("slots", info["n_slots"], "separator dim", info["sep_dim"], flush=True)
    drv = PassDriver(shards, bufs, info["n_slots"], arrow=True, sep_dim=info["sep_dim"], sep_prof=info.get("sep_prof"))
    if len(sys.argv) > 3 and sys.argv[3] == "relmeas":
        from slide_slam_amd.synth import make_relmeas
        logs = [make_robot_log(cfg, wm, r) for r in range(R)]
        rel = make_relmeas(cfg, logs)
        print("relmeas", [(k, a, b, [round(float(v), 2) for v in r7[:3]]) for (k, a, b, r7) in rel], "ghost slots", drv.setup_ghosts(rel), flush=True)
    prev = None
    for p in range(passes):
        t0 = time.time()
        drv.one_pass()
        cur = np.array([[sh.graph.get_pose12(0, k)[1] for k in range(P)] for sh in shards])
        step = float(np.abs(cur - prev).max()) if prev is not None else float("nan")
        prev = cur
        print(f"pass {p + 1}: max pose step {step:.3e}  {time.time() - t0:.1f}s", flush=True)


if __name__ == "__main__":
    main()
