"""CPU lab for the joint solve of the sharded pass: oracle shards (ingest-only build, as bench.py's workload), PassDriver with the
PCG phases of the oracle, iterations to a tolerance per pass.  usage: pcg_lab.py <preset> <passes> <kmax> <tol> [robots]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle as po                                           # noqa: E402
from slide_slam_amd.distributed import PassDriver, setup_local_shards         # noqa: E402
from slide_slam_amd.synth import SynthConfig, frame_detections, make_robot_log, make_world      # noqa: E402
from dist_worker import oracle_matcher                                        # noqa: E402


def ingest(shard, log, mode=2):
    for k in range(len(log["rel7"])):
        shard.process_frame(0, log["rel7"][k], log["gt7"][k], frame_detections(log, k), mode)
    assert shard.ingest_solve() == 0


def main():  # noqa: C901
    preset, passes, kmax, tol = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    cfg = SynthConfig.preset(preset)
    wm = make_world(cfg)
    R, P = cfg.robots, cfg.poses_per_robot
    L = po.lib(native=True)
    t0 = time.time()
    shards = [po.OracleBackend(po.OrcParams.default(num_threads=8), 1, L=L) for _ in range(R)]
    for r, sh in enumerate(shards):
        ingest(sh, make_robot_log(cfg, wm, r))
    print("built", R, "shards in", time.time() - t0, flush=True)
    bufs, info = setup_local_shards(shards, oracle_matcher)
    print("slots", info["n_slots"], flush=True)
    drv = PassDriver(shards, bufs, info["n_slots"], pcg_iters=kmax, pcg_tol=tol)
    prev = None
    for p in range(passes):
        t0 = time.time()
        drv.one_pass()
        cur = np.array([[sh.graph.get_pose12(0, k)[1] for k in range(P)] for sh in shards])
        step = float(np.abs(cur - prev).max()) if prev is not None else float("nan")
        prev = cur
        st = shards[0].graph.pcg_stats()
        print(f"pass {p + 1}: pcg iterations {drv.pcg_history[-1] if drv.pcg_history else 0}  gamma {st['gamma_first']:.3e} -> {st['gamma_last']:.3e}  "
              f"max pose step {step:.3e}  {time.time() - t0:.1f}s", flush=True)


if __name__ == "__main__":
    main()
