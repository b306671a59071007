"""Several robot shards in one process (ThreadGroup / ThreadComm, slide_slam_amd/distributed.py): the 8 / N-robots-per-GPU
layout of BASELINE's "8-robot graph at 1/2/4/8 GPUs".  CPU: oracle shards in threads; GPU: HIP shards on concurrent streams."""
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from slide_slam_amd.distributed import DistributedGraph, ThreadGroup      # noqa: E402
from slide_slam_amd.replay import replay_single                           # noqa: E402
from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world   # noqa: E402
from test_distributed import _check, _joint_optimum                       # noqa: E402
from dist_worker import gpu_matcher, oracle_matcher                       # noqa: E402


def _run_threads(make_shard, matcher, preset, iters, device=None, batch=None, local_pass=False, frames=None):
    cfg = SynthConfig.preset(preset)
    world_map = make_world(cfg)
    R = cfg.robots
    group = ThreadGroup(R)
    out, err, bufs = [None] * R, [], [None] * R

    def work(t):
        try:
            shard = make_shard()
            log = make_robot_log(cfg, world_map, t)
            nf = frames[t] if frames else len(log["rel7"])
            replay_single(shard, log, robot=0, n_frames=nf, collect=False)
            dg = DistributedGraph(shard, group.comm(t, device), t, R)
            info = dg.setup(matcher)
            if batch is not None:
                shard.graph.join_chol_batch(batch, t)
                dg.local_batch = local_pass == "local"
                bufs[t] = dg.buf
            if local_pass == "one-driver":
                group.barrier.wait()
                if t == 0:                       # one thread replays the captured pass of all robots
                    for _ in range(iters):
                        batch.pass_all([b.data_ptr() for b in bufs])
                group.barrier.wait()
            else:
                dg.gauss_newton(iters)
            if batch is not None:
                shard.graph.join_chol_batch(None)
            out[t] = (np.array([shard.graph.get_pose12(0, k)[1] for k in range(nf)]), info)
        except BaseException as e:      # a dead thread would leave the others at the barrier
            err.append(e)
            group.barrier.abort()

    th = [threading.Thread(target=work, args=(t,)) for t in range(R)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not err, err
    if frames:      # robots of different sizes: one array per robot
        return dict(n_slots=out[0][1]["n_slots"], **{f"poses{t}": out[t][0] for t in range(R)})
    return dict(poses=np.array([o[0] for o in out]), n_slots=out[0][1]["n_slots"], n_global=np.array(out[0][1]["n_global"]))


def test_thread_shards_oracle():
    from oracle import pyoracle as po
    joint, counts = _joint_optimum("C3tiny")
    z = _run_threads(lambda: po.OracleBackend(po.OrcParams.default(), 1), oracle_matcher, "C3tiny", 60)
    _check(z, joint, counts, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["streams", "batch", "batch-local", "batch-pass"])
def test_thread_shards_gpu(gpu, tmp_path, mode):
    """HIP shards of two robots on the one GPU: on concurrent streams, or with their dense factor + solve batched into one launch
    sequence (slide_chol_batch_*).  Runs in a fresh process: torch has to initialise the
    device before this library's HIP runtime is loaded (the other order leaves torch without a GPU in this image)."""
    import subprocess
    joint, counts = _joint_optimum("C3tiny")
    out = str(tmp_path / "t.npz")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), out, mode], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    _check(np.load(out), joint, counts, 1e-4)


@pytest.mark.gpu
def test_batched_pass_with_robots_of_different_size(gpu, tmp_path):
    """The batched kernels take every grid from the largest robot: robots with 40 and 27 poses (4 and 3 block columns) through the
    one-graph pass give what independent streams give."""
    import subprocess
    outs = []
    for mode in ("streams", "batch-pass"):
        out = str(tmp_path / f"{mode}.npz")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), out, mode, "40,27"], cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert int(a["n_slots"]) == int(b["n_slots"]) > 0
    for t in range(2):
        assert a[f"poses{t}"].shape == b[f"poses{t}"].shape and a[f"poses{t}"].shape[0] in (40, 27)
        assert np.abs(a[f"poses{t}"] - b[f"poses{t}"]).max() < 1e-8


if __name__ == "__main__":
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    import slide_slam_amd as s
    s.device_check()
    mode = sys.argv[2] if len(sys.argv) > 2 else "streams"
    z = _run_threads(lambda: s.SlideBackend(s.default_params(), 1), gpu_matcher, "C3tiny", 60, device=torch.device("cuda", 0),
                     batch=s.CholBatch(2) if mode != "streams" else None,
                     local_pass={"batch-local": "local", "batch-pass": "one-driver"}.get(mode, False),
                     frames=[int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None)
    np.savez(sys.argv[1], **z)
