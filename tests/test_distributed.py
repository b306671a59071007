"""Multi-process tests of the one-robot-per-rank path (world_size 2, gloo): the distributed Gauss-Newton pass
(shared-landmark all-reduce, per-robot Schur solves) converges to the optimum of the JOINT graph that a single
host replica (the reference's arrangement) optimises."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import pyoracle as po
from slide_slam_amd.replay import replay_multi
from slide_slam_amd.synth import SynthConfig, make_dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from chart_env import chart_kw  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_workers(backend, preset, iters, out, world=2, extra=()):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), backend, preset, str(iters), out,
           *extra]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return np.load(out)


def _joint_optimum(preset, gn_iters=25, relmeas=False):
    """One host replica holding both robots (replay_multi = sloamNode.cpp:912-1002 order), then batch GN."""
    cfg = SynthConfig.preset(preset)
    data = make_dataset(cfg)
    rel = data["relmeas"]
    data["relmeas"] = []        # streaming without them: the shards associate without them too (same factor graph)
    ob = po.OracleBackend(po.OrcParams.default(**chart_kw(po)), cfg.robots)
    replay_multi(ob, data, own_node_factory=lambda: po.OracleBackend(po.OrcParams.default(**chart_kw(po)), 1))
    if relmeas:
        from slide_slam_amd.synth import relmeas_keys
        if not isinstance(relmeas, bool):
            rel = relmeas                                   # the caller's own list (e.g. make_relmeas_dense: two different pose indices)
        assert len(rel) > 0
        for (ka, a, kb, b, r7) in map(relmeas_keys, rel):
            ob.graph.add_relative_meas(r7, ka, a, kb, b)    # ordinary Between factors of the joint replica
    ob.graph.set_relin_threshold(0.0)
    for _ in range(gn_iters):
        assert ob.graph.solve() == 0
    P = cfg.poses_per_robot
    poses = np.array([[ob.graph.get_pose12(r, k)[1] for k in range(P)] for r in range(cfg.robots)])
    return poses, ob.counts()


def _check(z, joint, counts, tol):
    assert int(z["n_slots"]) > 0, "the two robots must share landmarks for this test to mean anything"
    # same landmark inventory as the joint replica: the cross-robot association found the same physical landmarks
    assert list(z["n_global"]) == [counts["cyl"], counts["cube"], counts["point"]]
    d = z["poses"]
    assert d.shape == joint.shape
    rel = np.linalg.norm((d - joint).reshape(d.shape[0], -1), axis=1) / np.linalg.norm(joint.reshape(d.shape[0], -1), axis=1)
    assert rel.max() < tol, rel


def test_distributed_gn_oracle_shards_gloo(tmp_path):
    """CPU: oracle shards + gloo.  Covers the N > 1 orchestration (association merge, slot tables, buffer
    layouts, two all-reduces per pass) without a GPU."""
    joint, counts = _joint_optimum("C3tiny")
    z = _run_workers("oracle", "C3tiny", 60, str(tmp_path / "o.npz"))
    _check(z, joint, counts, 1e-5)     # block-Jacobi over robots converges linearly; 1e-4 is the north-star bar


@pytest.mark.gpu
def test_distributed_gn_gpu_shards(tmp_path, gpu):
    """GPU: the HIP shards run the same phases (two ranks share the one visible GPU; gloo staged through the
    host stands in for RCCL, which needs one GPU per rank)."""
    joint, counts = _joint_optimum("C3tiny")
    z = _run_workers("gpu", "C3tiny", 60, str(tmp_path / "g.npz"))
    _check(z, joint, counts, 1e-4)


def test_distributed_relmeas_oracle_shards_gloo(tmp_path):
    """CPU: inter-robot relative-pose factors sharded with ghost poses (one extra all-reduce of 12 doubles per ghost slot
    per pass) reach the optimum of the joint replica that holds them as ordinary Between factors."""
    joint, counts = _joint_optimum("C3rel", relmeas=True)
    plain, _ = _joint_optimum("C3rel", relmeas=False)
    assert np.abs(joint - plain).max() > 1e-6            # the factors do move the optimum: the test has teeth
    z = _run_workers("oracle", "C3rel", 120, str(tmp_path / "or.npz"), extra=("relmeas",))
    assert int(z["n_gslots"]) > 0
    _check(z, joint, counts, 1e-5)


@pytest.mark.gpu
def test_distributed_relmeas_gpu_shards(tmp_path, gpu):
    joint, counts = _joint_optimum("C3rel", relmeas=True)
    z = _run_workers("gpu", "C3rel", 120, str(tmp_path / "gr.npz"), extra=("relmeas",))
    assert int(z["n_gslots"]) > 0
    _check(z, joint, counts, 1e-4)


def test_threads_times_processes_oracle_shards_gloo(tmp_path):
    """CPU: 2 processes x 2 robot shards each (ThreadGroup: local sum in the process, gloo between the processes) — the layout of
    the 8-robot graph on 2 or 4 GPUs — give what 4 processes with one shard each give (same association, same passes; only the
    order of the floating-point sums of the exchange differs)."""
    a = _run_workers("oracle", "C4tiny", 40, str(tmp_path / "t22.npz"), world=2, extra=("threads=2",))
    b = _run_workers("oracle", "C4tiny", 40, str(tmp_path / "t41.npz"), world=4)
    assert int(a["n_slots"]) == int(b["n_slots"]) > 0 and list(a["n_global"]) == list(b["n_global"])
    assert a["poses"].shape == b["poses"].shape == (4, 30, 12)
    assert np.abs(a["poses"] - b["poses"]).max() < 1e-9
