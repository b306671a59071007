"""-m gpu tests of the BASELINE configurations at FULL size (the timed configuration was never checked):
configs[3] C4 as bench.py runs it, configs[2] C3 on two ranks, configs[4] C5 streaming with the latency budget asserted."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu


def _scenario(name, out, *extra, timeout=1100):
    # (the scenario's progress lines go straight to this process's stderr: a long run must not look hung)
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tests", "gpu_scenarios.py"), name, out, *map(str, extra)], cwd=ROOT,
                       timeout=timeout)
    assert r.returncode == 0


def test_c4_eight_robots_batched_pass_parity(gpu, tmp_path):
    """configs[3]: the 8-robot / 5 k-pose / 10 k-landmark job exactly as bench.py times it (eight sub-graphs in one CholBatch, 59
    block columns each, the whole pass one replayed hipGraph): per pass equal to the un-batched path (1e-8) and to eight oracle
    shards driven by the same PassDriver (1e-4 is the north-star bar)."""
    out = str(tmp_path / "c4.json")
    _scenario("c4_parity", out)
    z = json.load(open(out))
    assert z["finite"] and z["chol_dim"] == 3776
    assert z["n_slots"][0] == z["n_slots"][1] == z["n_slots"][2] > 500          # ~870 landmarks are really observed by two or more robots
    assert z["n_global"][0] == z["n_global"][1]
    assert len(z["batched_vs_unbatched"]) >= 3
    assert max(z["batched_vs_unbatched"]) < 1e-8, z["batched_vs_unbatched"]
    # the north-star bar is 1e-4; measured 1.6e-6 .. 2.5e-6 on these first passes from the un-refined ingest-only state (3776-dim
    # reduced systems with the 1e-6 prior sigma in them: the two Cholesky factorisations sum in different orders)
    assert max(z["batched_vs_oracle"]) < 1e-4, z["batched_vs_oracle"]
    assert max(z["batched_vs_oracle"]) < 2e-5, z["batched_vs_oracle"]


def test_c3_full_size_two_ranks_converge_to_the_joint_optimum(gpu, tmp_path):
    """configs[2] at size: 2 robots x 500 poses, 30 % shared landmarks, one rank per robot (two processes on the one visible GPU,
    gloo staged through the host standing in for RCCL), each rank's pass = the CholBatch parts with the all-reduce between them —
    converges to the optimum of the joint graph a single host replica holds (computed on the GPU as well: the oracle would need
    hours for the 1000-pose streaming replay; its agreement with the product is what the small-size tests establish)."""
    from test_distributed import _run_workers
    jout = str(tmp_path / "joint.npz")
    _scenario("c3_joint", jout)
    J = np.load(jout)
    z = _run_workers("gpu", "C3", 60, str(tmp_path / "c3.npz"), world=2, extra=("driver=1",))
    print("C3 shared slots:", int(z["n_slots"]), "inventory", list(z["n_global"]), "joint", list(J["counts"]))
    assert int(z["n_slots"]) > 150                               # the ~190 landmarks of the 36 m overlap strip both robots observed
    # (nearly) the same landmark inventory as the joint replica: the merge of the two final maps may differ from the replica's
    # frame-by-frame association by a landmark or two
    assert sum(abs(int(a) - int(b)) for a, b in zip(z["n_global"], J["counts"])) <= 2, (list(z["n_global"]), list(J["counts"]))
    d, joint = z["poses"], J["poses"]
    assert d.shape == joint.shape == (2, 500, 12)
    rel = np.linalg.norm((d - joint).reshape(2, -1), axis=1) / np.linalg.norm(joint.reshape(2, -1), axis=1)
    assert rel.max() < 1e-4, rel


def test_c5_eight_robots_streaming_within_the_latency_budget(gpu, tmp_path):
    """configs[4]: eight robots streaming, every node ingesting its neighbour's packets; every per-update latency (own frame +
    foreign packet + solves + map refresh) stays inside the 100 ms budget of a 10 Hz key-frame rate, at the full 625-frame length
    (final graphs: 1250 poses per node)."""
    out = str(tmp_path / "c5.json")
    _scenario("c5_stream", out)
    z = json.load(open(out))
    assert z["finite"] and z["ticks"] == 625 and z["robots"] == 8
    assert all(n == 1250 for n in z["n_pose"]) and all(r == 0 for r in z["rejected"])
    assert z["over_budget"] == 0 and z["max_ms"] <= z["budget_ms"], z
    assert z["p99_ms"] < 50.0, z
