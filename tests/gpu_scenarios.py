"""GPU scenarios of the BASELINE configs that need torch for device buffers (run in a fresh process: torch has to initialise the
device before this library's HIP runtime is loaded).  usage: gpu_scenarios.py <scenario> <out.json> [args]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


from chart_env import chart_kw, chart_name  # noqa: E402

T0 = time.perf_counter()


def say(*a):
    print(f"[{time.perf_counter() - T0:7.1f}s]", *a, file=sys.stderr, flush=True)


def ingest(shard, log, mode):
    """All frames at their ground-truth poses without per-frame solves (association against the un-refined map), one solve at the
    end: the same graph on the product and on the oracle in seconds (what bench.py --ingest-only and its cpu_baseline leg do)."""
    from slide_slam_amd.synth import frame_detections
    for k in range(len(log["rel7"])):
        shard.process_frame(0, log["rel7"][k], log["gt7"][k], frame_detections(log, k), mode)
    assert shard.ingest_solve() == 0


def poses_of(shards, P):
    return np.array([[sh.graph.get_pose12(0, k)[1] for k in range(P)] for sh in shards])


def c4_parity(out, preset="C4", passes=3):
    """BASELINE configs[3] as bench.py times it: eight robot sub-graphs in one CholBatch, the whole pass one replayed hipGraph —
    against (a) the un-batched path (slide_graph_dist_phase per robot, host-side sums) and (b) eight oracle shards, pass by pass."""
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    import slide_slam_amd as s
    from oracle import pyoracle as po
    from dist_worker import oracle_matcher
    from slide_slam_amd.distributed import PassDriver, gpu_matcher, setup_local_shards
    from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world
    cfg = SynthConfig.preset(preset)
    wm = make_world(cfg)
    logs = [make_robot_log(cfg, wm, r) for r in range(cfg.robots)]
    R, P = cfg.robots, cfg.poses_per_robot

    def gpu_shards():
        sh = [s.SlideBackend(s.default_params(**chart_kw(s)), 1) for _ in range(R)]
        for a, lg in zip(sh, logs):
            ingest(a, lg, s.FRAME_FOREIGN)
        return sh
    say("building 2 x", R, "GPU shards")
    A, B = gpu_shards(), gpu_shards()
    say("GPU shards built")
    batch = s.CholBatch(R)
    for t, a in enumerate(A):
        a.graph.join_chol_batch(batch, t)
    bufA, infoA = setup_local_shards(A, gpu_matcher, device=dev)
    bufB, infoB = setup_local_shards(B, gpu_matcher, device=dev)
    dA = PassDriver(A, bufA, infoA["n_slots"], batch=batch, device=dev)
    dB = PassDriver(B, bufB, infoB["n_slots"], device=dev)
    say("GPU shards associated:", infoA["n_slots"], "slots")
    L = po.lib(native=True)
    O = [po.OracleBackend(po.OrcParams.default(num_threads=min(os.cpu_count() or 1, 16), **chart_kw(po)), 1, L=L) for _ in range(R)]
    for o, lg in zip(O, logs):
        ingest(o, lg, 2)
    say("oracle shards built")
    bufO, infoO = setup_local_shards(O, oracle_matcher)
    dO = PassDriver(O, bufO, infoO["n_slots"])
    say("oracle shards associated:", infoO["n_slots"], "slots")
    res = dict(n_slots=[infoA["n_slots"], infoB["n_slots"], infoO["n_slots"]], n_global=[list(map(int, infoA["n_global"])), list(map(int, infoO["n_global"]))],
               batched_vs_unbatched=[], batched_vs_oracle=[], chol_dim=A[0].graph.stats()["chol_dim"], t_pass_batched_ms=[])
    for _ in range(passes):
        t0 = time.perf_counter()
        dA.one_pass()
        res["t_pass_batched_ms"].append((time.perf_counter() - t0) * 1e3)
        say("batched pass done")
        dB.one_pass()
        say("un-batched pass done")
        dO.one_pass()
        say("oracle pass done")
        a, b, o = poses_of(A, P), poses_of(B, P), poses_of(O, P)
        res["batched_vs_unbatched"].append(float(np.abs(a - b).max() / np.abs(b).max()))
        nrm = np.linalg.norm(o.reshape(R, -1), axis=1)
        res["batched_vs_oracle"].append(float((np.linalg.norm((a - o).reshape(R, -1), axis=1) / nrm).max()))
    # the joint solve (PCG on the global reduced system) batched in the replayed graph vs un-batched through dist_phase 31 / 32 / 33
    dA2 = PassDriver(A, bufA, infoA["n_slots"], batch=batch, device=dev, pcg_iters=4)
    dB2 = PassDriver(B, bufB, infoB["n_slots"], device=dev, pcg_iters=4)
    res["pcg_batched_vs_unbatched"] = []
    for _ in range(2):
        dA2.one_pass()
        dB2.one_pass()
        a, b = poses_of(A, P), poses_of(B, P)
        res["pcg_batched_vs_unbatched"].append(float(np.abs(a - b).max() / np.abs(b).max()))
        say("PCG pass: batched vs un-batched", res["pcg_batched_vs_unbatched"][-1], A[0].graph.pcg_stats())
    res["pcg_chi2"] = [sum(x.graph.chi2()["total"] for x in A), sum(x.graph.chi2()["total"] for x in B)]
    res["finite"] = bool(np.isfinite(poses_of(A, P)).all())
    for a in A:
        a.graph.join_chol_batch(None)
    json.dump(res, open(out, "w"))


def c3_joint(out, preset="C3", gn=12):
    """The joint optimum of BASELINE configs[2] on ONE GPU: a single host replica ingesting both robots (the reference's own
    arrangement, sloamNode.cpp:912-1002), then batch Gauss-Newton — what the two sharded ranks have to converge to."""
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    import slide_slam_amd as s
    from slide_slam_amd.replay import replay_multi
    from slide_slam_amd.synth import SynthConfig, make_dataset
    cfg = SynthConfig.preset(preset)
    data = make_dataset(cfg)
    data["relmeas"] = []
    gb = s.SlideBackend(s.default_params(number_of_robots=cfg.robots, **chart_kw(s)), cfg.robots)
    t0 = time.perf_counter()
    replay_multi(gb, data, own_node_factory=lambda: s.SlideBackend(s.default_params(**chart_kw(s)), 1))
    t_replay = time.perf_counter() - t0
    say("joint replica replayed", t_replay)
    gb.graph.gauss_newton(gn)
    P = cfg.poses_per_robot
    poses = np.array([[gb.graph.get_pose12(r, k)[1] for k in range(P)] for r in range(cfg.robots)])
    c = gb.counts()
    np.savez(out, poses=poses, counts=np.array([c["cyl"], c["cube"], c["point"]]), t_replay=t_replay, chol_dim=gb.graph.stats()["chol_dim"])


def c5_stream(out, preset="C4", ticks=None, budget_ms=100.0):
    """BASELINE configs[4]: eight robots streaming key frames; every robot's node runs its own frame (associate + add +
    iSAM2-equivalent update), ingests the packet its neighbour published for the same tick (sloamNode.cpp:912-1002: associate
    against the host's maps, add, one solve) and refreshes its map — the per-update latency of every node is measured against the
    100 ms budget of a 10 Hz key-frame rate.  One GPU hosts the eight nodes here, one after the other per tick; on the 8-GPU node
    each has its own."""
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    import slide_slam_amd as s
    from slide_slam_amd.replay import IDENT7, foreign_key_poses
    from slide_slam_amd.synth import SynthConfig, frame_detections, make_robot_log, make_world
    cfg = SynthConfig.preset(preset)
    wm = make_world(cfg)
    R = cfg.robots
    logs = [make_robot_log(cfg, wm, r) for r in range(R)]
    T = cfg.poses_per_robot if ticks is None else ticks
    # what every robot's own node publishes per key frame (PoseMstPair.keyPose, sloamNode.cpp:793-800)
    kposes = [foreign_key_poses(s.SlideBackend(s.default_params(**chart_kw(s)), 1), lg) for lg in logs]
    cols = cfg.grid[1]
    nbr = [(r + 1) if (r % cols) + 1 < cols else (r - 1) for r in range(R)]       # the neighbour in the same row of the grid
    nodes = [s.SlideBackend(s.default_params(number_of_robots=2, **chart_kw(s)), 2) for _ in range(R)]
    prev = [IDENT7.copy() for _ in range(R)]
    lat = np.zeros((T, R))
    say("key poses of the", R, "own nodes replayed")
    for k in range(T):
        if k % 100 == 0:
            say("tick", k, "max latency so far", float(lat.max()), "ms")
        for r in range(R):
            nd, o = nodes[r], nbr[r]
            t0 = time.perf_counter()
            rr = nd.process_frame(0, logs[r]["rel7"][k], prev[r], frame_detections(logs[r], k), s.FRAME_HOST_DEFERRED)
            assert rr["status"] == 0, (k, r, "host frame", rr["status"], s.api.last_error(), nd.graph.stats())
            nd.process_frame(1, logs[o]["rel7"][k], kposes[o][k], frame_detections(logs[o], k), s.FRAME_FOREIGN)
            st_i = nd.ingest_solve()
            assert st_i == 0, (k, r, "ingest", st_i, s.api.last_error(), nd.graph.stats())
            st, pose = nd.end_frame(0)
            assert st == 0
            lat[k, r] = (time.perf_counter() - t0) * 1e3
            prev[r] = pose.copy()
    stats = [nd.graph.stats() for nd in nodes]
    res = dict(ticks=T, robots=R, budget_ms=budget_ms, max_ms=float(lat.max()), p50_ms=float(np.median(lat)), p99_ms=float(np.percentile(lat, 99)),
               last_tick_ms=[float(v) for v in lat[-1]], over_budget=int((lat > budget_ms).sum()),
               n_pose=[st["n_pose"] for st in stats], chol_dim=[st["chol_dim"] for st in stats],
               rejected=[int(nd.graph.rejected_count()) for nd in nodes],
               worst=[(int(i // R), int(i % R), float(lat.flat[i])) for i in np.argsort(lat, axis=None)[::-1][:6]],      # (tick, robot, ms)
               finite=bool(all(np.isfinite(p).all() for p in prev)))
    json.dump(res, open(out, "w"))


def c3_converge(out, preset="C3", passes=400, every=20, pcg=0, unbatched=0, arrow=0, relmeas=0):
    """Convergence of the sharded passes (pcg = 0: block-Jacobi; > 0: joint solve by PCG; arrow = 1: the exact joint step) to the
    joint replica's optimum."""
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    import slide_slam_amd as s
    from slide_slam_amd.distributed import PassDriver, gpu_matcher, setup_local_shards
    from slide_slam_amd.replay import replay_multi, replay_single
    from slide_slam_amd.synth import SynthConfig, make_dataset
    cfg = SynthConfig.preset(preset)
    data = make_dataset(cfg)
    data["relmeas"] = []
    R, P = cfg.robots, cfg.poses_per_robot
    joint, chi2_joint, joint_counts = None, None, None
    if R <= 2:          # (the 8-robot replica's streaming build takes minutes: n = 30016 solved per frame)
        gb = s.SlideBackend(s.default_params(number_of_robots=R, **chart_kw(s)), R)
        replay_multi(gb, data, own_node_factory=lambda: s.SlideBackend(s.default_params(**chart_kw(s)), 1))
        prevj = None
        for it in range(30):
            gb.graph.gauss_newton(1)
            joint = np.array([[gb.graph.get_pose12(r, k)[1] for k in range(P)] for r in range(R)])
            if prevj is not None:
                say("joint GN", it, "step", float(np.abs(joint - prevj).max()))
            prevj = joint
        cj = gb.graph.chi2()
        chi2_joint = cj["total"]
        cnt = gb.counts()
        joint_counts = [cnt["cyl"], cnt["cube"], cnt["point"]]
        say("joint replica chi2", cj, gb.graph.stats(), cnt, "rejected", gb.graph.rejected_count())
        del gb
    shards = [s.SlideBackend(s.default_params(**chart_kw(s)), 1) for _ in range(R)]
    for sh, lg in zip(shards, data["logs"]):
        replay_single(sh, lg, collect=False)
    say("shards chi2 before", [sh.graph.chi2() for sh in shards], [sh.graph.stats() for sh in shards], [sh.counts() for sh in shards])
    batch = None if unbatched else s.CholBatch(R)
    if batch is not None:
        for t, sh in enumerate(shards):
            sh.graph.join_chol_batch(batch, t)
    bufs, info = setup_local_shards(shards, gpu_matcher, device=dev)
    drv = PassDriver(shards, bufs, info["n_slots"], batch=batch, device=dev, pcg_iters=pcg, arrow=bool(arrow), sep_dim=info["sep_dim"], sep_prof=info.get("sep_prof"))
    if relmeas:      # the job's inter-robot relative-pose factors (the joint replica above was built without them: `joint` is then no reference)
        from slide_slam_amd.synth import make_relmeas, make_relmeas_dense
        assert drv.setup_ghosts(make_relmeas_dense(cfg, data["logs"]) if relmeas == 2 else make_relmeas(cfg, data["logs"])) > 0      # (2: SURVEY 8d's density)
    nrm = np.linalg.norm(joint.reshape(R, -1), axis=1) if joint is not None else None
    hist, chi2_hist = [], []
    t_pass = 0.0
    for p in range(passes):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        drv.one_pass()
        torch.cuda.synchronize(); t_pass += time.perf_counter() - t0
        if (p + 1) % every == 0 or p < 5:
            d = poses_of(shards, P)
            e = float((np.linalg.norm((d - joint).reshape(R, -1), axis=1) / nrm).max()) if joint is not None else float("nan")
            hist.append((p + 1, e))
            chi2_hist.append(float(sum(sh.graph.chi2()["total"] for sh in shards)))
            say("pass", p + 1, "rel err vs joint", e, "chi2 sum", chi2_hist[-1], "ms/pass so far", 1e3 * t_pass / (p + 1),
                shards[0].graph.pcg_stats() if pcg else "")
    cs = [sh.graph.chi2() for sh in shards]
    say("shards chi2 after", cs, "sum", sum(c["total"] for c in cs))
    # replica consistency: the copies of every shared landmark must hold the same value in both shards
    from slide_slam_amd.distributed import associate_global, shared_slots
    tables = [[sh.landmark_table(c) for c in range(3)] for sh in shards]
    gid, n_global = associate_global(tables, (2.0, 2.0, 0.75), gpu_matcher)
    sl = [shared_slots(gid, n_global, r) for r in range(R)]
    worst, worst_tab = 0.0, 0.0
    for k in range(len(sl[0][0])):
        vals = [shards[r].graph.get_landmark(int(sl[r][0][k]), int(sl[r][1][k]))[1] for r in range(R) if sl[r][0][k] >= 0]
        tabs = [tables[r][int(sl[r][0][k])][0][int(sl[r][1][k])] for r in range(R) if sl[r][0][k] >= 0]
        for v in vals[1:]:
            worst = max(worst, float(np.abs(v - vals[0]).max()))
        for t in tabs[1:]:
            worst_tab = max(worst_tab, float(np.abs(t - tabs[0]).max()))
    say("shared landmark copies: max |difference| between the shards", worst, "(positions in the tables:", worst_tab, ") slots", len(sl[0][0]), info["n_slots"])
    if batch is not None:
        for sh in shards:
            sh.graph.join_chol_batch(None)
    json.dump(dict(hist=hist, chi2_hist=chi2_hist, worst_copy=worst, n_slots=info["n_slots"], final=poses_of(shards, P).tolist(), chi2_shards=sum(c["total"] for c in cs),
                   chi2_joint=(float(chi2_joint) if joint is not None else None), ms_per_pass=1e3 * t_pass / max(passes, 1),
                   n_global=[int(v) for v in info["n_global"]], joint_counts=joint_counts), open(out, "w"))


def c3_assoc_check(out, preset="C3"):
    """Diagnostic: is the landmark association of the joint replica (frame by frame against a growing map) the same partition of
    the detections as the merge of the two robots' final maps?"""
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    import slide_slam_amd as s
    from slide_slam_amd.distributed import associate_global, gpu_matcher
    from slide_slam_amd.replay import replay_multi, replay_single
    from slide_slam_amd.synth import SynthConfig, make_dataset
    cfg = SynthConfig.preset(preset)
    data = make_dataset(cfg)
    data["relmeas"] = []
    R, P = cfg.robots, cfg.poses_per_robot
    gb = s.SlideBackend(s.default_params(number_of_robots=R, **chart_kw(s)), R)
    jo = replay_multi(gb, data, own_node_factory=lambda: s.SlideBackend(s.default_params(**chart_kw(s)), 1))
    shards, outs = [], []
    for lg in data["logs"]:
        sh = s.SlideBackend(s.default_params(**chart_kw(s)), 1)
        outs.append(replay_single(sh, lg))
        shards.append(sh)
    tables = [[sh.landmark_table(c) for c in range(3)] for sh in shards]
    gid, n_global = associate_global(tables, (2.0, 2.0, 0.75), gpu_matcher)
    names = ("cyl_id", "cube_id", "ell_id")
    res = {}
    for c in range(3):
        pairs = set()
        for k in range(P):
            for r in range(R):
                jid = jo["ids"][k][r][c]
                lid = outs[r][names[c]][k]
                assert len(jid) == len(lid)
                for a, b in zip(jid, lid):
                    pairs.add((int(a), int(gid[r][c][int(b)])))
        j2s, s2j = {}, {}
        for a, b in pairs:
            j2s.setdefault(a, set()).add(b); s2j.setdefault(b, set()).add(a)
        res[names[c]] = dict(joint_ids=len(j2s), shard_ids=len(s2j), joint_split=sum(1 for v in j2s.values() if len(v) > 1),
                             shard_split=sum(1 for v in s2j.values() if len(v) > 1))
        say(names[c], res[names[c]])
    json.dump(res, open(out, "w"))


def tiny_pcg(out, preset="C3tiny", passes=15, pcg=8):
    """Sharded passes with the joint solve on the GPU against the ORACLE's joint replica (the reference's arrangement: one host graph
    holding every robot, batch Gauss-Newton to convergence) at a size the oracle replays in seconds."""
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    import slide_slam_amd as s
    from slide_slam_amd.distributed import PassDriver, gpu_matcher, setup_local_shards
    from slide_slam_amd.replay import replay_single
    from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world
    from test_distributed import _joint_optimum
    joint, counts = _joint_optimum(preset)
    cfg = SynthConfig.preset(preset)
    wm = make_world(cfg)
    R, P = cfg.robots, cfg.poses_per_robot
    res = {}
    for mode in ("batched", "unbatched"):
        shards = [s.SlideBackend(s.default_params(**chart_kw(s)), 1) for _ in range(R)]
        for r, sh in enumerate(shards):
            replay_single(sh, make_robot_log(cfg, wm, r), collect=False)
        batch = s.CholBatch(R) if mode == "batched" else None
        if batch is not None:
            for t, sh in enumerate(shards):
                sh.graph.join_chol_batch(batch, t)
        bufs, info = setup_local_shards(shards, gpu_matcher, device=dev)
        drv = PassDriver(shards, bufs, info["n_slots"], batch=batch, device=dev, pcg_iters=pcg)
        drv.gauss_newton(passes)
        d = poses_of(shards, P)
        rel = np.linalg.norm((d - joint).reshape(R, -1), axis=1) / np.linalg.norm(joint.reshape(R, -1), axis=1)
        res[mode] = dict(rel=float(rel.max()), n_slots=info["n_slots"], n_global=[int(v) for v in info["n_global"]],
                         chi2=sum(sh.graph.chi2()["total"] for sh in shards), poses=d.tolist())
        say(mode, "rel err vs the oracle's joint optimum", rel.max(), "slots", info["n_slots"])
        if batch is not None:
            for sh in shards:
                sh.graph.join_chol_batch(None)
    res["joint_counts"] = [counts["cyl"], counts["cube"], counts["point"]]
    res["batched_vs_unbatched"] = float(np.abs(np.array(res["batched"]["poses"]) - np.array(res["unbatched"]["poses"])).max())
    for m in ("batched", "unbatched"):
        del res[m]["poses"]
    json.dump(res, open(out, "w"))


def arrow_parity(out, preset="C4tiny", passes=6, mode="replay", with_joint=1, relmeas=0, assoc="merge"):
    """The EXACT joint step (shared landmarks as the separator of the joint graph) of the HIP shards in one CholBatch — the whole pass
    one replayed hipGraph — against oracle shards taking the same step pass by pass, and (small presets) against the optimum of the
    oracle's joint replica, the reference's arrangement.  mode: replay = streaming build per robot (per-frame solves), ingest = all
    frames at their ground-truth poses, one solve (bench.py's workload build)."""
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    import slide_slam_amd as s
    from oracle import pyoracle as po
    from dist_worker import oracle_matcher
    from slide_slam_amd.distributed import PassDriver, gpu_matcher, setup_local_shards
    from slide_slam_amd.replay import replay_single
    from slide_slam_amd.synth import SynthConfig, make_relmeas, make_relmeas_dense, make_robot_log, make_world
    cfg = SynthConfig.preset(preset)
    wm = make_world(cfg)
    R, P = cfg.robots, cfg.poses_per_robot
    logs = [make_robot_log(cfg, wm, r) for r in range(R)]
    joint, counts = None, None
    if with_joint:
        from test_distributed import _joint_optimum
        joint, counts = _joint_optimum(preset, relmeas=bool(relmeas))
    A = [s.SlideBackend(s.default_params(**chart_kw(s)), 1) for _ in range(R)]
    L = po.lib(native=True)
    O = [po.OracleBackend(po.OrcParams.default(num_threads=min(os.cpu_count() or 1, 16), **chart_kw(po)), 1, L=L) for _ in range(R)]
    own_out = []
    for a, o, lg in zip(A, O, logs):
        if mode == "ingest":
            ingest(a, lg, s.FRAME_FOREIGN)
            ingest(o, lg, 2)
        else:
            own_out.append(replay_single(a, lg, collect=(assoc == "ingest")))
            replay_single(o, lg, robot=0, collect=False)
    say("shards built")
    ingest_assoc = None
    if assoc == "ingest":
        # the reference's cross-robot association (sloamNode.cpp:912-1002): ONE host replica (the product plays it) ingests every robot's
        # packets frame by frame; the ids it gives the detections, against the ids the robots' own graphs gave them, define the global
        # landmarks (distributed.associate_by_ingest) — instead of the merge of the robots' FINAL maps
        from slide_slam_amd.distributed import associate_by_ingest
        from slide_slam_amd.replay import replay_multi
        rb = s.SlideBackend(s.default_params(**chart_kw(s)), R)
        rep = replay_multi(rb, dict(cfg=cfg, logs=logs, relmeas=[]), own_node_factory=lambda: s.SlideBackend(s.default_params(**chart_kw(s)), 1))
        names = ("cyl_id", "cube_id", "ell_id")
        own_ids = [[own_out[r][names[c]] for c in range(3)] for r in range(R)]
        rep_ids = [[[rep["ids"][k][r][c] for k in range(P)] for c in range(3)] for r in range(R)]
        gid, n_glob, st = associate_by_ingest(own_ids, rep_ids)
        say("ingest association:", n_glob, st)
        ingest_assoc = (gid, n_glob)
        rc_ = rb.counts()
        ingest_stats = dict(st, replica_counts=[rc_["cyl"], rc_["cube"], rc_["point"]])
    batch = s.CholBatch(R)
    for t, a in enumerate(A):
        a.graph.join_chol_batch(batch, t)
    bufA, infoA = setup_local_shards(A, gpu_matcher, device=dev, assoc=ingest_assoc)
    bufO, infoO = setup_local_shards(O, oracle_matcher, assoc=ingest_assoc)
    dA = PassDriver(A, bufA, infoA["n_slots"], batch=batch, device=dev, arrow=True, sep_dim=infoA["sep_dim"], sep_prof=infoA.get("sep_prof"))
    dO = PassDriver(O, bufO, infoO["n_slots"], arrow=True, sep_dim=infoO["sep_dim"], sep_prof=infoO.get("sep_prof"))
    say("associated:", infoA["n_slots"], infoO["n_slots"], "slots, separator", infoA["sep_dim"], infoO["sep_dim"])
    n_g = 0
    if relmeas:
        # inter-robot relative-pose factors (graph.cpp:247-258): ghost poses refreshed at the start of every pass, on both sides
        # (relmeas == 2: SURVEY 8d's density — every adjacent robot pair every 50 frames near the other's trajectory, two different
        # pose indices per factor)
        rel = make_relmeas_dense(cfg, logs) if relmeas == 2 else make_relmeas(cfg, logs)
        n_g = dA.setup_ghosts(rel)
        assert dO.setup_ghosts(rel) == n_g
        res_n_rel = len(rel)
        say("relative-pose measurements:", len(rel), "ghost slots:", n_g)
    res = dict(n_gslots=n_g, n_relmeas=(res_n_rel if relmeas else 0), chart=chart_name(), assoc=assoc, ingest=(ingest_stats if assoc == "ingest" else None), n_slots=[infoA["n_slots"], infoO["n_slots"]], sep_dim=[infoA["sep_dim"], infoO["sep_dim"]],
               n_global=[list(map(int, infoA["n_global"])), list(map(int, infoO["n_global"]))], gpu_vs_oracle=[], step=[], vs_joint=[], ms=[])
    prev = None
    for p in range(passes):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dA.one_pass()
        torch.cuda.synchronize(); res["ms"].append((time.perf_counter() - t0) * 1e3)
        dO.one_pass()
        a, o = poses_of(A, P), poses_of(O, P)
        nrm = np.linalg.norm(o.reshape(R, -1), axis=1)
        res["gpu_vs_oracle"].append(float((np.linalg.norm((a - o).reshape(R, -1), axis=1) / nrm).max()))
        res["step"].append(float(np.abs(a - prev).max()) if prev is not None else None)
        if joint is not None:
            res["vs_joint"].append(float((np.linalg.norm((a - joint).reshape(R, -1), axis=1) / np.linalg.norm(joint.reshape(R, -1), axis=1)).max()))
        prev = a
        res.setdefault("chi2_pass", []).append(sum(x.graph.chi2()["total"] for x in A))
        say("pass", p + 1, "GPU vs oracle", res["gpu_vs_oracle"][-1], "step", res["step"][-1], "vs joint", res["vs_joint"][-1:], "ms", res["ms"][-1], "chi2", res["chi2_pass"][-1])
    res["finite"] = bool(np.isfinite(prev).all())
    res["segments"] = [a.graph.segments() for a in A]
    say("segments of the bands:", res["segments"][:2])
    res["chi2"] = sum(x.graph.chi2()["total"] for x in A)
    if R * P <= 2000:
        res["final"] = prev.tolist()
    if counts is not None:
        res["joint_counts"] = [counts["cyl"], counts["cube"], counts["point"]]
    for a in A:
        a.graph.join_chol_batch(None)
    json.dump(res, open(out, "w"))


def pair_verify(out, preset="C4", relmeas=1):
    """Diagnostic: ONE un-captured exact joint pass (profile_exact_joint) of the preset's shards with SLIDE_PAIR_VERIFY=1 — every system the
    pair kernel factors is also factored by the step kernels on a copy and compared tile by tile (stderr)."""
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    import slide_slam_amd as s
    from slide_slam_amd.distributed import PassDriver, gpu_matcher, setup_local_shards
    from slide_slam_amd.synth import SynthConfig, make_relmeas, make_robot_log, make_world
    cfg = SynthConfig.preset(preset)
    wm = make_world(cfg)
    R = cfg.robots
    logs = [make_robot_log(cfg, wm, r) for r in range(R)]
    A = [s.SlideBackend(s.default_params(**chart_kw(s)), 1) for _ in range(R)]
    for a, lg in zip(A, logs):
        ingest(a, lg, s.FRAME_FOREIGN)
    batch = s.CholBatch(R)
    for t, a in enumerate(A):
        a.graph.join_chol_batch(batch, t)
    bufA, infoA = setup_local_shards(A, gpu_matcher, device=dev)
    dA = PassDriver(A, bufA, infoA["n_slots"], batch=batch, device=dev, arrow=True, sep_dim=infoA["sep_dim"], sep_prof=infoA.get("sep_prof"))
    if relmeas:
        dA.setup_ghosts(make_relmeas(cfg, logs))
    st, ns = batch.profile_exact_joint(dA.ptrs)
    say("stages", st, "separator block columns", ns, "pair timeouts", s.pair_timeouts())
    for a in A:
        a.graph.join_chol_batch(None)
    json.dump(dict(stages=st), open(out, "w"))


def rank_threads(out, preset="C4", world=8, passes=3, relmeas=0, mode="ingest"):
    """BASELINE configs[3]'s own arrangement on the ONE visible GPU: `world` ranks of cfg.robots / world robots, every rank a thread
    with its own CholBatch (LocalRanks: the TorchComm interface over barriers — the box allows few processes on the card), the pass cut
    at its exchanges exactly as a multi-process job cuts it (rank-owned leaves of the separator, the half's all-reduce among its
    ranks, one leader per half) — against ONE process holding all robots."""
    os.environ["SLIDE_NONBLOCKING_STREAMS"] = "1"      # several host threads capture and copy side by side (HostGraph::init)
    # Round 5: the rank threads CAPTURE and replay their passes like a process does.  Round 4 needed SLIDE_PASS_DIRECT=1 here; the two
    # causes are fixed at their source — every library thread runs in thread-local capture mode (host_graph.hip
    # thread_capture_mode_local: a thread in the default global mode invalidates other threads' captures with its hipMalloc / hipFree),
    # and the thread ranks' collectives no longer synchronise the whole device (LocalRankComm._sync).  SLIDE_TEST_THREAD_DIRECT=1: the
    # old direct issue.
    if os.environ.get("SLIDE_TEST_THREAD_DIRECT") == "1":
        os.environ["SLIDE_PASS_DIRECT"] = "1"
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    import slide_slam_amd as s
    from slide_slam_amd.distributed import PassDriver, gpu_matcher, setup_local_shards
    from slide_slam_amd.replay import replay_single
    from slide_slam_amd.synth import SynthConfig, make_relmeas, make_robot_log, make_world
    from test_pass_driver import run_thread_ranks
    cfg = SynthConfig.preset(preset)
    wm = make_world(cfg)
    R, P = cfg.robots, cfg.poses_per_robot
    logs = [make_robot_log(cfg, wm, r) for r in range(R)]
    rel = make_relmeas(cfg, logs) if relmeas else None

    def build():
        sh = [s.SlideBackend(s.default_params(**chart_kw(s)), 1) for _ in range(R)]
        for a, lg in zip(sh, logs):
            if mode == "ingest":
                ingest(a, lg, s.FRAME_FOREIGN)
            else:
                replay_single(a, lg, collect=False)
        return sh
    A, B = build(), build()
    say("2 x", R, "shards built")
    batch = s.CholBatch(R)
    for t, a in enumerate(A):
        a.graph.join_chol_batch(batch, t)
    bufs, info = setup_local_shards(A, gpu_matcher, device=dev)
    drv = PassDriver(A, bufs, info["n_slots"], batch=batch, device=dev, arrow=True, sep_dim=info["sep_dim"], sep_prof=info.get("sep_prof"))
    if rel:
        assert drv.setup_ghosts(rel) > 0
    drv.gauss_newton(passes)
    one = poses_of(A, P)
    for a in A:
        a.graph.join_chol_batch(None)
    say("one process:", passes, "passes done;", info["n_slots"], "slots")
    batches = []

    def factory(mine):
        b = s.CholBatch(len(mine))
        for t, a in enumerate(mine):
            a.graph.join_chol_batch(b, t)
        batches.append(b)
        return b
    infos = run_thread_ranks(B, gpu_matcher, world, device=dev, passes=passes, relmeas=rel, batch_factory=factory)
    many = poses_of(B, P)
    for b in B:
        b.graph.join_chol_batch(None)
    relerr = float((np.linalg.norm((many - one).reshape(R, -1), axis=1) / np.linalg.norm(one.reshape(R, -1), axis=1)).max())
    say(world, "thread ranks vs one process:", relerr, "owned:", [i["owned"] for i in infos])
    json.dump(dict(rel=relerr, finite=bool(np.isfinite(many).all()), n_slots=[int(info["n_slots"])] + [int(i["n_slots"]) for i in infos],
                   owned=[bool(i["owned"]) for i in infos], n_relmeas=len(rel) if rel else 0), open(out, "w"))


if __name__ == "__main__":
    fn = {"rank_threads": rank_threads, "pair_verify": pair_verify, "arrow_parity": arrow_parity, "c4_parity": c4_parity, "c3_joint": c3_joint, "c5_stream": c5_stream, "c3_converge": c3_converge, "c3_assoc_check": c3_assoc_check, "tiny_pcg": tiny_pcg}[sys.argv[1]]
    extra = [int(a) if a.lstrip("-").isdigit() else a for a in sys.argv[3:]]
    fn(sys.argv[2], *extra)
