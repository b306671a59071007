#!/usr/bin/env python3
"""bench.py — pose-graph updates/s + ms per Gauss-Newton iteration of the SlideSLAM backend hot path on MI355X.

A *step* is one pass of the hot path over the resident graph: relinearise every factor, assemble the
landmark-eliminated (Schur) pose system, factor + solve it (FP64-MFMA Cholesky), back-substitute the
landmarks and retract — i.e. one Gauss-Newton iteration = one pose-graph update
(reference: SemanticFactorGraph::solve, backend/sloam/src/factorgraph/graph.cpp:260-272).

Workload (BASELINE.json configs[3]): the 8-robot / 10 k-landmark / 5 k-pose synthetic graph, one sub-graph per robot (625 poses,
~1250 landmarks, ~12.5 k landmark factors each), 8 / N robots per GPU — the SAME graph at N = 1, 2, 4, 8 (strong scaling).
Robots that share a GPU run on concurrent HIP streams (one host thread each); a step = one distributed Gauss-Newton pass of all
eight robots, value = robot pose-graph updates/s = 8 * steps / time.  `--robots-per-gpu 1` is the weak-scaling variant (one robot
per GPU at every N).  Inputs are resident in HBM before the timed region.  Synthetic, seeded data (slide_slam_amd/synth.py).

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--robots-per-gpu R] [--no-cpu] [--frames F] [--ingest-only]
Multi-GPU: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix peak, public spec (MI355X_MICROARCH.md lists no f64 row)
HBM_PEAK_GBS = 8000.0


def build_graph(s, data, robot_log_idx, frames=None, ingest_only=False):
    """Stream one robot's frame log through the per-frame path (association + add + iSAM2-equivalent update).
    ingest_only (profiling aid): add every frame without solving (association against the un-refined map at the ground-truth
    poses, as the cpu_baseline leg does), then one solve — every k_chol_step launch of the run is then full-size."""
    from slide_slam_amd.replay import replay_single
    gb = s.SlideBackend(s.default_params(), 1)
    if not ingest_only:
        out = replay_single(gb, data["logs"][robot_log_idx], n_frames=frames, collect=False)
        return gb, out
    from slide_slam_amd.synth import frame_detections
    log = data["logs"][robot_log_idx]
    P = len(log["rel7"]) if frames is None else frames
    t_frame = []
    for k in range(P):
        t0 = time.perf_counter()
        gb.process_frame(0, log["rel7"][k], log["gt7"][k], frame_detections(log, k), s.FRAME_FOREIGN)
        t_frame.append(time.perf_counter() - t0)
    if gb.ingest_solve() != 0:
        raise RuntimeError("ingest solve failed")
    return gb, dict(t_frame=t_frame)


def cpu_baseline(data, robot_log_idx, frames, threads):
    """The oracle (CPU restatement, C++ -O3 -march=native, `threads` OpenMP threads) timed on the same graph:
    all frames are ingested without solving (association against the un-refined map), then full
    linearise + Schur + Cholesky + back-substitution passes (threshold 0) are timed for about 12 s; median."""
    from oracle import pyoracle as po
    from slide_slam_amd.synth import frame_detections
    L = po.lib(native=True)
    ob = po.OracleBackend(po.OrcParams.default(num_threads=threads), 1, L=L)
    log = data["logs"][robot_log_idx]
    P = len(log["rel7"]) if frames is None else frames
    gt = log["gt7"]
    for k in range(P):
        ob.process_frame(0, log["rel7"][k], gt[k], frame_detections(log, k), 2)
    ob.graph.set_relin_threshold(0.0)
    times = []
    t_start = time.perf_counter()
    while len(times) < 2 or (time.perf_counter() - t_start < 12.0 and len(times) < 60):     # bounded: about 12 s of CPU work
        t0 = time.perf_counter()
        st = ob.ingest_solve()
        times.append(time.perf_counter() - t0)
        if st != 0:
            raise RuntimeError("oracle solve failed")
    stats = ob.graph.stats()
    per_iter = float(np.median(times))
    return dict(value=1.0 / per_iter, unit="pose-graph updates/s", cores=threads, kind="port",
                sample=f"{len(times)} full Gauss-Newton iterations (about 12 s) of the same {stats['n_pose']}-pose / {stats['n_lm']}-landmark / "
                       f"{stats['n_factors']}-factor graph (oracle = CPU restatement of the reference, not GTSAM), median",
                ms_per_iter=per_iter * 1e3, t_linearize_s=stats["t_linearize"], t_schur_s=stats["t_schur"],
                t_chol_s=stats["t_chol"])


def run_local_robots(args, R, s, torch, dist, rank, world, dev_index, backend):
    """--robots-per-gpu R > 1: R robot shards per process, one thread and one HIP stream each, exchanging through
    ThreadGroup (local sum, then RCCL across processes).  With R = 8 / N this is BASELINE's "8-robot graph at 1/2/4/8 GPUs"
    (total work fixed); a step = one distributed Gauss-Newton pass of all robots."""
    import threading
    from slide_slam_amd.distributed import DistributedGraph, ThreadGroup, TorchComm, gpu_matcher
    from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world
    cfg = SynthConfig.preset(args.preset)
    world_map = make_world(cfg)
    device = torch.device("cuda", dev_index)
    base = TorchComm(device=device, stage_through_host=(backend != "nccl")) if world > 1 else None
    group = ThreadGroup(R, base=base, rank=rank, world=world)
    sync = threading.Barrier(R + 1)
    conc = int(os.environ.get("SLIDE_BENCH_CONCURRENCY", "0"))
    sem = threading.Semaphore(conc) if conc > 0 else None
    # one launch sequence for the factorisations of all local robots pays off from about eight robots per GPU on (measured: 2 / 4 /
    # 8 robots 1.91 / 3.32 / 6.10 ms per pass batched, 1.78 / 3.03 / 6.92 ms on independent streams); SLIDE_BENCH_BATCH=0|1 forces
    use_batch = os.environ.get("SLIDE_BENCH_BATCH", "1" if R > 4 else "0") == "1"
    batch = s.CholBatch(R) if use_batch else None
    timing = [None]
    batched_prof = [None]
    bufs = [None] * R
    one_driver = batch is not None and world == 1 and os.environ.get("SLIDE_BENCH_ONE_DRIVER", "1") == "1"
    shards, infos, reps, errs = [None] * R, [None] * R, [None] * R, []

    def work(t):
        try:
            torch.cuda.set_device(dev_index)
            robot = (rank * R + t) % cfg.robots
            data = dict(cfg=cfg, world=world_map, logs={robot: make_robot_log(cfg, world_map, robot)})
            gb, reps[t] = build_graph(s, data, robot, args.frames, args.ingest_only)
            if sem is not None:        # diagnostic: at most SLIDE_BENCH_CONCURRENCY shards inside a phase at a time
                orig = gb.graph.dist_phase

                def limited(ph, buf, orig=orig):
                    with sem:
                        return orig(ph, buf)
                gb.graph.dist_phase = limited
            dg = DistributedGraph(gb, group.comm(t, device), rank * R + t, world * R)
            infos[t] = dg.setup(gpu_matcher)
            shards[t] = gb
            if batch is not None:      # the dense factor + solve of all local robots as one launch sequence per pass
                gb.graph.join_chol_batch(batch, t)
                dg.local_batch = world == 1     # all robots of the job on this GPU: exchanges as device-side sums, one sync per pass
                bufs[t] = dg.buf
            if one_driver:
                # the whole pass of all robots is one captured graph replayed by the main thread (slide_chol_batch_pass)
                sync.wait()          # ready
                sync.wait()          # main thread is through
                gb.graph.join_chol_batch(None)
                return
            for _ in range(args.warmup):
                dg.gauss_newton(1)
            if t == 0 and os.environ.get("SLIDE_BENCH_TIMING") == "1":      # diagnostic: wall time per call of the pass, thread 0
                acc = {}
                def timed(name, fn):
                    def w(*a):
                        t0 = time.perf_counter(); r = fn(*a); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
                    return w
                orig_phase = gb.graph.dist_phase
                gb.graph.dist_phase = lambda ph, buf: timed(f"phase{ph}", orig_phase)(ph, buf)
                dg.comm.all_reduce = timed("all_reduce", dg.comm.all_reduce)
                timing[0] = acc
            sync.wait()          # warm-up done
            sync.wait()          # go
            for _ in range(args.steps):
                dg.gauss_newton(1)
            sync.wait()          # done
            if batch is not None:
                gb.graph.join_chol_batch(None)
        except BaseException as e:
            errs.append(e)
            group.barrier.abort()
            sync.abort()

    th = [threading.Thread(target=work, args=(t,)) for t in range(R)]
    for x in th:
        x.start()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    try:
        if one_driver:
            sync.wait()          # every robot built, associated and joined
            ptrs = [b.data_ptr() for b in bufs]
            for _ in range(args.warmup):
                batch.pass_all(ptrs)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                batch.pass_all(ptrs)
            barrier()
            dt = time.perf_counter() - t0
            # device time of the batched step kernels (HIP events on the batch's stream, un-captured passes) for the roofline
            pr = sorted(batch.profile(ptrs) for _ in range(5))
            batched_prof[0] = dict(ms_steps=pr[len(pr) // 2][0], launches=pr[0][1])
            sync.wait()
        else:
            sync.wait()
            barrier()
            t0 = time.perf_counter()
            sync.wait()
            sync.wait()
            barrier()
            dt = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        dt = float("nan")
    for x in th:
        x.join()
    if errs:
        raise errs[0]
    if timing[0] and rank == 0:
        sys.stderr.write("per-pass wall ms (thread 0): " + ", ".join(f"{k} {v / args.steps * 1e3:.3f}" for k, v in sorted(timing[0].items())) + "\n")
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    infos[0]["mode"] = ("one replayed hipGraph per pass, factorisations batched" if one_driver else
                        ("factorisations batched, one host thread per robot" if batch is not None else "concurrent HIP streams, one host thread per robot"))
    if batched_prof[0]:
        infos[0]["batched"] = dict(batched_prof[0], robots=R)
    return dt, shards[0], reps[0], infos[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preset", default="C4")
    ap.add_argument("--frames", type=int, default=None, help="truncate each robot's log (debug)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--ingest-only", action="store_true", help="build the graph without per-frame solves (profiling aid)")
    ap.add_argument("--robots-per-gpu", type=int, default=0,
                    help="robot shards per GPU, on concurrent streams; 0 = the preset's robots / N when that divides (the SAME "
                         "8-robot graph at every N: strong scaling), else 1; 1 = one robot per GPU at every N (weak scaling)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch
    dist = None
    # SLIDE_BENCH_BACKEND=gloo rehearses the N > 1 path with every rank on GPU 0 (collectives staged through the
    # host); the real runs use nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("SLIDE_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(0)

    import slide_slam_amd as s
    from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world
    s.device_check()
    cfg = SynthConfig.preset(args.preset)
    R = args.robots_per_gpu if args.robots_per_gpu > 0 else (cfg.robots // world if cfg.robots % world == 0 else 1)
    if R > 1:
        # several robot shards on this GPU (one thread + one HIP stream each); everything below reports on robot 0's shard
        t_b0 = time.perf_counter()
        dt, gb, rep, dg_info = run_local_robots(args, R, s, torch, dist, rank, world, dev_index, backend)
        data = dict(cfg=cfg, logs={0: make_robot_log(cfg, make_world(cfg), 0)}) if rank == 0 and not args.no_cpu else None
        return report(args, s, cfg, rank, world, R, backend, dt, gb, rep, dg_info, dist, data, 0, time.perf_counter() - t_b0 - dt)
    # one robot per GPU: rank r replays robot r of the shared world (weak scaling; N = cfg.robots is the full config)
    robot = rank % cfg.robots
    world_map = make_world(cfg)
    data = dict(cfg=cfg, world=world_map, logs={robot: make_robot_log(cfg, world_map, robot)})
    t_b0 = time.perf_counter()
    gb, rep = build_graph(s, data, robot, args.frames, args.ingest_only)
    t_build = time.perf_counter() - t_b0
    g = gb.graph
    st = g.stats()
    dg_info = None
    if world > 1:
        # one robot per GPU: shared landmarks are associated across ranks once, then every Gauss-Newton pass
        # exchanges their normal-equation blocks with two all-reduces (slide_slam_amd/distributed.py)
        from slide_slam_amd.distributed import DistributedGraph, TorchComm, gpu_matcher
        comm = TorchComm(device=torch.device("cuda", dev_index), stage_through_host=(backend != "nccl"))
        dg = DistributedGraph(gb, comm, rank, world)
        dg_info = dg.setup(gpu_matcher)
        step = lambda: dg.gauss_newton(1)
    else:
        step = lambda: g.gauss_newton(1)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    return report(args, s, cfg, rank, world, 1, backend, dt, gb, rep, dg_info, dist, data, robot, t_build)


def report(args, s, cfg, rank, world, R, backend, dt, gb, rep, dg_info, dist, data, robot, t_build):
    """Profile pass on this rank's first shard + the JSON line (rank 0)."""
    g = gb.graph
    st = g.stats()
    robots = world * R
    # per-kernel device time (HIP events on the launch stream) over a separate profiled pass
    g.set_profiling(True)
    nprof = max(3, min(args.steps, 5))
    for _ in range(nprof):
        g.gauss_newton(1)
    prof = g.get_profile()
    g.set_profiling(False)

    if rank == 0:
        T = st["chol_dim"] // 64
        n = st["chol_dim"]
        # algorithmic FLOPs of one factorisation (RHS row included), per block column k with n_k rows below it:
        # trailing update n_k^2 * 64 + triangular solve n_k * 64^2 + diagonal block 64^3 / 3   (= n^3/3 overall)
        nk = [(T - k - 1) * 64 + 1 for k in range(T)]
        upd_flops = sum(v * v * 64.0 + v * 64.0 * 64.0 + 64.0 ** 3 / 3.0 for v in nk)
        upd = prof.get("chol_step", dict(ms=0.0, launches=1))
        upd_ms = upd["ms"] / max(upd["launches"], 1)
        upd_launches_per_iter = upd["launches"] / nprof
        flops_per_launch = upd_flops / max(upd_launches_per_iter, 1)
        ach = flops_per_launch / (upd_ms * 1e-3) / 1e12 if upd_ms > 0 else 0.0
        # HBM bytes per launch of the same kernel from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
        # passes, gfx950 FETCH_SIZE x2 correction): collected offline with tools/chol_big.py on the same reduced-system size
        # and committed under profiles/ (a PMC pass cannot run inside the timed bench)
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
                pmc = json.load(fh)
            if pmc.get("kernel") == "k_chol_step" and n == 3776:
                traffic = pmc["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            traffic = None
        single = dict(kernel="k_chol_step (one robot alone)", achieved=ach, frac=ach / FP64_MFMA_PEAK_TFLOPS, flops_per_launch=flops_per_launch,
                      avg_launch_ms=upd_ms, traffic=traffic)
        roof_kernel = "k_chol_step (v_mfma_f64_16x16x4_f64)"
        bt = dg_info.get("batched") if dg_info else None
        if bt:
            # the timed region ran k_chol_step_batched: all robots of the GPU per launch
            roof_kernel = f"k_chol_step_batched (v_mfma_f64_16x16x4_f64, {bt['robots']} factorisations per launch)"
            upd_launches_per_iter = bt["launches"]
            upd_ms = bt["ms_steps"] / max(bt["launches"], 1)
            flops_per_launch = bt["robots"] * upd_flops / max(bt["launches"], 1)
            ach = flops_per_launch / (upd_ms * 1e-3) / 1e12
            traffic = None
            try:
                with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_batched.json")) as fh:
                    pmc = json.load(fh)
                if pmc.get("kernel") == "k_chol_step_batched" and n == 3776 and pmc.get("robots") == bt["robots"]:
                    traffic = pmc["hbm_bytes_per_launch"]
            except (OSError, ValueError, KeyError):
                traffic = None
        kernel_ms = {k: v["ms"] / nprof for k, v in prof.items()}
        dominant = max(kernel_ms, key=kernel_ms.get)
        res = {
            "metric": "pose-graph updates/sec + ms/Gauss-Newton iter, 8-robot 10k-landmark graph",
            "value": robots * args.steps / dt,
            "unit": "pose-graph updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_gn_iter": dt / args.steps * 1e3,
            "higher_is_better": True,
            # the preset's robots over N GPUs (total work fixed) -> strong; a fixed number of robots per GPU -> weak
            "scaling": "strong" if (args.robots_per_gpu == 0 and robots == cfg.robots) else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (seeded, slide_slam_amd/synth.py)",
            "config": {"workload": f"{cfg.name} (BASELINE configs[3]): {robots} robot sub-graphs, {R} per GPU"
                                   + (f" ({dg_info['mode']})" if R > 1 and dg_info and "mode" in dg_info else "")
                                   + f" ({st['n_pose']} poses, {st['n_lm']} landmarks, {st['n_factors']} factors in robot 0's); "
                                     "a step = one Gauss-Newton pass of all of them, value = robot pose-graph updates/s",
                       "robots": robots, "robots_per_gpu": R, "reduced_system_dim": n, "chol_tile": 64,
                       "collective": None if dg_info is None else
                       (("local sum + " if R > 1 else "") + (f"{backend} " if world > 1 else "no inter-GPU ") +
                        f"all-reduce x2 per pass over {dg_info['n_slots']} shared-landmark slots ({dg_info['n_slots'] * 63 * 8} B per pass)")},
            "roofline": {"bound": "mfma", "kernel": roof_kernel, "achieved": ach,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_unit": "HBM-side bytes per launch (PMC, profiles/r01_pmc_traffic*.json)",
                         "one_robot_alone": single,
                         "flops_per_launch": flops_per_launch, "avg_launch_ms": upd_ms,
                         "launches_per_iter": upd_launches_per_iter, "dominant_by_time": dominant,
                         "scope": ("HIP events on the launch stream around the step launches of un-captured passes after the timed region; "
                                   "one_robot_alone = robot 0's sub-graph by itself (the per-GPU load of the N = 8 run)")},
            "kernel_ms_per_iter": kernel_ms,
            "stream_replay": {"frames": len(rep["t_frame"]), "updates_per_s": len(rep["t_frame"]) / max(sum(rep["t_frame"]), 1e-9),
                              "ms_last_frame": rep["t_frame"][-1] * 1e3, "build_s": t_build},
        }
        if not args.no_cpu:
            thr = args.cpu_threads or min(os.cpu_count() or 1, 16)
            try:
                res["cpu_baseline"] = cpu_baseline(data, robot, args.frames, thr)
            except Exception as e:  # the GPU number stands on its own
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
