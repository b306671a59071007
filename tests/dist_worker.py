"""Worker of the multi-process tests: one robot per rank, distributed Gauss-Newton, results to a .npz.

backend = oracle  -> CPU restatement shards, gloo all-reduce on host buffers   (runs anywhere; -m "not gpu")
backend = gpu     -> HIP shards (all ranks on the visible GPU), gloo all-reduce staged through the host
                     (RCCL needs one GPU per rank; the driver exercises that path with bench.py --gpus N)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def oracle_matcher(cls, xyz, lab, gxyz, glab, thresh):
    import ctypes as C
    from oracle import pyoracle as po
    n, m = len(lab), len(glab)
    out = np.full(max(n, 1), -1, np.int32)
    P = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    x, g = np.ascontiguousarray(xyz, dtype=np.float64), np.ascontiguousarray(gxyz, dtype=np.float64)
    la, gl = np.ascontiguousarray(lab, dtype=np.int32), np.ascontiguousarray(glab, dtype=np.int32)
    if cls == 0:   # cylinders: roots only are exchanged; vertical rays reproduce the matcher's point-at-height rule
        ray = np.tile([0.0, 0.0, 1.0], (n, 1)); gray = np.tile([0.0, 0.0, 1.0], (m, 1))
        rad = np.zeros(max(n, 1)); grad = np.zeros(max(m, 1))
        po.lib().orc_match_cylinders(C.c_int(n), P(x), P(ray), P(rad), P(la), C.c_int(m), P(g), P(gray), P(grad), P(gl),
                                     C.c_double(thresh), P(out))
    else:
        po.lib().orc_match_boxes(C.c_int(cls), C.c_int(n), P(x), P(la), C.c_int(m), P(g), P(gl), C.c_double(thresh), P(out))
    return out[:n]


from slide_slam_amd.distributed import gpu_matcher      # noqa: E402  (the product's matcher; re-exported for the tests)


def main_threads(R):
    """R robot shards per process (threads), several processes: ThreadGroup over a gloo TorchComm (oracle shards, CPU)."""
    import threading
    backend, preset, iters, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    assert backend == "oracle"
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from oracle import pyoracle as po
    from slide_slam_amd.distributed import DistributedGraph, ThreadGroup, TorchComm
    from slide_slam_amd.replay import replay_single
    from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world
    cfg = SynthConfig.preset(preset)
    assert cfg.robots == world * R
    world_map = make_world(cfg)
    group = ThreadGroup(R, base=TorchComm(device=None), rank=rank, world=world)
    out, err = [None] * R, []

    def work(t):
        try:
            v = rank * R + t
            shard = po.OracleBackend(po.OrcParams.default(), 1)
            log = make_robot_log(cfg, world_map, v)
            replay_single(shard, log, robot=0, collect=False)
            dg = DistributedGraph(shard, group.comm(t, None), v, world * R)
            info = dg.setup(oracle_matcher)
            dg.gauss_newton(iters)
            out[t] = (np.array([shard.graph.get_pose12(0, k)[1] for k in range(len(log["rel7"]))]), info)
        except BaseException as e:
            err.append(e)
            group.barrier.abort()

    th = [threading.Thread(target=work, args=(t,)) for t in range(R)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    if err:
        raise err[0]
    gathered = [None] * world
    dist.all_gather_object(gathered, [o[0] for o in out])
    if rank == 0:
        np.savez(out_path, poses=np.array([p for part in gathered for p in part]), n_slots=out[0][1]["n_slots"],
                 n_global=np.array(out[0][1]["n_global"]), n_gslots=0)


def main_driver(R):
    """R robot shards per process, ONE driver thread (setup_local_shards + PassDriver), several processes.
    backend oracle: CPU shards, gloo.  backend gpu: HIP shards in a CholBatch on the one visible GPU, the pass in its three captured
    parts with the gloo all-reduce staged through the host between them (stands in for RCCL, which needs one GPU per rank)."""
    backend, preset, iters, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    nccl = os.environ.get("SLIDE_TEST_NCCL") == "1"      # one GPU per rank, RCCL (a node with >= world GPUs); else gloo, every rank on GPU 0
    if nccl:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from slide_slam_amd.distributed import PassDriver, TorchComm, setup_local_shards
    from slide_slam_amd.replay import replay_single
    from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world
    cfg = SynthConfig.preset(preset)
    assert cfg.robots == world * R
    world_map = make_world(cfg)
    logs = [make_robot_log(cfg, world_map, rank * R + t) for t in range(R)]
    if backend == "oracle":
        from oracle import pyoracle as po
        shards = [po.OracleBackend(po.OrcParams.default(), 1) for _ in range(R)]
        base, device, batch, matcher = TorchComm(device=None), None, None, oracle_matcher
    else:
        import slide_slam_amd as s
        dev_index = torch.cuda.current_device() if nccl else 0
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
        torch.zeros(1, device=device)
        shards = [s.SlideBackend(s.default_params(), 1) for _ in range(R)]
        base, matcher = TorchComm(device=device, stage_through_host=not nccl), gpu_matcher
        batch = s.CholBatch(R)
    for sh, lg in zip(shards, logs):
        replay_single(sh, lg, robot=0, collect=False)
    if batch is not None:
        for t, sh in enumerate(shards):
            sh.graph.join_chol_batch(batch, t)
    bufs, info = setup_local_shards(shards, matcher, base=base, rank=rank, world=world, device=device)
    pcg = int(sys.argv[6].split("=")[1]) if len(sys.argv) > 6 and sys.argv[6].startswith("pcg=") else 0
    arrow = "arrow" in sys.argv[6:]       # the exact joint step: ONE all-reduce (the separator system) per pass
    drv = PassDriver(shards, bufs, info["n_slots"], batch=batch, base=base, world=world, device=device, pcg_iters=pcg, arrow=arrow,
                     sep_dim=info["sep_dim"], sep_prof=info.get("sep_prof"))
    if "relmeas" in sys.argv[6:] or "relmeas_dense" in sys.argv[6:]:     # inter-robot relative-pose factors of the job (every rank regenerates the same seeded list)
        from slide_slam_amd.synth import make_relmeas, make_relmeas_dense
        all_logs = [make_robot_log(cfg, world_map, r) for r in range(cfg.robots)]
        rel = make_relmeas_dense(cfg, all_logs) if "relmeas_dense" in sys.argv[6:] else make_relmeas(cfg, all_logs)      # (dense: SURVEY 8d's density)
        assert drv.setup_ghosts(rel, rank=rank) > 0
    drv.gauss_newton(iters)
    P = cfg.poses_per_robot
    mine = [np.array([sh.graph.get_pose12(0, k)[1] for k in range(P)]) for sh in shards]
    if batch is not None:
        for sh in shards:
            sh.graph.join_chol_batch(None)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        np.savez(out_path, poses=np.array([p for part in gathered for p in part]), n_slots=info["n_slots"],
                 n_global=np.array(info["n_global"]), n_gslots=0, owned=int(getattr(drv, "sep_owner", None) is not None))
    dist.barrier()
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 5 and sys.argv[5].startswith("threads="):
        return main_threads(int(sys.argv[5].split("=")[1]))
    if len(sys.argv) > 5 and sys.argv[5].startswith("driver="):
        return main_driver(int(sys.argv[5].split("=")[1]))
    backend, preset, iters, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from slide_slam_amd.distributed import DistributedGraph, TorchComm
    from slide_slam_amd.replay import replay_single
    from slide_slam_amd.synth import SynthConfig, make_robot_log, make_world
    cfg = SynthConfig.preset(preset)
    world_map = make_world(cfg)
    log = make_robot_log(cfg, world_map, rank)
    if backend == "oracle":
        from oracle import pyoracle as po
        shard = po.OracleBackend(po.OrcParams.default(), 1)
        comm = TorchComm(device=None)
        matcher = oracle_matcher
    else:
        import slide_slam_amd as s
        torch.cuda.set_device(0)
        shard = s.SlideBackend(s.default_params(), 1)
        comm = TorchComm(device=torch.device("cuda", 0), stage_through_host=True)
        matcher = gpu_matcher
    # every rank's robot is "robot 0" of its own shard (symbol X); the rank is the robot id of the job
    replay_single(shard, log, robot=0, collect=False)
    dg = DistributedGraph(shard, comm, rank, world)
    info = dg.setup(matcher)
    if len(sys.argv) > 5 and sys.argv[5] == "relmeas":
        # inter-robot relative-pose measurements of the job (every rank regenerates the same seeded list)
        from slide_slam_amd.synth import make_relmeas
        logs = [make_robot_log(cfg, world_map, r) for r in range(world)]
        info["n_gslots"] = dg.setup_ghosts(make_relmeas(cfg, logs))
    dg.gauss_newton(iters)
    P = len(log["rel7"])
    poses = np.array([shard.graph.get_pose12(0, k)[1] for k in range(P)])
    gathered = [None] * world
    dist.all_gather_object(gathered, (poses, info["n_slots"], info["n_global"], info.get("n_gslots", 0)))
    if rank == 0:
        np.savez(out_path, poses=np.array([g[0] for g in gathered]), n_slots=gathered[0][1], n_global=np.array(gathered[0][2]),
                 n_gslots=gathered[0][3])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
