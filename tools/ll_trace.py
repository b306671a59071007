"""Diagnostic: the left-looking persistent factorisation (k_chol_ll) on one dense system with the progress trace on (SLIDE_LL_TRACE=1:
every task leaves the wall clock of its stages in host-pinned memory; a watchdog in slide_dense_spd_solve_ex prints where every task
stands when the launch is not through within 8 s, and leaves).  Prints the cadence of the diagonal chain: per block column, when the
chain task started, had the older panels summed, saw the flag of tile (k, k-1), was staged, had factored, had published.
usage: python tools/ll_trace.py [n ...]"""
import os
import sys
import time

os.environ["SLIDE_LL_TRACE"] = "1"
os.environ["SLIDE_LL_TRACE_FILE"] = "/tmp/ll_trace.bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402

import slide_slam_amd as s  # noqa: E402

s.device_check()
for n in [int(a) for a in sys.argv[1:]] or [1000, 3776]:
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, n))
    A = B @ B.T / n + np.eye(n)
    b = rng.normal(size=n)
    x, ms = s.dense_spd_solve(A, b, method=1)
    t0 = time.perf_counter()
    x, ms = s.dense_spd_solve(A, b, method=1)
    ref = np.linalg.solve(A, b)
    T = (n + 63) // 64
    print(f"n {n:5d}  T {T:3d}  rel err {np.linalg.norm(x - ref) / np.linalg.norm(ref):.2e}  device {ms:.3f} ms = {ms * 1e3 / T:.1f} us per block column  wall {time.perf_counter() - t0:.2f} s", flush=True)
    x0, ms0 = s.dense_spd_solve(A, b, method=0)
    x0, ms0 = s.dense_spd_solve(A, b, method=0)
    print(f"         step kernels: rel diff {np.linalg.norm(x - x0) / np.linalg.norm(ref):.2e}  device {ms0:.3f} ms = {ms0 * 1e3 / T:.1f} us per block column", flush=True)
    tr = np.fromfile("/tmp/ll_trace.bin", dtype=np.int32).reshape(-1, 16)
    chain = tr[tr[:, 1] == 0]
    chain = chain[np.argsort(chain[:, 2])]
    t_ref = chain[0, 9]
    us = lambda v: (np.int64(v) - t_ref) / 100.0      # noqa: E731
    print("   k   start  summed flag(k,k-1) staged factored published | from the previous column's publication: flag staged factored published")
    for i, c in enumerate(chain):
        prev = us(chain[i - 1, 13]) if i else 0.0
        print(f"  {c[2]:3d} {us(c[9]):7.1f} {us(c[10]):7.1f} {us(c[14]) if c[14] else 0:9.1f} {us(c[11]):7.1f} {us(c[12]):8.1f} {us(c[13]):9.1f} |"
              f" {(us(c[14]) - prev) if c[14] else 0:6.1f} {us(c[11]) - prev:6.1f} {us(c[12]) - prev:7.1f} {us(c[13]) - prev:7.1f}")
    tiles = tr[tr[:, 1] == 1]
    if len(tiles):
        d = (tiles[:, 13].astype(np.int64) - tiles[:, 15]) / 100.0
        w = (tiles[:, 15].astype(np.int64) - tiles[:, 10]) / 100.0
        pre = (tiles[:, 10].astype(np.int64) - tiles[:, 9]) / 100.0
        print(f"   tile tasks: {len(tiles)}; start -> sums done: mean {pre.mean():.1f} us (max {pre.max():.1f}); waiting for the diagonal block: mean {w.mean():.1f}; "
              f"diagonal seen -> published: mean {d.mean():.1f} us (max {d.max():.1f})")
