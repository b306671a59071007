import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import slide_slam_amd as s
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16700
rng = np.random.default_rng(1)
B = rng.normal(size=(n, 64)).astype(np.float64)
A = B @ B.T / 64.0
A[np.diag_indices(n)] += 2.0 + rng.uniform(0, 1, n)
b = rng.normal(size=n)
t0 = time.time(); x, ms = s.dense_spd_solve(A, b, repeats=2); t1 = time.time()
r = A @ x - b
print("n", n, "T", (n + 63) // 64, "gpu ms/solve", ms / 2, "wall", round(t1 - t0, 2), "rel resid", np.linalg.norm(r) / np.linalg.norm(b))
