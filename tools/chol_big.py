# Dense SPD solve at the bench's reduced-system size through the C-ABI (diagnostic for rocprofv3 kernel traces).
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import slide_slam_amd as s
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3776
rng = np.random.default_rng(0)
G = rng.normal(size=(n, n))
A = G @ G.T / n + np.eye(n)
b = rng.normal(size=n)
x, ms = s.dense_spd_solve(A, b, repeats=5)
print("n", n, "ms/solve", ms / 5, "resid", np.abs(A @ x - b).max())
