# Where a block of the chained backward substitution spends its time (experiment build, -DSLIDE_STAMPS):
#   python -m slide_slam_amd.build --stamps && python3 tools/chain_stamps.py [n]
# Blocks 30 and 31 of the chain stamp the 100 MHz wall clock: 31 waits for what 30 publishes.
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "slide_slam_amd", "_lib", "exp_stamps.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3776
rng = np.random.default_rng(0)
G = rng.normal(size=(n, n))
A = np.asfortranarray(G @ G.T / n + np.eye(n))
b = rng.normal(size=n)
x = np.zeros(n)
ms = C.c_double(0)
P = lambda a: a.ctypes.data_as(C.c_void_p)
rc = L.slide_dense_spd_solve(P(A), C.c_int(n), P(b), P(x), C.c_int(3), C.byref(ms))
st = (C.c_ulonglong * 32)()
L.slide_debug_chain_stamps(st)
t = np.array(st[:32], dtype=np.float64) * 10.0     # ns
a, bb = t[:16], t[16:]
print("rc", rc, "ms per solve", ms.value / 3, "residual", np.abs(A @ x - b).max())
for name, u in (("block 30", a), ("block 31", bb)):
    print(f"{name}: entry -> inverse built {u[1]-u[0]:7.0f} ns | last-but-one x seen at {u[7]-u[0]:7.0f} | last poll begins {u[2]-u[0]:7.0f}, "
          f"x seen {u[3]-u[0]:7.0f} | tile product done +{u[4]-u[3]:5.0f} | rhs in LDS +{u[5]-u[4]:5.0f} | published +{u[6]-u[5]:5.0f}")
print(f"hop: block 30 publishes -> block 31 sees it: {bb[3]-a[6]:6.0f} ns;  block 31 publishes {bb[6]-a[6]:6.0f} ns after block 30 (= time per block)")
