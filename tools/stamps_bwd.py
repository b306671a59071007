import sys, ctypes as C, numpy as np, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
rng = np.random.default_rng(0); n = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
G = rng.normal(size=(n, n)); A = np.asfortranarray(G @ G.T / n + np.eye(n)); b = rng.normal(size=n)
L = C.CDLL(os.path.abspath('slide_slam_amd/_lib/exp_stamps.so')); x = np.zeros(n); ms = C.c_double(0)
L.slide_dense_spd_solve(A.ctypes.data_as(C.c_void_p), C.c_int(n), b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_int(3), C.byref(ms))
out = (C.c_ulonglong * 40)(); L.slide_debug_stamps(out)
t = np.array(out[:40], dtype=np.float64); t -= t[0]
print("M built", int(t[1]))
for k in range(4): print("pair", k, "poll start", int(t[2 + 3 * k]), "poll end", int(t[3 + 3 * k]), "fma end", int(t[4 + 3 * k]))
print("ys written", int(t[14]), "barrier", int(t[15]), "xA stored", int(t[16]), "barrier(h1)", int(t[17]), "xB stored(h1)", int(t[18]), "err", np.abs(A @ x - b).max())
