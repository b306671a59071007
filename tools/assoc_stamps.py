# Phase timing of the association kernel (workgroup 0) from the -DSLIDE_STAMPS experiment build:
#   python -m slide_slam_amd.build --stamps && python3 tools/assoc_stamps.py
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "slide_slam_amd", "_lib", "exp_stamps.so"))
rng = np.random.default_rng(1)
n_map, K, n_obs, nq = 10000, 1000, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 1
model = np.column_stack([rng.uniform(0, 440, n_map), rng.uniform(0, 220, n_map), rng.normal(0, 0.3, n_map)])
cloud = np.ascontiguousarray(model.astype(np.float32))
label = rng.integers(1, 7, n_map).astype(np.int32)
pick = rng.integers(0, n_map, (nq, n_obs))
qpos = np.ascontiguousarray(np.column_stack([model[pick[:, 0], :2], np.full(nq, 2.0)]))
obs = np.ascontiguousarray(model[pick]); olab = np.ascontiguousarray(label[pick])
out = np.zeros((nq, n_obs), np.int32); ms = C.c_double(0)
P = lambda a: a.ctypes.data_as(C.c_void_p)
rc = L.slide_assoc_sweep_batch(P(cloud), P(model), P(label), C.c_int(n_map), P(qpos), P(obs), P(olab), C.c_int(nq), C.c_int(n_obs), C.c_int(K),
                               C.c_double(0.75), P(out), C.c_int(1), C.byref(ms))
st = (C.c_ulonglong * 16)(); L.slide_debug_assoc_stamps(st)
t = np.array(st[:7], dtype=np.float64)
names = ["start", "distance words in LDS", "radix select done", "compaction + padding", "sort done", "candidates staged", "matching done"]
print("rc", rc, "launch ms", ms.value, "n_query", nq)
for i in range(1, 7):
    print(f"  {names[i]:28s} +{(t[i] - t[i-1]) * 0.01:8.2f} us   (at {(t[i] - t[0]) * 0.01:8.2f} us)")
x = np.array(st[12:15], dtype=np.float64)
if x[0] > 0:
    print("  select: histogram over linear bins done at %.2f us, K-th key's bin found +%.2f us, lower bins + candidates placed +%.2f us, ranks + padding +%.2f us" %
          ((x[0] - t[0]) * 0.01, (x[1] - x[0]) * 0.01, (x[2] - x[1]) * 0.01, (t[3] - x[2]) * 0.01))
if st[7] and st[15]:
    print("  staging by label: gathers issued + groups found at +%.2f us, labels mapped + counted +%.2f us, scanned + placed +%.2f us" %
          ((st[7] - t[4]) * 0.01, (st[15] - st[7]) * 0.01, (t[5] - st[15]) * 0.01))
w = np.array(st[8:12], dtype=np.float64)
print("  matching, wave 0: detection data loaded +%.2f us, candidate loop +%.2f us, roots + reduction +%.2f us (first at %.2f us)" %
      ((w[1] - w[0]) * 0.01, (w[2] - w[1]) * 0.01, (w[3] - w[2]) * 0.01, (w[0] - t[0]) * 0.01))
