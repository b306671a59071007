import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
import slide_slam_amd as s
rng=np.random.default_rng(0); n=640
Q,_=np.linalg.qr(rng.normal(size=(n,n))); A=(Q*np.exp(rng.uniform(0,5,n)))@Q.T
x,ms=s.dense_spd_solve(A, rng.normal(size=n), repeats=3)
out=(C.c_ulonglong*16)(); s.lib().slide_debug_stamps(out)
t=np.array(out[:7],dtype=np.float64); print("phases (cycles): loads %d update %d reshape %d factor %d publish+inv16 %d trsm %d total %d"%tuple(list(np.diff(t))+[t[6]-t[0]]), "ms", ms)
