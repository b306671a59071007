import sys, ctypes as C, numpy as np, glob, os
sys.path.insert(0,'.')
rng=np.random.default_rng(0); n=640
Q,_=np.linalg.qr(rng.normal(size=(n,n))); A=np.asfortranarray((Q*np.exp(rng.uniform(0,5,n)))@Q.T); b=rng.normal(size=n)
for f in sorted(glob.glob('slide_slam_amd/_lib/exp_*.so')):
    L=C.CDLL(os.path.abspath(f)); x=np.zeros(n); ms=C.c_double(0)
    L.slide_dense_spd_solve(A.ctypes.data_as(C.c_void_p), C.c_int(n), b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_int(3), C.byref(ms))
    out=(C.c_ulonglong*16)(); L.slide_debug_stamps(out)
    t=np.array(out[:7],dtype=np.float64)
    print(os.path.basename(f), "phases (cycles): loads %d update %d exchange %d factor %d publish %d trsm %d total %d"%tuple(list(np.diff(t))+[t[6]-t[0]]), "err", np.abs(A@x-b).max())
