import sys, ctypes as C, numpy as np, glob, os
sys.path.insert(0,'.')
rng=np.random.default_rng(0); n=640
Q,_=np.linalg.qr(rng.normal(size=(n,n))); A=np.asfortranarray((Q*np.exp(rng.uniform(0,5,n)))@Q.T); b=rng.normal(size=n)
for f in sorted(glob.glob('slide_slam_amd/_lib/exp_*.so')):
    L=C.CDLL(os.path.abspath(f)); x=np.zeros(n); ms=C.c_double(0)
    L.slide_dense_spd_solve(A.ctypes.data_as(C.c_void_p), C.c_int(n), b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_int(3), C.byref(ms))
    out=(C.c_ulonglong*40)(); L.slide_debug_stamps(out)
    t=np.array(out[:15],dtype=np.float64); t0=t[0]
    names=["start","all waves done(barrier)","trsm done","chain: loop start","ph0 end","ph1 start","ph1 end","ph2 start","ph2 end","ph3 start","chain end","w1: it_done(3) seen","w1: col posted","w1: D handed","panel X3 stored"]
    order=[0,3,4,11,12,13,5,6,7,8,9,10,14,1,2]
    tt=np.array(out[:40],dtype=np.float64)-t0
    print("w2 handoff", int(tt[36]), "w1: staged", int(tt[38]), "barrier", int(tt[39]), "upfront done", int(tt[37]));    print("iter1: start", int(tt[24]), "lanes read", int(tt[25]), "mop", int(tt[26]), "xm", int(tt[27]), "posted", int(tt[28]), "updated", int(tt[29]));    print("chain posts it0..3:", tt[32:36].astype(int), " w1 sees it0..2:", tt[16:19].astype(int), "it3:", int(tt[11]), " w3 sees it0..11:", tt[20:32].astype(int))
    print(os.path.basename(f), " | ".join("%s %d"%(names[i], t[i]-t0) for i in order), "err", np.abs(A@x-b).max())
