"""Experiment build of the library with other compile-time constants: python tools/build_variant.py <name> <file.hip> -D...  ->
slide_slam_amd/_lib/<name>.so (the named source recompiled with the extra flags, every other object as built).  Loaded with
SLIDE_LIB_VARIANT=<name> (slide_slam_amd/api.py).  Kernel tuning only."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slide_slam_amd import build as B
name, src = sys.argv[1], sys.argv[2]
flags = sys.argv[3:]
B.build()
objs = []
for n, extra in B.SOURCES:
    obj = os.path.join(B.OUT_DIR, n.replace(".hip", ".o"))
    if n == src:
        obj = os.path.join(B.OUT_DIR, n.replace(".hip", f".{name}.o"))
        subprocess.run([B.HIPCC, *B.COMMON, *extra, *flags, "-c", os.path.join(B.CSRC, n), "-o", obj], check=True)
    objs.append(obj)
out = os.path.join(B.OUT_DIR, name + ".so")
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-o", out, *objs], check=True)
print(out)
