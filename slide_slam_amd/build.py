"""Builds slide_slam_amd/_lib/libslide_gpu.so (HIP kernels for gfx950 + the C-ABI of include/slide_gpu.h).

hipcc cross-compiles without a GPU; the .so stays in-tree (git-ignored, shipped by gpurun).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_lib")
LIB = os.path.join(OUT_DIR, "libslide_gpu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off where results are compared against thresholds or must round like the reference's
# un-fused x86-64 arithmetic (association ids, SlideMatch inlier counts, host-side pose composition);
# the solver / Cholesky kernels keep hipcc's default contraction.
SOURCES = [
    ("solver_kernels.hip", []),
    # VGPR-form MFMA: the factorisation edits single registers of accumulator tiles between MFMAs; with the default
    # AGPR destinations every such edit costs v_accvgpr copies and hazard nops
    ("chol_kernels.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form"]),
    ("pcg_kernels.hip", []),
    ("assoc_kernels.hip", ["-ffp-contract=off"]),
    ("place_kernels.hip", ["-ffp-contract=off"]),
    ("clipper_kernels.hip", ["-ffp-contract=off"]),
    ("host_graph.hip", ["-ffp-contract=off"]),
    ("host_backend.hip", ["-ffp-contract=off"]),
    ("capi.hip", ["-ffp-contract=off"]),
    ("wire.hip", ["-ffp-contract=off"]),     # host code only: sloam_msgs wire codec + rosbag reader (include/slide_wire.h)
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-invalid-offsetof",
          "-Wno-unused-result"]


def _deps_newer(obj: str, src: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    deps.append(os.path.join(HERE, "..", "include", "slide_gpu.h"))
    deps.append(os.path.join(HERE, "..", "include", "slide_wire.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, stamps: bool = False) -> str:
    """stamps: an EXPERIMENT copy of the library (_lib/exp_stamps.so, -DSLIDE_STAMPS: in-kernel time stamps read back through
    slide_debug_*_stamps; tools/stamps2.py, tools/assoc_stamps.py) — never loaded by the package."""
    os.makedirs(OUT_DIR, exist_ok=True)
    objs = []
    procs = []
    lib_path = os.path.join(OUT_DIR, "exp_stamps.so") if stamps else LIB
    for name, extra in SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(OUT_DIR, name.replace(".hip", ".stamps.o" if stamps else ".o"))
        if stamps:
            extra = [*extra, "-DSLIDE_STAMPS", *(["-DSLIDE_STAMP_BLOCK=" + os.environ["SLIDE_STAMP_BLOCK"]] if os.environ.get("SLIDE_STAMP_BLOCK") else [])]
        objs.append(obj)
        if force or _deps_newer(obj, src):
            cmd = [HIPCC, *COMMON, *extra, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for name, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {name} ---\n{out}\n")
        elif verbose and out.strip():
            print(f"--- {name} ---\n{out}")
    if failed:
        raise RuntimeError("hipcc failed")
    if procs or not os.path.exists(lib_path):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-o", lib_path, *objs]
        subprocess.run(cmd, check=True)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, stamps="--stamps" in sys.argv))
