"""Builds libgm3d_hip.so (the C-ABI HIP back end) in-tree with hipcc for gfx950.

`python -m gm3d_amd.build` or `gm3d_amd.build.build()`.  hipcc cross-compiles without a
GPU; the resulting .so is git-ignored but travels with the working tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libgm3d_hip.so")
ARCH = "gfx950"

# Index-exact kernels must not contract a*b+c into FMA (oracle contract); the MFMA
# kernels have no such constraint.
SOURCES = {
    "capi.hip": [],
    "fps.hip": ["-ffp-contract=off"],
    "knn.hip": ["-ffp-contract=off"],
    "chamfer.hip": ["-ffp-contract=off"],
    "attention.hip": [],
    "rowops.hip": [],
    "embed.hip": [],
    "optim.hip": [],
    "gemm.hip": [],
    "gemm_nt.hip": [],
    "gemm_ring.hip": [],
    "gemm_dma.hip": [],
    "gemm_ws.hip": [],
    "gather.hip": [],
    "attention_masked.hip": [],
}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT_DIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, "common.hpp"), os.path.join(HERE, "..", "include", "gm3d.h")]
    objs = []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(OUT_DIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
