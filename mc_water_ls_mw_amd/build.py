"""Build libmw_hip.so (the C-ABI engine) and libmw_comms.so (the RCCL exchange layer for a Fortran host, include/mw_comms.h)
in-tree with hipcc for gfx950.

    python -m mc_water_ls_mw_amd.build [--force]

hipcc cross-compiles gfx950 without a GPU, so this also runs in the build
container.  The .so stays inside the package directory (git-ignored, but it
travels to the GPU box with the gpurun snapshot).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmw_hip.so")
SOURCES = [os.path.join(CSRC, "mw_api.hip")]
DEPS = SOURCES + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip.h")) + \
    [os.path.join(os.path.dirname(PKG), "include", "mw_energy.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


COMMS_LIB = os.path.join(PKG, "libmw_comms.so")
COMMS_SRC = os.path.join(CSRC, "mw_comms.hip")
COMMS_DEPS = [COMMS_SRC, os.path.join(os.path.dirname(PKG), "include", "mw_comms.h")]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm is required to build libmw_hip.so)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in DEPS)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """`out`: another file name for a variant of the library (diagnostic builds, A/B measurements with MW_HIP_LIB)."""
    lib = out or LIB
    if not force and out is None and not needs_build():
        return LIB
    if out is not None and not force and os.path.exists(out) and all(os.path.getmtime(p) <= os.path.getmtime(out) for p in DEPS):
        return out
    cmd = [hipcc_path(), *HIPCC_FLAGS, *extra_flags, "-o", lib, *SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


def build_comms(force=False, verbose=False):
    """libmw_comms.so: host code only (no kernels), linked against RCCL."""
    if not force and os.path.exists(COMMS_LIB) and all(os.path.getmtime(p) <= os.path.getmtime(COMMS_LIB) for p in COMMS_DEPS):
        return COMMS_LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", COMMS_LIB, COMMS_SRC,
           "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return COMMS_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
    print(build_comms(force="--force" in sys.argv, verbose=True))
