#!/usr/bin/env python3
"""Diagnostic variants of libmw_hip.so, built next to this file (git-ignored, they travel with gpurun) and loaded with MW_HIP_LIB:

    python tools/variants.py stamps     # -DMW_SWEEP_STAMPS: cycle stamps inside k_sweep (tools/sweep_stamps.py)
    python tools/variants.py spill      # -DMW_SWEEP_WAVES_CAP=5: every k_sweep build capped at 96 vector registers, i.e. SPILLING
    python tools/variants.py <name> -DFOO=1 ...   # any other set of flags
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mc_water_ls_mw_amd import build as mwbuild  # noqa: E402

KNOWN = {"stamps": ["-DMW_SWEEP_STAMPS"], "spill": ["-DMW_SWEEP_WAVES_CAP=5"],
         "nodecide": ["-DMW_ABL_NODECIDE"], "noeval": ["-DMW_ABL_NOEVAL"]}


def path(name):
    return os.path.join(ROOT, "tools", "variants", f"libmw_hip_{name}.so")


def build(name, flags=None, force=False, save_temps=False):
    os.makedirs(os.path.dirname(path(name)), exist_ok=True)
    flags = list(flags if flags else KNOWN[name])
    if save_temps:
        flags += ["-save-temps=obj"]
    return mwbuild.build(force=force, verbose=True, extra_flags=flags, out=path(name))


if __name__ == "__main__":
    name = sys.argv[1]
    extra = [a for a in sys.argv[2:] if a.startswith("-") and a not in ("--force", "--save-temps")]
    print(build(name, extra or None, force="--force" in sys.argv, save_temps="--save-temps" in sys.argv))
