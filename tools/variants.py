#!/usr/bin/env python3
"""Diagnostic variants of libmw_hip.so, built next to this file (git-ignored, they travel with gpurun) and loaded with MW_HIP_LIB:

    python tools/variants.py stamps     # -DMW_SWEEP_STAMPS: cycle stamps inside k_sweep (tools/sweep_stamps.py)
    python tools/variants.py spill      # -DMW_SWEEP_WAVES_CAP=5: every k_sweep build capped at 96 vector registers, i.e. SPILLING
    python tools/variants.py nodecide noeval dnobin ...   # several known variants at once (tools/sweep_ablate.sh)
    python tools/variants.py <name> -DFOO=1 ...   # any other set of flags
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mc_water_ls_mw_amd import build as mwbuild  # noqa: E402

KNOWN = {"stamps": ["-DMW_SWEEP_STAMPS"], "spill": ["-DMW_SWEEP_WAVES_CAP=5"],
         "nodecide": ["-DMW_ABL_NODECIDE"], "noeval": ["-DMW_ABL_NOEVAL"],
         # parts of a decision taken out (the numbers are then wrong: shares of the time only)
         "dnobin": ["-DMW_ABL_D_NOBIN"], "dnoswexp": ["-DMW_ABL_D_NOSWEXP"], "dnoexp": ["-DMW_ABL_D_NOEXP"], "dnowl": ["-DMW_ABL_D_NOWL"],
         "dnosw": ["-DMW_ABL_D_NOSW"],
         "dnoall": ["-DMW_ABL_D_NOBIN", "-DMW_ABL_D_NOSWEXP", "-DMW_ABL_D_NOEXP", "-DMW_ABL_D_NOWL", "-DMW_ABL_D_NOSW"],
         "frame": ["-DMW_ABL_NODECIDE", "-DMW_ABL_NOEVAL"],
         # a volume move's split full-box energy checked against the one-wavefront routine inside the kernel (stamps 43..47)
         "splitcheck": ["-DMW_SWEEP_STAMPS", "-DMW_SPLIT_CHECK"]}


def path(name):
    return os.path.join(ROOT, "tools", "variants", f"libmw_hip_{name}.so")


def build(name, flags=None, force=False, save_temps=False):
    os.makedirs(os.path.dirname(path(name)), exist_ok=True)
    flags = list(flags if flags else KNOWN[name])
    if save_temps:
        flags += ["-save-temps=obj"]
    return mwbuild.build(force=force, verbose=True, extra_flags=flags, out=path(name))


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if not a.startswith("-")]
    extra = [a for a in sys.argv[2:] if a.startswith("-") and a not in ("--force", "--save-temps")]
    for name in names:        # (several known names at once; extra flags go with a single name)
        print(build(name, (extra or None) if len(names) == 1 else None, force="--force" in sys.argv, save_temps="--save-temps" in sys.argv))
