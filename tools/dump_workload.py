#!/usr/bin/env python3
"""Write bench.py's workload (ih4096_t015 walkers + trial moves) as one binary file for tools/kbench.

    python tools/dump_workload.py OUT.bin [--walkers 512] [--moves 2048] [--reps 8] [--sigma 0.15]

Layout (little endian): int32 nwalkers, N, moves; float64 h[9]; float64 pos[nwalkers][N][3];
int32 imol[nwalkers*moves] (1-based); float64 trial[nwalkers*moves][3].
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mc_water_ls_mw_amd import lattice as lat  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--walkers", type=int, default=512)
    ap.add_argument("--moves", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--sigma", type=float, default=0.15)
    a = ap.parse_args()
    h0, x0 = lat.ice_box("ih", (a.reps,) * 3, 0.0)
    with open(a.out, "wb") as fh:
        np.array([a.walkers, len(x0), a.moves], dtype=np.int32).tofile(fh)
        np.asarray(h0, dtype=np.float64).tofile(fh)
        xs = []
        for w in range(a.walkers):
            x = lat.thermalise(x0, a.sigma, 20250228 + w)
            xs.append(x)
            x.tofile(fh)
        imols, trials = [], []
        for w in range(a.walkers):
            im, tr = lat.trial_moves(xs[w], a.moves, seed=1 + w)
            imols.append(im)
            trials.append(tr)
        np.concatenate(imols).astype(np.int32).tofile(fh)
        np.concatenate(trials).astype(np.float64).tofile(fh)


if __name__ == "__main__":
    main()
