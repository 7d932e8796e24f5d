#!/usr/bin/env python3
"""The drop-in single call (compute_local_real_energy through the resident server, its moment path included) on seeded random
systems against the C oracle: the caller's pattern of mc_water_translation -- ask, move the molecule on the HOST only, ask again,
sometimes put it back -- 120 steps per system.  Run on the GPU box: python tools/server_soak.py [first_seed] [count]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import test_gpu_fuzz as fz  # noqa: E402
from conftest import RTOL  # noqa: E402
from oracle import COracle  # noqa: E402
from mc_water_ls_mw_amd.energy import load_boxes  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
c = COracle()
bad, skipped, calls = [], 0, 0
for seed in range(first, first + count):
    rng = np.random.default_rng(31000 + seed)
    kind, h, x = fz.random_system(rng)
    n = len(x)
    iv = c.ivects(h)
    if n < 2 or len(iv) > 1000:
        skipped += 1
        continue
    nn, jn, vn = c.neighbours(x, iv, 64)
    if nn.max() > 64:
        skipped += 1
        continue
    em = load_boxes([h], [x], maxneigh=64)
    try:
        xyz = np.array(x)
        worst = 0.0
        for step in range(120):
            imol = int(rng.integers(1, n + 1))
            for phase in (0, 1):
                if phase == 1:
                    disp = rng.normal(0, 0.35, 3)
                    em.ljr[0, imol - 1] += disp
                    xyz[imol - 1] += disp
                e = em.compute_local_real_energy(imol, 1)
                r = c.local_energy(imol, xyz, iv, nn, jn, vn)
                calls += 1
                # the suite's bar (tests/test_gpu_fuzz.py): 1e-10 relative with a floor of 1e-14 Ha -- a molecule with one distant
                # neighbour has an energy of 1e-6 Ha and below; one pushed on top of another (E ~ r^-4, above 1 Ha) is left out
                err = max(0.0, abs(e - r) - 1e-14) / max(abs(r), 1e-300) if r != 0.0 else abs(e)
                worst = max(worst, err if abs(r) < 1.0 else 0.0)
            if step % 3 == 1:                                      # reject: the host reverts silently
                em.ljr[0, imol - 1] -= disp
                xyz[imol - 1] -= disp
            if step % 40 == 39:
                em.compute_model_energy(1)                         # an exclusive entry point: the server stops and starts again
        if not worst <= RTOL:
            bad.append((seed, kind, n, worst))
    finally:
        em.energy_deinit()
    if (seed - first) % 25 == 24:
        print(f"... {seed - first + 1} systems, {calls} calls, {len(bad)} failures, {skipped} skipped", flush=True)
print("all clean" if not bad else f"FAILED: {bad}")
sys.exit(1 if bad else 0)
