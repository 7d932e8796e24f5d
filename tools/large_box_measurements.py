#!/usr/bin/env python3
"""configs[4] alone: list rebuild and full-box energy of B x 32768-molecule boxes (bench.py's `secondary` entry, for A/B runs
with MW_HIP_LIB inside one GPU session).  Run on the GPU box: python tools/large_box_measurements.py [B]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import bench  # noqa: E402
from mc_water_ls_mw_amd import lattice as lat  # noqa: E402
from mc_water_ls_mw_amd.energy import load_boxes  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
h, x = lat.ice_box("ih", (16, 16, 16), 0.15, seed=20250228)
ideal = lat.ice_box("ih", (16, 16, 16), 0.0)[1]
xs = [x] + [lat.thermalise(ideal, 0.15, 500 + b) for b in range(1, B)]
em = load_boxes([h] * B, xs)
ms_list = min(bench.timed(em, 0, lambda: em.build_neighbours_launch(1, B), 5) for _ in range(3))
ms_full = min(bench.timed(em, 1, lambda: em.model_energy_launch(1, B), 10) for _ in range(3))
e = em.model_energy_fetch(1, 1)[0]
gold = os.path.join(ROOT, "tests", "golden", "ih32768_t015.npz")
ref = float(np.load(gold)["model_energy"])
print(json.dumps({"boxes": B, "list_rebuild_ms": ms_list, "full_energy_ms": ms_full, "box1_rel_err_vs_golden": abs(e - ref) / abs(ref)}))
em.energy_deinit()
