#!/usr/bin/env python3
"""64 x 32768-molecule boxes: a few list rebuilds and full-energy launches for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from mc_water_ls_mw_amd import lattice as lat
from mc_water_ls_mw_amd.energy import load_boxes
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
h, x = lat.ice_box("ih", (16, 16, 16), 0.15, seed=20250228)
ideal = lat.ice_box("ih", (16, 16, 16), 0.0)[1]
xs = [x] + [lat.thermalise(ideal, 0.15, 500 + b) for b in range(1, B)]
em = load_boxes([h] * B, xs)
for _ in range(30):
    em.build_neighbours_launch(1, B)
    em.model_energy_launch(1, B)
em.sync()
em.energy_deinit()
