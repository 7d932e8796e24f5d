#!/usr/bin/env python3
"""How the per-launch time of the full-box kernel depends on the number of back-to-back launches timed (GPU clocks)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from mc_water_ls_mw_amd import lattice as lat
from mc_water_ls_mw_amd.energy import load_boxes
h, x = lat.ice_box("ih", (16, 16, 16), 0.15, seed=20250228)
ideal = lat.ice_box("ih", (16, 16, 16), 0.0)[1]
B = 64
xs = [x] + [lat.thermalise(ideal, 0.15, 500 + b) for b in range(1, B)]
em = load_boxes([h] * B, xs)
em.build_neighbours_launch(1, B); em.sync()
for reps in (2, 5, 10, 20, 50, 100, 200, 10, 5):
    em.sync()
    em.timer_start(1)
    for _ in range(reps):
        em.model_energy_launch(1, B)
    em.timer_stop(1)
    print("32768 x 64: reps", reps, "ms/launch", em.timer_ms(1) / reps, flush=True)
    time.sleep(0.5)
em.energy_deinit()
