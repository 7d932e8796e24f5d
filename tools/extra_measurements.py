#!/usr/bin/env python3
"""Measurements that are not the bench line but belong in DESIGN.md / profiles/:
  * latency of the drop-in single calls through the C ABI (launch + sync + PCIe inclusive)
  * BASELINE.json configs[4]: 32768-molecule ice Ih, neighbour-list rebuild + full energy
  * BASELINE.json configs[2]: the 1536-molecule Ih<->Ic lattice-switch pair, single-move path
Prints one JSON object.  Run on the GPU box:  python tools/extra_measurements.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (one HIP runtime per process: torch first)

from mc_water_ls_mw_amd import lattice as lat  # noqa: E402
from mc_water_ls_mw_amd.energy import load_boxes  # noqa: E402

out = {}

# ---- single-call latency on a 4096 box ----------------------------------------------------
h, x = lat.ice_box("ih", (8, 8, 8), 0.15, seed=20250228)
em = load_boxes([h], [x])
rng = np.random.default_rng(1)
for _ in range(200):
    em.compute_local_real_energy(int(rng.integers(1, 4097)), 1)
n = 3000
t0 = time.perf_counter()
for _ in range(n):
    em.compute_local_real_energy(int(rng.integers(1, 4097)), 1)
t_loc = (time.perf_counter() - t0) / n
n = 300
t0 = time.perf_counter()
for _ in range(n):
    em.compute_model_energy(1)
t_full = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(50):
    em.compute_neighbours(1)
t_nb = (time.perf_counter() - t0) / 50
out["single_call_latency_4096"] = {
    "compute_local_real_energy_us": t_loc * 1e6, "compute_model_energy_us (incl. 96 KiB H2D)": t_full * 1e6,
    "compute_neighbours_us (incl. H2D)": t_nb * 1e6,
    "note": "host wall time per synchronous call through ctypes -> C ABI -> one launch -> stream sync"}
em.energy_deinit()

# ---- 32768 stress box ----------------------------------------------------------------------
h, x = lat.ice_box("ih", (16, 16, 16), 0.15, seed=20250228)
em = load_boxes([h], [x])
N = len(x)
reps = 20
em.sync()
em.timer_start(0)
for _ in range(reps):
    em.build_neighbours_launch(1, 1)
em.timer_stop(0)
ms_list = em.timer_ms(0) / reps
em.timer_start(1)
for _ in range(reps):
    em.model_energy_launch(1, 1)
em.timer_stop(1)
ms_full = em.timer_ms(1) / reps
entries = em.neighbour_total(1, 1)
p, t = em.model_energy_counts(1)
b_list = N * (24 + 4) + 8 * entries
b_full = N * (24 + 8) + 8 * entries
gold = os.path.join(ROOT, "tests", "golden", "ih32768_t015.npz")
ref = float(np.load(gold)["model_energy"])
e = em.model_energy_fetch(1, 1)[0]
out["stress_32768"] = {
    "list_rebuild_ms": ms_list, "list_algorithmic_GBps": b_list / ms_list / 1e6,
    "full_energy_ms": ms_full, "full_algorithmic_GBps": b_full / ms_full / 1e6,
    "full_interactions_per_s": (p + t) / (ms_full * 1e-3), "rel_err_vs_golden": abs(e - ref) / abs(ref),
    "note": "ONE box per launch (128 workgroups on 256 CUs): latency, not throughput"}
em.energy_deinit()

# ---- 32768 stress boxes, batched: the throughput regime of the large-box path (positions gathered from L2/HBM) --
B = 64
ideal = lat.ice_box("ih", (16, 16, 16), 0.0)[1]
xs = [x] + [lat.thermalise(ideal, 0.15, 500 + b) for b in range(1, B)]
em = load_boxes([h] * B, xs)
em.sync()
em.timer_start(0)
for _ in range(5):
    em.build_neighbours_launch(1, B)
em.timer_stop(0)
ms_list = em.timer_ms(0) / 5
em.model_energy_launch(1, B); em.sync()
em.timer_start(1)
for _ in range(10):
    em.model_energy_launch(1, B)
em.timer_stop(1)
ms_full = em.timer_ms(1) / 10
entries = em.neighbour_total(1, B)
cnt = [em.model_energy_counts(b) for b in range(1, B + 1)]
inter = sum(c[0] + c[1] for c in cnt)
e = em.model_energy_fetch(1, B)
out["stress_32768_batched"] = {
    "boxes": B, "list_rebuild_ms": ms_list, "list_algorithmic_GBps": (B * N * (24 + 4) + 8 * entries) / ms_list / 1e6,
    "full_energy_ms": ms_full, "full_algorithmic_GBps": (B * N * (24 + 8) + 8 * entries) / ms_full / 1e6,
    "full_interactions_per_s": inter / (ms_full * 1e-3), "box1_rel_err_vs_golden": abs(e[0] - ref) / abs(ref),
    "note": "64 boxes per launch; positions do not fit LDS (786 KiB): gathered through L2"}
em.energy_deinit()

# ---- 1536-molecule lattice-switch pair, single-move path -------------------------------------
g1 = np.load(os.path.join(ROOT, "tests", "golden", "ic1536.npz"))
g2 = np.load(os.path.join(ROOT, "tests", "golden", "ih1536.npz"))
W = 256                                   # walkers, each an (Ic, Ih) pair of 1536 molecules
hs, xs = [], []
for w in range(W):
    hs += [g1["h"], g2["h"]]
    xs += [lat.thermalise(g1["xyz"], 0.12, 1000 + w), lat.thermalise(g2["xyz"], 0.12, 2000 + w)]
em = load_boxes(hs, xs)
M = 1536
ils = np.repeat(np.arange(1, 2 * W + 1, dtype=np.int32), M)
imol = np.concatenate([lat.trial_moves(xs[b], M, seed=b)[0] for b in range(2 * W)])
trial = np.concatenate([lat.trial_moves(xs[b], M, seed=b)[1] for b in range(2 * W)])
em.moves_upload(ils, imol, trial)
em.moves_launch(); em.sync()
em.timer_start(2)
for _ in range(10):
    em.moves_launch()
em.timer_stop(2)
ms = em.timer_ms(2) / 10
io, so, inw, sn = em.moves_counts()
out["ls_pair_1536"] = {"walkers": W, "moves_per_lattice": M, "ms_per_launch": ms,
                       "interactions_per_s": (io + inw) / (ms * 1e-3),
                       "move_evaluations_per_s": 2 * len(imol) / (ms * 1e-3),
                       "algorithmic_GBps": (24 * 2 * len(imol) + 32 * (so + sn)) / ms / 1e6}
em.energy_deinit()
print(json.dumps(out, indent=1))
