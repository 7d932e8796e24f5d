#!/usr/bin/env python3
"""Throughput of the device-resident translation-move driver (many walkers, one wavefront each).
Run on the GPU box: python tools/sweep_measurements.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401

from mc_water_ls_mw_amd import lattice as lat  # noqa: E402
from mc_water_ls_mw_amd.energy import EnergyModule  # noqa: E402
from mc_water_ls_mw_amd.sweep import MuGrid, WalkerFarm  # noqa: E402


def farm_for(cells, nlat, walkers, sigma, temperature, mu_range=8000.0):
    n = len(cells[0][1])
    em = EnergyModule(n, walkers * nlat)
    for w in range(walkers):
        for l in range(nlat):
            em.hmatrix[w * nlat + l] = cells[l][0]
            em.ljr[w * nlat + l] = lat.thermalise(cells[l][1], sigma, 1000 * l + w)
    em.setup_boxes()
    em.build_neighbours_batch(1, walkers * nlat)
    em.model_energy_batch(1, walkers * nlat)
    grid = MuGrid(101, -mu_range, mu_range) if nlat == 2 else None
    farm = WalkerFarm(em, nlat, temperature, 1.1, grid=grid)
    farm.set_states(1)
    return em, farm


def run(name, cells, nlat, walkers, nmoves, out, wl=False, npt=False, sigma=0.1, mu_range=8000.0):
    em, farm = farm_for(cells, nlat, walkers, sigma, 200.0, mu_range)
    if wl:            # what the replica farm runs per move: Wang-Landau update + a lattice-switch attempt (farm.py set_options)
        farm.options(record=True, samplerun=False, always_switch=True, npt=npt, wl_factor=float(os.environ.get("MW_SWEEP_WLF", "0.05")))
    if npt:           # io.f90:171-172: volume moves 1/N against translations 0.5
        farm.moves(trans_prob=0.5, vol_prob=1.0 / len(cells[0][1]), dv_max_ang=0.924)
    pre = int(os.environ.get("MW_SWEEP_PRE", "20"))         # moves before the timed launch (a long run-in shows what the
    farm.sweep_launch(pre, seed=1, move0=0)                 # rate does once the walkers have left their starting configurations)
    em.sync()
    em.timer_start(0)
    farm.sweep_launch(nmoves, seed=1, move0=pre)
    em.timer_stop(0)
    ms = em.timer_ms(0)
    acc = np.mean([farm.state(w)["accepted"] for w in range(1, min(walkers, 64) + 1)]) / (nmoves + pre)
    # drift check of walker 1 against a fresh full-box energy (the reference's own consistency test)
    st = farm.state(1)
    fresh = em.model_energy_batch(1, nlat)
    out[name] = {"walkers": walkers, "lattices": nlat, "molecules": len(cells[0][1]), "moves_per_walker": nmoves,
                 "ms": ms, "moves_per_s": walkers * nmoves / (ms * 1e-3),
                 "local_energy_evaluations_per_s": walkers * nmoves * 2 * nlat / (ms * 1e-3),
                 "us_per_move_per_walker": ms * 1e3 / nmoves, "acceptance": acc,
                 "drift_walker1_Ha": [st["model_energy"][l] - fresh[l] for l in range(nlat)]}
    em.energy_deinit()


only = os.environ.get("MW_SWEEP_CASE")
out = {}
g = lambda n: dict(np.load(os.path.join(ROOT, "tests", "golden", n + ".npz")))  # noqa: E731
ic48, ih48 = g("ic48"), g("ih48")
if only in (None, "pair48"):
    run("pair48 x 8192 walkers", [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])], 2, 8192, 480, out)
if only in (None, "pair48wl"):
    run("pair48 x 8192 walkers, WL update + switch per move", [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])], 2, 8192, 480, out, wl=True)
if only in (None, "npt48"):
    run("pair48 x 8192 walkers, NPT (volume moves), WL update + switch per move", [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])], 2, 8192, 480, out,
        wl=True, npt=True)
if only in ("farm48",):     # the replica farm's own settings (farm.py defaults): +-400 window, so walkers do switch lattice
    run("pair48 x 8192 walkers, WL update + switch per move, farm window", [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])], 2, 8192, 480, out,
        wl=True, sigma=0.05, mu_range=400.0)
if only in ("farm48npt",):
    run("pair48 x 8192 walkers, NPT, WL update + switch per move, farm window", [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])], 2, 8192, 480, out,
        wl=True, npt=True, sigma=0.05, mu_range=400.0)
if only in ("eight48",):
    run("pair48 x 8 walkers, WL update + switch per move", [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])], 2, 8, 4800, out,
        wl=True, sigma=0.05, mu_range=400.0)
# a handful of the reference's own walkers (it runs one per MPI rank): the speed of ONE chain, look-ahead inside it
if only in ("one48",):
    run("pair48 x 1 walker, WL update + switch per move", [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])], 2, 1, 4800, out,
        wl=True, sigma=0.05, mu_range=400.0)
if only in ("one48plain",):
    run("pair48 x 1 walker", [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])], 2, 1, 4800, out, sigma=0.05, mu_range=400.0)
if only in ("one48single",):
    run("ic48 x 1 walker (one lattice)", [(ic48["h"], ic48["xyz"])], 1, 1, 4800, out, sigma=0.05)
if only in ("one48npt",):
    run("pair48 x 1 walker, NPT, WL update + switch per move", [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])], 2, 1, 4800, out,
        wl=True, npt=True, sigma=0.05, mu_range=400.0)
if only in ("eight48npt",):
    run("pair48 x 8 walkers, NPT, WL update + switch per move", [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])], 2, 8, 4800, out,
        wl=True, npt=True, sigma=0.05, mu_range=400.0)
if only in ("few48npt",):
    run("pair48 x 256 walkers, NPT, WL update + switch per move", [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])], 2, 256, 2400, out,
        wl=True, npt=True, sigma=0.05, mu_range=400.0)
if only is not None and only.startswith("n48npt_"):       # n48npt_<walkers>: where look-ahead stops paying for small walkers
    nwalk = int(only.split("_")[1])
    run(f"pair48 x {nwalk} walkers, NPT, WL update + switch per move", [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])], 2, nwalk, 1200, out,
        wl=True, npt=True, sigma=0.05, mu_range=400.0)
if only is not None and only.startswith("n48wl_"):
    nwalk = int(only.split("_")[1])
    run(f"pair48 x {nwalk} walkers, WL update + switch per move", [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])], 2, nwalk, 1200, out,
        wl=True, sigma=0.05, mu_range=400.0)
ic1536, ih1536 = g("ic1536"), g("ih1536")
if only in (None, "pair1536"):
    run("pair1536 x 2048 walkers", [(ic1536["h"], ic1536["xyz"]), (ih1536["h"], ih1536["xyz"])], 2, 2048, 300, out)
h, x = lat.ice_box("ih", (8, 8, 8), 0.0)
if only in (None, "ih4096"):
    run("ih4096 x 2048 walkers", [(h, x)], 1, 2048, 300, out)
# few walkers: the chip is filled from inside the chains (look-ahead, MW_SWEEP_AHEAD)
if only in ("few4096",):
    run("ih4096 x 64 walkers", [(h, x)], 1, 64, 600, out)
if only in ("many4096",):   # more walkers than four wavefronts per SIMD hold: what a fifth (<= 96 VGPRs) is worth
    run("ih4096 x 6144 walkers", [(h, x)], 1, 6144, 200, out)
if only is not None and only.startswith("n4096_"):          # n4096_<walkers>: where the moment path of walkers in global memory starts to pay
    run(f"ih4096 x {int(only.split('_')[1])} walkers", [(h, x)], 1, int(only.split("_")[1]), 300, out)
if only in ("one4096",):
    run("ih4096 x 1 walker", [(h, x)], 1, 1, 2000, out)
if only in ("one1536",):
    run("pair1536 x 1 walker, WL update + switch per move", [(ic1536["h"], ic1536["xyz"]), (ih1536["h"], ih1536["xyz"])], 2, 1, 2000, out, wl=True)
if only in ("few1536",):
    run("pair1536 x 32 walkers, WL update + switch per move", [(ic1536["h"], ic1536["xyz"]), (ih1536["h"], ih1536["xyz"])], 2, 32, 600, out, wl=True)
print(json.dumps(out, indent=1))
