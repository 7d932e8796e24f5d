#!/usr/bin/env python3
"""Where the time of ONE chain goes inside k_sweep: cycle stamps of walker 0's first wavefront (a diagnostic build of the library,
tools/variants.py stamps).  Run on the GPU box:

    python tools/variants.py stamps && MW_HIP_LIB=tools/variants/libmw_hip_stamps.so python tools/sweep_stamps.py [case ...]

Per case and look-ahead: microseconds per move of each phase of a round (own evaluation, waiting for the other wavefronts, the
decisions, commit), of each part of a decision, and of each stage of move_energy_wave (MW_STAMP k-1 -> k)."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MW_HIP_LIB", os.path.join(ROOT, "tools", "variants", "libmw_hip_stamps.so"))
import torch  # noqa: E402,F401

from mc_water_ls_mw_amd import lattice as lat  # noqa: E402
from mc_water_ls_mw_amd.energy import EnergyModule, load_library  # noqa: E402
from mc_water_ls_mw_amd.sweep import MuGrid, WalkerFarm  # noqa: E402

PHASES = {12: "bulk pass", 0: "round: own evaluation", 1: "round: wait for the other wavefronts", 2: "round: decisions", 3: "round: commit + barrier",
          6: "uniforms", 10: "decide: mu, bins, eta", 11: "decide: exp, accept, state", 12: "decide: post_move", 13: "post_move: WL update",
          14: "post_move: switch", 15: "decide: log + tail"}
STAGES = {17: "eval: pass 0 gathers + in-range", 18: "eval: compaction, marks", 19: "eval: row fetch issue + pair terms + records",
          20: "eval: P items (j-i-k)", 21: "eval: scan", 22: "eval: final flush", 23: "eval: reductions"}


def stamps(reset=True):
    L = load_library()
    buf = (ctypes.c_ulonglong * 48)()
    assert L.mw_debug_sweep_stamps(buf, 48, 1 if reset else 0) == 0
    return np.array(buf[:], dtype=np.float64)


def case(name, cells, nlat, walkers, nmoves, wl=False, npt=False, sigma=0.05, mu_range=400.0):
    out = {}
    for ahead in ("1", "2", "4", "6"):
        os.environ["MW_SWEEP_AHEAD"] = ahead
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
        farm = WalkerFarm(em, nlat, 200.0, 1.1, grid=grid)
        farm.set_states(1)
        if wl:
            farm.options(record=True, samplerun=False, always_switch=True, npt=npt, wl_factor=0.05)
        if npt:
            farm.moves(trans_prob=0.5, vol_prob=1.0 / n, dv_max_ang=0.924)
        farm.sweep_launch(20, seed=1, move0=0)
        em.sync()
        stamps(reset=True)
        em.timer_start(0)
        farm.sweep_launch(nmoves, seed=1, move0=20)
        em.timer_stop(0)
        ms = em.timer_ms(0)
        st = stamps(reset=True)
        cyc_per_us = st[7] / (st[8] / 100.0) if st[8] > 0 else float("nan")     # shader cycles per microsecond (wall clock: 100 MHz)
        moves, rounds = st[4], st[5]
        rec = {"us_per_move_event": ms * 1e3 / nmoves, "shader_MHz": cyc_per_us, "moves_in_translation_rounds": moves, "rounds": rounds,
               "moves_per_round": moves / max(rounds, 1.0), "kernel_us_per_move_stamped": st[7] / cyc_per_us / nmoves}
        for k, label in PHASES.items():
            rec[label + " [us/move]"] = st[k] / cyc_per_us / max(moves, 1.0)
        rec["bulk: calls, moves committed, slots predicted, stop bits"] = [st[16], st[9], st[32], st[33]]
        rec["evaluations of wavefront 0 the moment path declined: all, row > 32, own image listed, > 11 in range, a triplet of the 0.99 rule"] = [st[43], st[44], st[45], st[46], st[47]]
        nv = max(st[40], 1.0)
        rec["volume move [us per volume move]: count, set-up, rescale, ivects+recip, full-box energy, decide, restore+mirror, total in routine, total in branch"] = \
            [st[40]] + [st[k] / cyc_per_us / nv for k in (34, 35, 37, 36, 38, 39, 41, 42)]
        rec["evaluation stages of wavefront 0 [us per evaluation]"] = {label: st[k] / cyc_per_us / max(rounds, 1.0) for k, label in STAGES.items()}
        out["look-ahead " + ahead] = rec
        em.energy_deinit()
    return {name: out}


g = lambda n: dict(np.load(os.path.join(ROOT, "tests", "golden", n + ".npz")))  # noqa: E731
ic48, ih48 = g("ic48"), g("ih48")
pair48 = [(ic48["h"], ic48["xyz"]), (ih48["h"], ih48["xyz"])]
which = sys.argv[1:] or ["one48", "one48plain", "one48npt", "one4096"]
res = {}
if "one48" in which:
    res.update(case("pair48 x 1 walker, WL update + switch per move", pair48, 2, 1, 4800, wl=True))
if "one48plain" in which:
    res.update(case("pair48 x 1 walker, plain", pair48, 2, 1, 4800))
if "one48npt" in which:
    res.update(case("pair48 x 1 walker, NPT, WL update + switch per move", pair48, 2, 1, 4800, wl=True, npt=True))
if "eight48npt" in which:
    res.update(case("pair48 x 8 walkers, NPT, WL update + switch per move", pair48, 2, 8, 4800, wl=True, npt=True))
if "one4096" in which:
    h, x = lat.ice_box("ih", (8, 8, 8), 0.0)
    res.update(case("ih4096 x 1 walker", [(h, x)], 1, 1, 2000, sigma=0.1))
if "one1536" in which:
    ic, ih = g("ic1536"), g("ih1536")
    res.update(case("pair1536 x 1 walker, WL update + switch per move", [(ic["h"], ic["xyz"]), (ih["h"], ih["xyz"])], 2, 1, 2000, wl=True, sigma=0.1, mu_range=8000.0))
print(json.dumps(res, indent=1))
