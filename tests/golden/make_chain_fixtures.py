#!/usr/bin/env python3
"""Golden chains of the Monte Carlo driver for bench.py's self-check (and tests/test_gpu_bench_chain.py): walker 0 of the
workloads the bench's `production_driver` section runs, advanced by the ORACLE (oracle/mw_oracle.c: mwo_sweep_cycle,
mwo_sweep_full, mwo_sweep_translation -- the restatement of mc_moves.F90:966-1213,1216-1534,1536-1689 pinned against the
reference program itself, tests/test_sweep_pin.py).  CPU only; run in the build container:

    python tests/golden/make_chain_fixtures.py

Per chain: the recipe (which cells, thermalisation seed, temperature, step sizes, grid, seed of the random stream), the
starting full-box energies and order parameter, and the move log -- molecule, outcome flags (1 accepted, 2 lattice switch,
4 volume move), the four local energies (or the new full-box energies and volumes of a volume move), the order parameter
after the move, the acceptance exponent."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from mc_water_ls_mw_amd import lattice as lat          # noqa: E402
from mc_water_ls_mw_amd.sweep import KB, MuGrid         # noqa: E402
from oracle import COracle, FullSweepState, SweepOracle  # noqa: E402

#: the replica farm's own settings (mc_water_ls_mw_amd/farm.py run(), bench.py production_driver)
FARM = dict(temperature=200.0, max_trans_ang=1.1, dv_max_ang=0.924, mu_range=400.0, nbins=101, wl_factor=0.05, sigma_ang=0.05,
            seed=2025, pressure_atm=1.0)


def pair48_positions(walker_global=0):
    z1, z2 = np.load(os.path.join(GOLD, "ic48.npz")), np.load(os.path.join(GOLD, "ih48.npz"))
    hs = [z1["h"], z2["h"]]
    xs = [lat.thermalise(z["xyz"], FARM["sigma_ang"], 7919 * walker_global + l) for l, z in enumerate((z1, z2))]   # farm.py: 7919 (rank W + w) + l
    return hs, xs


def initial_mu(e, v, n, beta, pressure):
    import math
    mu = (e[0] + pressure * v[0]) - (e[1] + pressure * v[1])
    return mu * beta - n * math.log(v[0] / v[1])


def farm48(npt, nmoves):
    C, so = COracle(), SweepOracle()
    hs, xs = pair48_positions(0)
    grid = MuGrid(FARM["nbins"], -FARM["mu_range"], FARM["mu_range"])
    beta = 1.0 / (KB * FARM["temperature"])
    max_trans = FARM["max_trans_ang"] * lat.ANG_TO_BOHR
    p_au = FARM["pressure_atm"] / 2.90363081e8
    n = len(xs[0])
    st = FullSweepState(C, hs, xs)
    e0 = st.model_energy.copy()
    mu0 = initial_mu(e0, st.volume, n, beta, p_au)
    st.ls_mu = mu0
    wt, hi, uh = np.zeros(grid.nbins), np.zeros(grid.nbins), np.zeros(grid.nbins)
    transP = 0.5 / (0.5 + 1.0 / n) if npt else 1.0        # io.f90:171-172: volume moves 1/N against translations 0.5
    dv_max = FARM["dv_max_ang"] * lat.ANG_TO_BOHR if npt else 0.0
    log = so.full(st, nmoves, FARM["seed"], 0, 0, transP, dv_max, beta, max_trans, grid, wt, hi, uh, record=True, samplerun=False,
                  always_switch=True, npt=npt, wl_factor=FARM["wl_factor"], pressure=p_au)
    return dict(log=log, e0=e0, mu0=mu0, weight=wt, histogram=hi, final_model_energy=st.model_energy.copy(), final_ls=st.ls,
                final_mu=st.ls_mu, accepted=st.accepted, switches=st.switches, nmoves=nmoves, npt=int(npt),
                **{k: v for k, v in FARM.items()})


def ih4096(nmoves):
    C, so = COracle(), SweepOracle()
    h, x0 = lat.ice_box("ih", (8, 8, 8), 0.0)
    x = lat.thermalise(x0, 0.1, 0)                         # tools/sweep_measurements.py ih4096: sigma 0.1, seed 1000 l + w
    beta = 1.0 / (KB * 200.0)
    r = so.sweep(nmoves, 1, 0, 0, [h], [x], beta, 1.1 * lat.ANG_TO_BOHR)
    return dict(log=r["log"], nmoves=nmoves, accepted=r["accepted"], final_model_energy=r["model_energy"], temperature=200.0,
                max_trans_ang=1.1, sigma_ang=0.1, seed=1, thermalise_seed=0)


if __name__ == "__main__":
    n = 48
    # a farm rebuilds its lists before cycle 10 (list_update_int = 10): the first nine cycles run on the starting lists
    for name, d in (("chain_farm48_nvt", farm48(False, 9 * n)), ("chain_farm48_npt", farm48(True, 9 * n)), ("chain_ih4096", ih4096(256))):
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **d)
        lg = d["log"]
        print(name, "moves", len(lg), "accepted", int((lg[:, 1].astype(int) & 1).sum()), "switches", int(((lg[:, 1].astype(int) >> 1) & 1).sum()),
              "volume moves", int(((lg[:, 1].astype(int) >> 2) & 1).sum()))
