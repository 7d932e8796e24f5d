"""GPU: the whole weight-generation / sampling loop of mc_cycle on the device-resident farm -- moves, Wang-Landau updates,
switch attempts on the GPU; exchange step, flatness checks (WangLandauSchedule) and the delta G read-out on the host --
against the oracle replay of the SAME scenarios that tests/test_schedule_pin.py pins to the reference program (48-molecule
Ic/Ih pair, seed 424242, flat_chk_int = 8).  One walker = one rank of the reference, so every table must agree."""
import os

import numpy as np
import pytest

from conftest import load_golden

import test_schedule_pin as tsp
import test_sweep_pin as pin

pytestmark = pytest.mark.gpu

F0 = float(np.float32(0.05))


@pytest.fixture(scope="module")
def so():
    from oracle import SweepOracle
    return SweepOracle()


def boxes48():
    z1, z2 = load_golden("ic48_t015"), load_golden("ih48_t020")
    from mc_water_ls_mw_amd.lattice import ANG_TO_BOHR   # noqa: F401  (fixtures are already in bohr)
    return [(z1["h"], z1["xyz"]), (z2["h"], z2["xyz"])]


def run_farm(cycles, **kw):
    from mc_water_ls_mw_amd.farm import run
    b = boxes48()
    return run([b[0][0], b[1][0]], [b[0][1], b[1][1]], walkers=1, cycles=cycles, temperature=200.0, seed=pin.SEED,
               thermalise=False, list_update_int=10, mpi_sync_int=10 ** 9, **kw)


@pytest.mark.parametrize("name,kw,okw,expect", [
    ("reset then halving", dict(wl_schedule=1, wl_minhist=-1), dict(schedule=1, minhist=-1),
     ["first reset", "halved", "halved", "halved", "halved"]),
    ("switch to 1/t", dict(wl_schedule=0, wl_flattol=1.0e9, wl_useinvt=True), dict(schedule=0, flattol=1.0e9, useinvt=True),
     ["halved", "invt", "invt", "invt", "invt"]),
    ("not flat", dict(wl_schedule=2), dict(schedule=2), ["checked"] * 5),
])
def test_weight_generation_with_flatness_checks_follows_the_pinned_oracle(tmp_path, so, c_oracle, name, kw, okw, expect):
    from mc_water_ls_mw_amd import io as mwio
    from mc_water_ls_mw_amd.sweep import MuGrid
    from oracle import schedule as osch
    grid = MuGrid(101, -400.0, 400.0)
    res = run_farm(40, wl_factor=F0, flat_chk_int=8, outdir=str(tmp_path), **kw)
    st = osch.new_state(F0, **okw)
    xs, w, hi, uh, events = tsp.replay_with_schedule(so, c_oracle, boxes48(), 40, grid, st, 8)
    assert [e["action"] for e in res["flatness_events"]] == [e[1] for e in events] == expect
    assert res["wl_factor"] == st["wl_factor"] and res["wl_invt_active"] == st["invt_active"]
    wt_d, hi_d, _ = res["walker1_tables"]
    assert np.allclose(hi_d, hi, rtol=0, atol=1e-12)
    assert np.allclose(wt_d, w, rtol=1e-10, atol=1e-11)
    assert np.abs(np.array(res["walker1_positions"]) - xs).max() < 1e-9
    if st["wlf"]:
        wlf = mwio.read_wlf(str(tmp_path))
        assert [c for c, _ in wlf] == [c for c, _ in st["wlf"]]
        assert np.allclose([f for _, f in wlf], [f for _, f in st["wlf"]], rtol=1e-11)
        assert any(f.startswith("eta_weights.dat_") for f in os.listdir(tmp_path))


def test_sampling_run_delta_g_follows_the_pinned_oracle(tmp_path, so, c_oracle):
    from mc_water_ls_mw_amd.sweep import MuGrid
    from oracle import schedule as osch
    grid = MuGrid(101, -400.0, 400.0)
    weight = 0.5 * np.abs(grid.mu_bin)
    res = run_farm(60, samplerun=True, weight=weight, deltaG_int=60, max_mc_cycles=60, eq_mc_cycles=1, outdir=str(tmp_path))
    lun = osch.unbiased_norm(weight, grid.av_binwidth, 60, 1, 1, 48)
    xs, w, hi, uh, _ = tsp.replay_with_schedule(so, c_oracle, boxes48(), 60, grid, None, 10 ** 9, samplerun=True,
                                                weight=weight, lun=lun)
    _, hi_d, uh_d = res["walker1_tables"]
    assert np.allclose(hi_d, hi, rtol=0, atol=1e-12) and uh[:50].sum() > 0 and uh[51:].sum() > 0
    assert np.allclose(uh_d, uh, rtol=1e-9, atol=0)
    assert res["delta_g"]["cycle"] == 60
    assert res["delta_g"]["kT"] == pytest.approx(osch.delta_g(uh, grid.binwidth), rel=1e-9)
    assert os.path.exists(os.path.join(tmp_path, "unbiased_histogram_0000000060.dat"))
