"""GPU: the reference's -DMINU variant as a run option of the device-resident farm (mw_sweep_minu) against the oracle
replay of the scenarios tests/test_minu_pin.py pins to the reference program compiled -DMINU (48-molecule Ic/Ih pair,
seed 424242, leshift on so that the lattice of lower enthalpy actually changes)."""
import numpy as np
import pytest

from conftest import load_golden

import test_minu_pin as tmp_
import test_sweep_pin as pin

pytestmark = pytest.mark.gpu

F0 = float(np.float32(0.05))


@pytest.fixture(scope="module")
def so():
    from oracle import SweepOracle
    s = SweepOracle()
    yield s
    s.set_leshift(0.0, 0.0); s.set_minu(False)


def boxes48():
    z1, z2 = load_golden("ic48_t015"), load_golden("ih48_t020")
    return [(z1["h"], z1["xyz"]), (z2["h"], z2["xyz"])]


def run_farm(cycles, **kw):
    from mc_water_ls_mw_amd.farm import run
    b = boxes48()
    return run([b[0][0], b[1][0]], [b[0][1], b[1][1]], walkers=1, cycles=cycles, temperature=200.0, seed=pin.SEED,
               thermalise=False, list_update_int=10, mpi_sync_int=10 ** 9, wl_factor=F0, flat_chk_int=10 ** 9, leshift=True, **kw)


@pytest.mark.parametrize("npt", [False, True])
def test_minu_run_follows_the_pinned_oracle(so, c_oracle, npt):
    cycles = 16 if npt else 30
    transP = 0.5 / (0.5 + 1.0 / 48) if npt else 1.0
    res = run_farm(cycles, minu=True, npt=npt, pressure_atm=1.0)
    st, w, hi, visited = tmp_.replay(so, c_oracle, boxes48(), cycles, npt, transP, True)
    fw = res["first_walkers"][0]
    assert fw["ls"] == st.ls and np.abs(np.array(fw["positions"]) - st.xyz).max() < 1e-8
    assert hi.sum() > 0 and np.allclose(fw["tables"][1], hi, rtol=1e-12, atol=1e-12)
    assert np.allclose(fw["tables"][0], w, rtol=1e-9, atol=1e-11)
    if npt:
        assert res["volume_moves_walker1"][0] == st.nvol[0] > 0 and res["volume_moves_walker1"][1] == st.nvol[1]
    plain = run_farm(cycles, minu=False, npt=npt, pressure_atm=1.0)            # the option matters on this input
    assert np.abs(plain["first_walkers"][0]["tables"][1] - hi).max() > 0.5


def test_minu_needs_two_lattices():
    from mc_water_ls_mw_amd.energy import MwError, load_boxes
    from mc_water_ls_mw_amd.sweep import WalkerFarm
    z = load_golden("ic48_t015")
    em = load_boxes([z["h"]], [z["xyz"]])
    try:
        farm = WalkerFarm(em, 1, 200.0)
        with pytest.raises(MwError, match="two lattices"):
            farm.minu(True)
    finally:
        em.energy_deinit()
