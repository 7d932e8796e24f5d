"""GPU: the remaining run options of the weights / histogram layer on the device-resident farm --
leshift, wl_swetnam, parallel_strategy = 'dd' -- through farm.run, against the oracle replay of the SAME scenarios that
tests/test_options_pin.py pins to the reference program (48-molecule Ic/Ih pair, seed 424242)."""
import numpy as np
import pytest

from conftest import load_golden

import test_options_pin as top
import test_sweep_pin as pin

pytestmark = pytest.mark.gpu

F0 = float(np.float32(0.05))


@pytest.fixture(scope="module")
def so():
    from oracle import SweepOracle
    s = SweepOracle()
    yield s
    s.set_leshift(0.0, 0.0); s.set_swetnam(False); s.set_dd(False)


def boxes48():
    z1, z2 = load_golden("ic48_t015"), load_golden("ih48_t020")
    return [(z1["h"], z1["xyz"]), (z2["h"], z2["xyz"])]


def run_farm(cycles, walkers=1, **kw):
    from mc_water_ls_mw_amd.farm import run
    b = boxes48()
    return run([b[0][0], b[1][0]], [b[0][1], b[1][1]], walkers=walkers, cycles=cycles, temperature=200.0, seed=pin.SEED,
               thermalise=False, list_update_int=10, mpi_sync_int=10 ** 9, wl_factor=F0, **kw)


def start_energies(c_oracle):
    from mc_water_ls_mw_amd.sweep import KB
    return top.start_state(c_oracle, boxes48(), 1.0 / (KB * 200.0))[4]


def test_leshift_run_follows_the_pinned_oracle(so, c_oracle):
    from mc_water_ls_mw_amd.sweep import MuGrid
    grid = MuGrid(101, -400.0, 400.0)
    res = run_farm(40, leshift=True)
    me0 = start_energies(c_oracle)
    assert np.allclose(res["ref_enthalpy"], me0, rtol=1e-12)               # NVT: model energies, no PV term
    so.set_leshift(me0[0], me0[1])
    try:
        xs, ls, mu, w, hi = top.replay(so, c_oracle, boxes48(), 40, grid, F0, dref=me0[0] - me0[1])
    finally:
        so.set_leshift(0.0, 0.0)
    wt_d, hi_d, _ = res["walker1_tables"]
    assert hi.sum() > 0 and np.allclose(hi_d, hi, rtol=0, atol=1e-12) and np.allclose(wt_d, w, rtol=1e-10, atol=1e-11)
    assert np.abs(np.array(res["walker1_positions"]) - xs).max() < 1e-9


def test_leshift_enters_delta_g(c_oracle):
    """beta (H_ref(2) - H_ref(1)) is added back to the free-energy difference of a sample run (mc_moves.F90:2586)."""
    from mc_water_ls_mw_amd.sweep import KB, MuGrid
    grid = MuGrid(101, -400.0, 400.0)
    weight = 0.05 * np.abs(grid.mu_bin)
    kw = dict(samplerun=True, weight=weight, deltaG_int=30, max_mc_cycles=30, eq_mc_cycles=1)
    shifted = run_farm(30, leshift=True, **kw)
    me0 = start_energies(c_oracle)
    _, _, uh = shifted["walker1_tables"]
    from oracle import schedule as osch
    assert uh[:50].sum() > 0 and uh[50:].sum() > 0                          # mu starts at ~ -0.1, in the middle bin: both sides are seen
    expect = osch.delta_g(uh, grid.binwidth) + (me0[1] - me0[0]) / (KB * 200.0)
    assert shifted["delta_g"]["kT"] == pytest.approx(expect, rel=1e-9)


def test_swetnam_run_follows_the_pinned_oracle(so, c_oracle):
    from mc_water_ls_mw_amd.sweep import MuGrid
    grid = MuGrid(101, -400.0, 400.0)
    res = run_farm(30, wl_swetnam=True, wl_alpha=0.01, flat_chk_int=10)
    so.set_swetnam(True, 0.01, F0, -400.0, 400.0, 0.0)
    try:
        xs, ls, mu, w, hi = top.replay(so, c_oracle, boxes48(), 30, grid, F0)
        sumhist, factor = so.get_swetnam()
    finally:
        so.set_swetnam(False)
    wt_d, hi_d, _ = res["walker1_tables"]
    assert res["wl_factor"] == pytest.approx(factor, rel=1e-10) and factor < 0.0
    assert [e["action"] for e in res["flatness_events"]] == ["swetnam"] * 3  # the flatness test itself is off (mc_moves.F90:2018)
    assert np.allclose(hi_d, hi, rtol=0, atol=1e-12) and np.allclose(wt_d, w, rtol=1e-9, atol=1e-10)
    assert np.abs(np.array(res["walker1_positions"]) - xs).max() < 1e-9


def test_dd_farm_of_two_windows_follows_the_pinned_oracle(so, c_oracle):
    """Two walkers = ranks 0 and 1 of a two-window run.  With leshift the pair starts at mu ~ -0.1, inside the overlap of
    both windows, so both pass the equilibration check: three equilibration cycles without updates or switch attempts,
    then weight generation confined to each walker's window, a flatness check per window every 8 cycles (reset, then
    halvings -- no weight shift in 'dd'), and the windows stitched at the end (comms_join_eta)."""
    from mc_water_ls_mw_amd.comms import WalkerComms
    from mc_water_ls_mw_amd.sweep import MuGrid
    from oracle import schedule as osch
    grid = MuGrid(101, -400.0, 400.0)
    res = run_farm(40, walkers=2, leshift=True, parallel_strategy="dd", window_overlap=2, eq_mc_cycles=3, flat_chk_int=8,
                   wl_schedule=1, wl_minhist=-1)
    me0 = start_energies(c_oracle)
    assert res["in_window"] == [True, True]
    tables, events_all = [], []
    for k in range(2):
        w0 = grid.window(k, 2, 2)
        assert w0 == res["windows"][k]
        st = osch.new_state(F0, schedule=1, minhist=-1)
        events = []

        def per_cycle(cyc, hi, w):
            if cyc % 8 == 0:
                what = osch.flatness_step(st, cyc, 48, hi, w, start_bin=w0["start_bin"], end_bin=w0["end_bin"], dd=True)
                if what not in ("none", "checked"):
                    events.append((cyc, k, what))

        so.set_leshift(me0[0], me0[1])
        so.set_dd(True, 3, False)
        try:
            xs, ls, mu, w, hi = top.replay(so, c_oracle, boxes48(), 40, grid.restricted(w0), lambda cyc: st["wl_factor"],
                                           dref=me0[0] - me0[1], eq=3, per_cycle=per_cycle, walker=k, ls=w0["ls"] or 1)
            in_window, failed = so.get_dd()
        finally:
            so.set_dd(False); so.set_leshift(0.0, 0.0)
        assert in_window and not failed
        tables.append((w, hi, xs, st["wl_factor"]))
        events_all += events
    got = sorted((e["cycle"], e["walker"], e["action"]) for e in res["flatness_events"])
    assert got == sorted(events_all) and len(got) >= 2
    wt_d, hi_d, _ = res["walker1_tables"]
    assert np.allclose(hi_d, tables[0][1], rtol=0, atol=1e-12) and np.allclose(wt_d, tables[0][0], rtol=1e-10, atol=1e-11)
    assert np.abs(np.array(res["walker1_positions"]) - tables[0][2]).max() < 1e-9
    e0 = grid.window(0, 2, 2)["end_bin"]
    assert wt_d[e0:].max() == 0.0 and wt_d[:e0].max() > 0                   # nothing outside the window
    # the stitched table: the two oracle tables through the join (restated independently in tests/test_comms.py)
    joined = WalkerComms(101).join_eta(np.array([tables[0][0], tables[1][0]]), 2)
    assert np.allclose(res["joined"]["weight"], joined, rtol=1e-10, atol=1e-11)
    assert res["wl_factor"] == max(tables[0][3], tables[1][3])


def test_dd_walker_outside_its_window_after_equilibration_stops_the_run():
    """Without leshift the pair starts at mu ~ -330: rank 1 of 2 (mu > -3) cannot get there in two cycles, and the
    reference stops with this message at cycle eq_mc_cycles (mc_moves.F90:187-201)."""
    with pytest.raises(RuntimeError, match="Not all walkers have reached their designated window"):
        run_farm(12, walkers=2, parallel_strategy="dd", window_overlap=2, eq_mc_cycles=2)
