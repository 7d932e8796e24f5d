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


def test_starting_weights_survive_the_first_synchronisation_of_several_walkers():
    """mc_init's comms_allreduce_eta makes the table read from eta_weights.dat the baseline of the delta scheme
    (mc_moves.F90:738-776): with fixed weights (a sample run) every later synchronisation returns that table, however many
    walkers there are -- not walkers x the table."""
    from mc_water_ls_mw_amd.sweep import MuGrid
    grid = MuGrid(101, -400.0, 400.0)
    weight = 0.05 * np.abs(grid.mu_bin)
    b = boxes48()
    from mc_water_ls_mw_amd.farm import run
    res = run([b[0][0], b[1][0]], [b[0][1], b[1][1]], walkers=4, cycles=6, temperature=200.0, seed=pin.SEED, thermalise=True,
              list_update_int=10, mpi_sync_int=3, wl_factor=F0, samplerun=True, weight=weight, deltaG_int=10 ** 6, max_mc_cycles=6, eq_mc_cycles=1)
    synced_w, synced_h, synced_u = res["tables"]
    assert np.array_equal(synced_w, weight)
    for w in res["first_walkers"]:
        assert np.array_equal(w["tables"][0], weight)
    assert synced_h.sum() > 0 and synced_u.sum() > 0


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


# ---- the reference's own 'dd' examples in miniature (examples/ice1_gen_weights_dd, examples/ice1_sample_dd): the shipped
# inputs (ideal Ic / Ih of 48 molecules = tests/golden/ic48, ih48), NPT at 200 K and 1 atm, volume moves at 1/N against
# translations at 0.5, a switch attempt after every move, two windows overlapping by two bins.  max_mc_cycles /
# eq_mc_cycles / flat_chk_int are cut from 5 10^6 / 10^4 / 10^4 to 24 (48) / 4 / 8 so that the oracle replays them in seconds.

def ideal48():
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    return [(z1["h"], z1["xyz"]), (z2["h"], z2["xyz"])]


def replay_npt_window(so, c_oracle, k, size, cycles, eq, wl_factor0, samplerun=False, weight=None, lun=0.0, flat_int=8,
                      flattol=0.05):
    """Rank k of `size` of an NPT 'dd' run, cycle by cycle on the oracle (mwo_sweep_full: the restatement
    tests/test_sweep_pin.py pins to the reference program under NPT, with the 'dd' rules of tests/test_options_pin.py)."""
    from mc_water_ls_mw_amd.lattice import ANG_TO_BOHR
    from mc_water_ls_mw_amd.sweep import KB, MuGrid
    from oracle import FullSweepState
    from oracle import schedule as osch
    grid = MuGrid(101, -400.0, 400.0)
    w0 = grid.window(k, size, 2)
    g = grid.restricted(w0)
    b = ideal48()
    beta, p, transP = 1.0 / (KB * 200.0), 1.0 / pin.AUP_TO_ATM, 0.5 / (0.5 + 1.0 / 48)
    st = FullSweepState(c_oracle, [b[0][0], b[1][0]], [b[0][1], b[1][1]])
    mu = st.model_energy[0] + p * st.volume[0] - st.model_energy[1] - p * st.volume[1]
    st.ls_mu = mu * beta - 48.0 * np.log(st.volume[0] / st.volume[1])
    st.ls = w0["ls"] or 1
    w = np.zeros(101) if weight is None else np.array(weight, dtype=float)
    w[:w0["start_bin"] - 1] = 0.0                                          # mc_moves.F90:808-812
    w[w0["end_bin"]:] = 0.0
    hi, uh = np.zeros(101), np.zeros(101)
    sched = osch.new_state(wl_factor0, schedule=0, flattol=flattol)
    events = []
    so.set_dd(True, eq, False)
    try:
        for cyc in range(1, cycles + 1):
            if cyc % 10 == 0:
                st.rebuild_lists(c_oracle)
            so.full(st, 48, pin.SEED, k, (cyc - 1) * 48, transP, 0.924 * ANG_TO_BOHR, beta, 1.1 * ANG_TO_BOHR, g, w, hi, uh,
                    record=cyc >= eq, samplerun=samplerun, always_switch=True, npt=True,
                    wl_factor=0.0 if samplerun else sched["wl_factor"], log_unbiased_norm=lun, pressure=p)
            if not samplerun and cyc % flat_int == 0:
                what = osch.flatness_step(sched, cyc, 48, hi, w, start_bin=w0["start_bin"], end_bin=w0["end_bin"], dd=True)
                if what not in ("none", "checked"):
                    events.append((cyc, k, what))
        in_window, failed = so.get_dd()
    finally:
        so.set_dd(False)
    assert in_window and not failed
    return dict(st=st, weight=w, hist=hi, uhist=uh, factor=sched["wl_factor"], events=events, window=w0)


def run_example(cycles, **kw):
    from mc_water_ls_mw_amd.farm import run
    b = ideal48()
    return run([b[0][0], b[1][0]], [b[0][1], b[1][1]], walkers=2, cycles=cycles, temperature=200.0, seed=pin.SEED,
               thermalise=False, list_update_int=10, mpi_sync_int=250, npt=True, pressure_atm=1.0, parallel_strategy="dd",
               window_overlap=2, eq_mc_cycles=4, flat_chk_int=8, **kw)


def test_ice1_gen_weights_dd_example_in_miniature(so, c_oracle):
    from mc_water_ls_mw_amd.comms import WalkerComms
    res = run_example(24, wl_factor=0.005, wl_flattol=0.05)
    ranks = [replay_npt_window(so, c_oracle, k, 2, 24, 4, 0.005) for k in range(2)]
    assert res["in_window"] == [True, True] and res["volume_moves_walker1"][0] == ranks[0]["st"].nvol[0] > 0
    assert sorted((e["cycle"], e["walker"], e["action"]) for e in res["flatness_events"]) == sorted(ranks[0]["events"] + ranks[1]["events"])
    wt_d, hi_d, _ = res["walker1_tables"]
    assert hi_d.sum() > 0 and np.allclose(hi_d, ranks[0]["hist"], rtol=1e-12, atol=1e-12)
    assert np.allclose(wt_d, ranks[0]["weight"], rtol=1e-9, atol=1e-11)
    assert np.abs(np.array(res["walker1_positions"]) - ranks[0]["st"].xyz).max() < 1e-8
    joined = WalkerComms(101).join_eta(np.array([ranks[0]["weight"], ranks[1]["weight"]]), 2)
    assert np.abs(joined).max() > 0 and np.allclose(res["joined"]["weight"], joined, rtol=1e-9, atol=1e-10)
    assert res["wl_factor"] == max(r["factor"] for r in ranks)


def test_ice1_sample_dd_example_in_miniature(so, c_oracle):
    """Fixed weights from the example's own eta_weights.dat, each walker keeping its window's part; the unbiased
    histograms of the two windows stitched (comms_join_uhist) and turned into a free-energy difference."""
    import os
    from conftest import GOLDEN
    from mc_water_ls_mw_amd import io as mwio
    from mc_water_ls_mw_amd.comms import WalkerComms
    from mc_water_ls_mw_amd.sweep import MuGrid
    from oracle import schedule as osch
    grid = MuGrid(101, -400.0, 400.0)
    factor, mu_bin, weight = mwio.read_table(os.path.join(GOLDEN, "ice1_sample_dd_eta_weights.dat"))
    assert np.allclose(mu_bin, grid.mu_bin, rtol=1e-12, atol=1e-12) and factor == pytest.approx(0.05, rel=1e-6)
    res = run_example(48, samplerun=True, weight=weight, deltaG_int=48, max_mc_cycles=48)     # (48: every seam bin visited)
    lun = osch.unbiased_norm(weight, grid.av_binwidth, 48, 4, 2, 48)
    ranks = [replay_npt_window(so, c_oracle, k, 2, 48, 4, 0.0, samplerun=True, weight=weight, lun=lun) for k in range(2)]
    assert all(r["uhist"][47:52].min() > 0 for r in ranks)
    _, hi_d, uh_d = res["walker1_tables"]
    assert np.allclose(hi_d, ranks[0]["hist"], rtol=1e-12, atol=1e-12) and np.allclose(uh_d, ranks[0]["uhist"], rtol=1e-8, atol=0)
    ju = WalkerComms(101).join_uhist(np.array([r["uhist"] for r in ranks]), 2)
    assert np.allclose(res["joined"]["unbiased_hist"], ju, rtol=1e-8, atol=0)
    assert ju[:50].sum() > 0 and ju[50:].sum() > 0
    assert res["delta_g"]["kT"] == pytest.approx(osch.delta_g(ju, grid.binwidth), rel=1e-7)


def test_equilibration_step_size_tuning_on_the_farm_follows_the_pinned_oracle(so, c_oracle):
    """eq_adjust_mc (every shipped example sets it): two independent NPT walkers, a report every 4 cycles, 13 equilibration
    cycles -- each walker tunes its own mc_max_trans / mc_dv_max from its own acceptance ratios (each rank of the reference
    does) and gets its stored energies refreshed, as tests/test_options_pin.py pins against the reference program."""
    from mc_water_ls_mw_amd.farm import run
    b = boxes48()
    res = run([b[0][0], b[1][0]], [b[0][1], b[1][1]], walkers=2, cycles=20, temperature=200.0, seed=pin.SEED,
              thermalise=False, list_update_int=10, mpi_sync_int=10 ** 9, wl_factor=F0, npt=True, pressure_atm=1.0,
              eq_mc_cycles=13, eq_adjust_mc=True, monitor_int=4, flat_chk_int=10 ** 9)
    transP = 0.5 / (0.5 + 1.0 / 48)
    for k in range(2):
        st, w, hi, max_trans, dv_max, history = top.replay_with_step_adjustment(so, c_oracle, boxes48(), 20, 13, 4, transP, walker=k)
        fw = res["first_walkers"][k]
        assert res["max_trans_bohr"][k] == pytest.approx(max_trans, rel=1e-12) and res["dv_max_bohr"][k] == pytest.approx(dv_max, rel=1e-12)
        assert np.abs(np.array(fw["positions"]) - st.xyz).max() < 1e-8 and fw["ls"] == st.ls
        assert np.allclose(fw["tables"][1], hi, rtol=1e-12, atol=1e-12) and np.allclose(fw["tables"][0], w, rtol=1e-9, atol=1e-11)
    assert res["max_trans_bohr"][0] != res["max_trans_bohr"][1]              # every walker its own


def test_chain_synchronisation_inside_the_farm_loop_follows_the_pinned_oracle(so, c_oracle):
    """latt_sync_int = 5: every fifth cycle of an NPT run lattice 2 is re-imposed from lattice 1 (the loop
    tests/test_sweep_pin.py pins against the reference program, here through farm.run)."""
    from mc_water_ls_mw_amd.farm import run
    from mc_water_ls_mw_amd.lattice import ANG_TO_BOHR
    from mc_water_ls_mw_amd.sweep import KB, MuGrid
    from oracle import FullSweepState
    b = boxes48()
    res = run([b[0][0], b[1][0]], [b[0][1], b[1][1]], walkers=1, cycles=30, temperature=200.0, seed=pin.SEED,
              thermalise=False, list_update_int=10, mpi_sync_int=10 ** 9, wl_factor=F0, npt=True, pressure_atm=1.0,
              eq_mc_cycles=1, latt_sync_int=5, flat_chk_int=10 ** 9)
    grid = MuGrid(101, -400.0, 400.0)
    beta, p, transP = 1.0 / (KB * 200.0), 1.0 / pin.AUP_TO_ATM, 0.5 / (0.5 + 1.0 / 48)
    st = FullSweepState(c_oracle, [b[0][0], b[1][0]], [b[0][1], b[1][1]])
    mu = st.model_energy[0] + p * st.volume[0] - st.model_energy[1] - p * st.volume[1]
    st.ls_mu = mu * beta - 48.0 * np.log(st.volume[0] / st.volume[1])
    w, hi, uh = np.zeros(101), np.zeros(101), np.zeros(101)
    for cyc in range(1, 31):
        if cyc % 10 == 0:
            st.rebuild_lists(c_oracle)
        so.full(st, 48, pin.SEED, 0, (cyc - 1) * 48, transP, 0.924 * ANG_TO_BOHR, beta, 1.1 * ANG_TO_BOHR, grid, w, hi, uh,
                record=True, samplerun=False, always_switch=True, npt=True, wl_factor=F0, pressure=p)
        if cyc % 5 == 0:
            so.chain_sync(st, beta, p)
    fw = res["first_walkers"][0]
    assert st.nvol[1] > 0 and np.abs(np.array(fw["positions"]) - st.xyz).max() < 1e-8 and fw["ls"] == st.ls
    assert abs(fw["ls_mu"] - st.ls_mu) < 1e-6 * (1 + abs(st.ls_mu))
    assert np.allclose(fw["tables"][1], hi, rtol=1e-12, atol=1e-12) and np.allclose(fw["tables"][0], w, rtol=1e-9, atol=1e-11)


def test_farm_restarts_from_its_own_checkpoints_and_the_files_are_the_references(tmp_path):
    """chkpt_dump_int / restart: two NPT walkers write checkpointRRR.dat.{1,2} in the reference's format (mc_checkpoint_write,
    mc_moves.F90:324-390); a second run restarted from them ends where the uninterrupted run ends (lists are rebuilt at the
    same point of both runs: list_update_int = 11, restart after cycle 10)."""
    from mc_water_ls_mw_amd import io as mwio
    from mc_water_ls_mw_amd.farm import run
    b = boxes48()
    kw = dict(walkers=2, temperature=200.0, seed=pin.SEED, thermalise=False, list_update_int=11, mpi_sync_int=10 ** 9,
              wl_factor=F0, npt=True, pressure_atm=1.0, eq_mc_cycles=1, flat_chk_int=10 ** 9, outdir=str(tmp_path))
    full = run([b[0][0], b[1][0]], [b[0][1], b[1][1]], cycles=20, **kw)
    run([b[0][0], b[1][0]], [b[0][1], b[1][1]], cycles=10, chkpt_dump_int=5, **kw)
    path, c = mwio.latest_checkpoint(str(tmp_path), 1)                      # walker 2 = "rank" 1
    assert path.endswith("checkpoint001.dat.2") and c["cycle"] == 10 and c["nwater"] == 48 and c["hmatrix"].shape == (2, 3, 3)
    assert c["histogram"].sum() > 0 and not c["samplerun"]
    rest = run([b[0][0], b[1][0]], [b[0][1], b[1][1]], cycles=10, restart=True, **kw)
    for k in range(2):
        a, r = full["first_walkers"][k], rest["first_walkers"][k]
        assert np.abs(np.array(a["positions"]) - np.array(r["positions"])).max() < 1e-8 and a["ls"] == r["ls"]
        assert np.allclose(a["tables"][1], r["tables"][1], rtol=1e-12, atol=1e-12)
        assert np.allclose(a["tables"][0], r["tables"][0], rtol=1e-9, atol=1e-11)


def test_restart_continues_through_table_synchronisations(tmp_path):
    """Three walkers, tables synchronised every 4 cycles, checkpoint at cycle 8 (a synchronisation cycle): the restarted run's
    next synchronisation must see the checkpointed table as its baseline -- with the baseline left at zero it returned 3 x the
    weights -- and the run ends where the uninterrupted one ends."""
    from mc_water_ls_mw_amd.farm import run
    b = boxes48()
    kw = dict(walkers=3, temperature=200.0, seed=pin.SEED, thermalise=True, list_update_int=4, mpi_sync_int=4,
              wl_factor=F0, flat_chk_int=10 ** 9, outdir=str(tmp_path))
    full = run([b[0][0], b[1][0]], [b[0][1], b[1][1]], cycles=16, **kw)
    run([b[0][0], b[1][0]], [b[0][1], b[1][1]], cycles=8, chkpt_dump_int=8, **kw)
    rest = run([b[0][0], b[1][0]], [b[0][1], b[1][1]], cycles=8, restart=True, **kw)
    assert full["tables"][0].max() > 0
    assert np.allclose(rest["tables"][0], full["tables"][0], rtol=1e-9, atol=1e-11)      # weights
    assert np.allclose(rest["tables"][1], full["tables"][1], rtol=1e-12, atol=1e-12)     # histogram
    for k in range(3):
        a, r = full["first_walkers"][k], rest["first_walkers"][k]
        assert np.abs(np.array(a["positions"]) - np.array(r["positions"])).max() < 1e-8 and a["ls"] == r["ls"]


@pytest.mark.parametrize("mode", ["dd", "swetnam"])
def test_farm_restart_restores_the_per_walker_wang_landau_state(tmp_path, mode):
    """mc_checkpoint_load reads wl_factor / wl_invt_active on EVERY rank (mc_moves.F90:447-464).  With 'dd' every walker is
    such a rank -- its own increment, halved by its own flatness checks, its own first-cycle state: a restarted run picks all
    of it up from the walkers' files and ends where the uninterrupted run ends.  With wl_swetnam the increment and the visit
    total live on the device; the loader's own rule for the total is sumhist = sum(histogram) past the 'mw' branch's early
    return and the initial 0 inside it (:469-475), so the device must hold exactly that after a restart."""
    from mc_water_ls_mw_amd import io as mwio
    from mc_water_ls_mw_amd.farm import run
    b = boxes48()
    kw = dict(walkers=2, temperature=200.0, seed=pin.SEED, thermalise=False, list_update_int=11, mpi_sync_int=10 ** 9,
              wl_factor=F0, outdir=str(tmp_path))
    if mode == "dd":      # (leshift puts the pair inside both windows' overlap; flatness checks every 4 cycles halve the increments)
        kw.update(leshift=True, parallel_strategy="dd", window_overlap=2, eq_mc_cycles=2, flat_chk_int=4, wl_schedule=1, wl_minhist=-1)
    else:
        kw.update(wl_swetnam=True, wl_alpha=0.01, flat_chk_int=10 ** 9)
    full = run([b[0][0], b[1][0]], [b[0][1], b[1][1]], cycles=24, **kw)
    first = run([b[0][0], b[1][0]], [b[0][1], b[1][1]], cycles=12, chkpt_dump_int=12, **kw)
    chk = [mwio.latest_checkpoint(str(tmp_path), k)[1] for k in range(2)]
    assert all(c["cycle"] == 12 for c in chk)
    if mode == "dd":      # (ref_enthalpy of a leshift run is the starting configuration's; a restart is handed the same values)
        assert all(c["wl_factor"] == first["wl_factor"] < F0 for c in chk)      # halved twice before the restart, the second time AT cycle 12
        kw.update(input_ref_enthalpy=np.array(full["ref_enthalpy"]))
    else:
        assert all(np.isfinite(c["wl_factor"]) and c["wl_factor"] != F0 for c in chk) and chk[0]["wl_factor"] != chk[1]["wl_factor"]
    rest = run([b[0][0], b[1][0]], [b[0][1], b[1][1]], cycles=12, restart=True, **kw)
    fac, sumh, _ = rest["restart_factors"]
    assert fac == [c["wl_factor"] for c in chk]                                 # every walker its own
    assert sumh == ([float(np.sum(c["histogram"])) for c in chk] if mode == "dd" else [0.0, 0.0])
    if mode == "dd":
        assert rest["wl_factor"] == pytest.approx(full["wl_factor"], rel=1e-12)
        assert [(e["cycle"], e["walker"], e["action"]) for e in first["flatness_events"] + rest["flatness_events"]] == \
               [(e["cycle"], e["walker"], e["action"]) for e in full["flatness_events"]]
        for k in range(2):
            a, r = full["first_walkers"][k], rest["first_walkers"][k]
            assert np.abs(np.array(a["positions"]) - np.array(r["positions"])).max() < 1e-8 and a["ls"] == r["ls"]
            assert np.allclose(a["tables"][1], r["tables"][1], rtol=1e-12, atol=1e-12)
            assert np.allclose(a["tables"][0], r["tables"][0], rtol=1e-9, atol=1e-11)
    else:
        assert np.isfinite(rest["wl_factor"]) and rest["histogram_total"] is None
