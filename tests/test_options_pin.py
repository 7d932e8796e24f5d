"""CPU: pin the remaining run options of the weights / histogram layer against the REFERENCE
PROGRAM (oracle/_ref/mc_water_ref_rng: unmodified main / mc_moves / io with the RNG and stack-scrub interpositions,
`comms::size` / `comms::myrank` poked from the environment -- oracle/ref_wrap_rng.c):

  * leshift (userparams.f90:41; main.f90:146-150,173; mc_moves.F90:1567-1584): reference enthalpies in the order
    parameter and in the lattice-switch acceptance;
  * wl_swetnam (mc_moves.F90:1636-1653): the Wang-Landau increment recomputed from the histogram after every move;
  * parallel_strategy = 'dd' (mc_moves.F90:181-210,243-248,659-709,2002-2016): window assignment, choice of the active
    lattice, the equilibration rules, weights kept inside the window, per-window flatness check -- the serial program
    run as rank r of R.  (The window JOINS need other ranks' data and stay unpinned.)

The oracle (oracle/mw_oracle.c, oracle/schedule.py) replays each run and must end at the reference's checkpoint."""
import os
import re
import struct

import numpy as np
import pytest

import test_sweep_pin as pin

pytestmark = pytest.mark.skipif(not os.path.exists(pin.RNG), reason="oracle/_ref/mc_water_ref_rng not built")

F0 = float(np.float32(0.05))     # userparams.f90:32: a single-precision literal


@pytest.fixture(scope="module")
def so():
    from oracle import SweepOracle
    s = SweepOracle()
    yield s
    s.set_leshift(0.0, 0.0); s.set_swetnam(False); s.set_dd(False)


def start_state(c_oracle, boxes, beta, dref=0.0):
    hs = [b[0] for b in boxes]
    xs = [np.array(b[1]) for b in boxes]
    ivs = [c_oracle.ivects(h) for h in hs]
    lists = [c_oracle.neighbours(xs[l], ivs[l]) for l in range(2)]
    me = [c_oracle.model_energy(xs[l], ivs[l], *lists[l]) for l in range(2)]
    p = 1.0 / pin.AUP_TO_ATM
    v = [abs(np.linalg.det(h)) for h in hs]
    mu = me[0] + p * v[0] - me[1] - p * v[1]
    mu = mu - dref                                                          # main.f90:173
    mu = mu * beta - 48.0 * np.log(v[0] / v[1])
    return hs, xs, ivs, lists, me, mu, p


def replay(so, c_oracle, boxes, cycles, grid, wl_factor, ls=1, dref=0.0, weight=None, eq=1, per_cycle=None, walker=0):
    """mc_cycle with translations, mc_update_wl_bins (from cycle `eq` on) and a switch attempt after every move."""
    from mc_water_ls_mw_amd.lattice import ANG_TO_BOHR
    from mc_water_ls_mw_amd.sweep import KB
    beta = 1.0 / (KB * 200.0)
    hs, xs, ivs, lists, me, mu, p = start_state(c_oracle, boxes, beta, dref)
    w = np.zeros(grid.nbins) if weight is None else np.array(weight, dtype=float)
    hi, uh = np.zeros(grid.nbins), np.zeros(grid.nbins)
    for cyc in range(1, cycles + 1):
        if cyc % 10 == 0:
            lists = [c_oracle.neighbours(xs[l], ivs[l]) for l in range(2)]
        f = wl_factor(cyc) if callable(wl_factor) else wl_factor
        r = so.cycle(48, pin.SEED, walker, (cyc - 1) * 48, hs, xs, beta, 1.1 * ANG_TO_BOHR, grid, w, hi, uh, ls=ls, ls_mu=mu,
                     model_energy=me, lists=lists, record=cyc >= eq, samplerun=False, always_switch=True, npt=False,
                     wl_factor=f, pressure=p)
        xs = [r["xyz"][0], r["xyz"][1]]
        me, ls, mu, w, hi, uh = list(r["model_energy"]), r["ls"], r["ls_mu"], r["weight"], r["histogram"], r["unbiased_hist"]
        if per_cycle is not None:
            per_cycle(cyc, hi, w)
    return np.array(xs), ls, mu, w, hi


def test_leshift_matches_the_reference_program(tmp_path, so, c_oracle):
    """leshift = .true.: ref_enthalpy = the initial model energies (NVT: no PV term, main.f90:146-147), so the run starts
    at mu = -N log(V1/V2) + beta P (V1 - V2) instead of ~ -110 and every lattice switch sees the shifted enthalpy
    difference.  The replay without the shift must NOT reproduce the run (the option matters on this input)."""
    from mc_water_ls_mw_amd.sweep import KB, MuGrid
    grid = MuGrid(101, -400.0, 400.0)
    boxes, e_ref, ljr, ls, hist, wgt = pin.run_reference(str(tmp_path / "run"), 2, 200, 40, samplerun=False, always_switch=True,
                                                         tables=True, mc_extra="leshift = .true.")
    beta = 1.0 / (KB * 200.0)
    _, _, _, _, me0, _, _ = start_state(c_oracle, boxes, beta)
    so.set_leshift(me0[0], me0[1])
    try:
        xs, ls_or, mu, w, hi = replay(so, c_oracle, boxes, 40, grid, F0, dref=me0[0] - me0[1])
    finally:
        so.set_leshift(0.0, 0.0)
    assert np.abs(xs - ljr).max() < 1e-10 and ls == ls_or
    assert hist.sum() > 0 and np.allclose(hi, hist, rtol=1e-12, atol=1e-12)
    assert np.allclose(w, wgt, rtol=1e-11, atol=1e-12)
    xs0, _, _, w0, hi0 = replay(so, c_oracle, boxes, 40, grid, F0)          # leshift off: another trajectory
    assert np.abs(hi0 - hist).max() > 0.5


def test_swetnam_increment_matches_the_reference_program(tmp_path, so, c_oracle):
    """wl_swetnam = .true., wl_alpha = 0.01: after every move the increment is min(orig, alpha nbins log(rms deviation of
    the histogram from flat)); the checkpoint holds the last one next to histogram and weights."""
    from mc_water_ls_mw_amd.sweep import MuGrid
    grid = MuGrid(101, -400.0, 400.0)
    boxes, e_ref, ljr, ls, hist, wgt = pin.run_reference(str(tmp_path / "run"), 2, 200, 30, samplerun=False, always_switch=True,
                                                         tables=True, mc_extra="wl_swetnam = .true.\nwl_alpha = 0.01")
    ref_factor = struct.unpack("<d", pin.run_reference.records[3])[0]
    so.set_swetnam(True, 0.01, F0, -400.0, 400.0, 0.0)      # (namelist values are read in double precision)
    try:
        xs, ls_or, mu, w, hi = replay(so, c_oracle, boxes, 30, grid, F0)
        sumhist, factor = so.get_swetnam()
    finally:
        so.set_swetnam(False)
    assert sumhist == 30 * 48 and factor == pytest.approx(ref_factor, rel=1e-12) and factor < 0.0
    assert np.abs(xs - ljr).max() < 1e-10 and ls == ls_or
    assert np.allclose(hi, hist, rtol=1e-12, atol=1e-12) and np.allclose(w, wgt, rtol=1e-10, atol=1e-11)


PAR = "parallel_strategy = 'dd'\nwindow_overlap    = 2"


def dd_window(grid, rank, size, overlap):
    from mc_water_ls_mw_amd.sweep import MuGrid   # noqa: F401
    return grid.window(rank, size, overlap)


@pytest.mark.parametrize("rank", [0, 1, 2])
def test_dd_window_assignment_matches_the_reference_program(tmp_path, rank):
    """Rank r of three: my_start_bin / my_end_bin / my_mu_min / my_mu_max and the active lattice (mc_moves.F90:659-709),
    read from the rank's log."""
    from mc_water_ls_mw_amd.sweep import MuGrid
    grid = MuGrid(101, -400.0, 400.0)
    d = str(tmp_path / "run")
    pin.run_reference(d, 2, 200, 2, samplerun=False, always_switch=True, tables=True, par_extra=PAR, book_extra="eq_mc_cycles = 5",
                      run_env=dict(MW_WRAP_SIZE="3", MW_WRAP_RANK=str(rank)))
    log = open(os.path.join(d, "node000.log")).read()          # (opened before the rank is poked)
    m = re.search(r"This rank will use bin\s+(\d+)\s+to\s+(\d+)", log)
    lo = re.search(r"Lower limit of mu:\s*([-0-9.]+)", log)
    hi = re.search(r"Upper limit of mu:\s*([-0-9.]+)", log)
    w = grid.window(rank, 3, 2)
    assert (int(m.group(1)), int(m.group(2))) == (w["start_bin"], w["end_bin"])
    assert float(lo.group(1)) == pytest.approx(w["mu_min"], abs=1e-6) and float(hi.group(1)) == pytest.approx(w["mu_max"], abs=1e-6)
    ls_ref = struct.unpack("<i", pin.run_reference.records[-1])[0]
    assert ls_ref == w["ls"] if w["ls"] is not None else ls_ref == 1


def test_dd_run_in_its_window_matches_the_reference_program(tmp_path, so, c_oracle):
    """Rank 0 of two (window: the negative half of mu plus the overlap; the Ic/Ih pair starts at mu ~ -330, inside it):
    two equilibration cycles without Wang-Landau updates or switch attempts, then weight generation confined to the
    window -- a move that would leave it meets eta = huge and is rejected -- with a flatness check per window every 8
    cycles (wl_schedule 1, wl_minhist -1: first a reset, then 'flat' every time: histogram zeroed, increment halved, no
    weight shift and no wlf.dat in 'dd')."""
    from mc_water_ls_mw_amd.sweep import MuGrid
    from oracle import schedule as osch
    grid = MuGrid(101, -400.0, 400.0)
    w0 = grid.window(0, 2, 2)
    gw = grid.restricted(w0)
    d = str(tmp_path / "run")
    boxes, e_ref, ljr, ls, hist, wgt = pin.run_reference(d, 2, 200, 40, samplerun=False, always_switch=True, tables=True,
                                                         par_extra=PAR, mc_extra="wl_schedule = 1\nwl_minhist = -1",
                                                         book_extra="eq_mc_cycles = 3\nflat_chk_int = 8",
                                                         run_env=dict(MW_WRAP_SIZE="2", MW_WRAP_RANK="0", MW_WRAP_SWITCH_FROM_MOVE=str(2 * 48)))
    ref_factor = struct.unpack("<d", pin.run_reference.records[3])[0]
    st = osch.new_state(F0, schedule=1, minhist=-1)
    events = []

    def per_cycle(cyc, hi, w):
        if cyc % 8 == 0:
            events.append(osch.flatness_step(st, cyc, 48, hi, w, start_bin=w0["start_bin"], end_bin=w0["end_bin"], dd=True))

    so.set_dd(True, 3, False)
    try:
        xs, ls_or, mu, w, hi = replay(so, c_oracle, boxes, 40, gw, lambda cyc: st["wl_factor"], eq=3, per_cycle=per_cycle)
        in_window, failed = so.get_dd()
    finally:
        so.set_dd(False)
    assert in_window and not failed
    assert events == ["first reset", "halved", "halved", "halved", "halved"] and st["wl_factor"] == ref_factor == F0 / 16
    assert np.abs(xs - ljr).max() < 1e-10 and ls == ls_or == 1
    assert np.allclose(hi, hist, rtol=0, atol=1e-12) and np.allclose(w, wgt, rtol=1e-11, atol=1e-12)
    assert wgt[w0["end_bin"]:].max() == 0.0 and wgt[:w0["end_bin"]].max() > 0      # nothing outside the window
    assert not os.path.exists(os.path.join(d, "wlf.dat"))



def test_dd_npt_run_in_its_window_matches_the_reference_program(tmp_path, so, c_oracle):
    """Rank 0 of two under NPT (~1 move in 6 a volume move): during the two equilibration cycles no switch attempt follows
    a volume move either (mc_moves.F90:243-248 sits below both branches), and mc_volume sees the window's walls."""
    from mc_water_ls_mw_amd.lattice import ANG_TO_BOHR
    from mc_water_ls_mw_amd.sweep import KB, MuGrid
    from oracle import FullSweepState
    grid = MuGrid(101, -400.0, 400.0)
    w0 = grid.window(0, 2, 2)
    vol_prob = 0.1
    transP = 0.5 / (0.5 + vol_prob)
    boxes, e_ref, ljr, ls, hist, wgt = pin.run_reference(str(tmp_path / "run"), 2, 200, 12, samplerun=False, always_switch=True,
                                                         tables=True, npt=True, vol_prob=vol_prob, transP=transP, par_extra=PAR,
                                                         book_extra="eq_mc_cycles = 3",
                                                         run_env=dict(MW_WRAP_SIZE="2", MW_WRAP_RANK="0", MW_WRAP_SWITCH_FROM_MOVE=str(2 * 48)))
    beta, p = 1.0 / (KB * 200.0), 1.0 / pin.AUP_TO_ATM
    st = FullSweepState(c_oracle, [b[0] for b in boxes], [b[1] for b in boxes])
    mu = st.model_energy[0] + p * st.volume[0] - st.model_energy[1] - p * st.volume[1]
    st.ls_mu = mu * beta - 48.0 * np.log(st.volume[0] / st.volume[1])
    w, hi, uh = np.zeros(101), np.zeros(101), np.zeros(101)
    so.set_dd(True, 3, False)
    try:
        for cyc in range(1, 13):
            if cyc % 10 == 0:
                st.rebuild_lists(c_oracle)
            so.full(st, 48, pin.SEED, 0, (cyc - 1) * 48, transP, 0.924 * ANG_TO_BOHR, beta, 1.1 * ANG_TO_BOHR, grid.restricted(w0),
                    w, hi, uh, record=cyc >= 3, samplerun=False, always_switch=True, npt=True, wl_factor=F0, pressure=p)
        in_window, failed = so.get_dd()
    finally:
        so.set_dd(False)
    assert in_window and not failed and st.nvol[0] > 40 and 0 < st.nvol[1]
    assert np.abs(st.h - pin.run_reference.hmatrix).max() < 1e-10
    assert np.abs(st.xyz - ljr).max() < 1e-9 and st.ls == ls
    assert hist.sum() > 0 and np.allclose(hi, hist, rtol=1e-12, atol=1e-12) and np.allclose(w, wgt, rtol=1e-10, atol=1e-11)


def replay_with_step_adjustment(so, c_oracle, boxes, cycles, eq, monitor_int, transP, target=0.5, walker=0):
    """NPT mc_cycle loop with mc_monitor_stats' equilibration tuning (mc_moves.F90:1724-1732): every monitor_int cycles
    below eq_mc_cycles, max_trans *= (accepted / attempted translations since the last report) / target (at least 0.1
    bohr), dv_max likewise from the volume moves (at least 1e-4 bohr); and the stored energies are replaced by freshly
    computed ones (:1783-1787)."""
    from mc_water_ls_mw_amd.lattice import ANG_TO_BOHR
    from mc_water_ls_mw_amd.sweep import KB, MuGrid
    from oracle import FullSweepState
    grid = MuGrid(101, -400.0, 400.0)
    beta, p = 1.0 / (KB * 200.0), 1.0 / pin.AUP_TO_ATM
    st = FullSweepState(c_oracle, [b[0] for b in boxes], [b[1] for b in boxes])
    mu = st.model_energy[0] + p * st.volume[0] - st.model_energy[1] - p * st.volume[1]
    st.ls_mu = mu * beta - 48.0 * np.log(st.volume[0] / st.volume[1])
    w, hi, uh = np.zeros(101), np.zeros(101), np.zeros(101)
    max_trans, dv_max = 1.1 * ANG_TO_BOHR, 0.924 * ANG_TO_BOHR
    acc0, vol0, moves0 = 0, np.zeros(2, dtype=np.int64), 0
    history = []
    for cyc in range(1, cycles + 1):
        if cyc % 10 == 0:
            st.rebuild_lists(c_oracle)
        so.full(st, 48, pin.SEED, walker, (cyc - 1) * 48, transP, dv_max, beta, max_trans, grid, w, hi, uh,
                record=cyc >= eq, samplerun=False, always_switch=True, npt=True, wl_factor=F0, pressure=p)
        if cyc % monitor_int == 0:
            att_v, acc_v = int(st.nvol[0] - vol0[0]), int(st.nvol[1] - vol0[1])
            att_t, acc_t = cyc * 48 - moves0 - att_v, int(st.accepted - acc0)
            if cyc < eq:
                max_trans = max(max_trans * (acc_t / att_t) / target, 0.1)
                dv_max = max(dv_max * (acc_v / att_v) / target, 0.0001)
            history.append((cyc, max_trans, dv_max))
            for l in range(2):                                                 # compute_model_energy(ils), :1783-1787
                st.model_energy[l] = c_oracle.model_energy(st.xyz[l], st.iv(l), *st.lists[l])
            acc0, vol0, moves0 = st.accepted, st.nvol.copy(), cyc * 48
    return st, w, hi, max_trans, dv_max, history


def test_equilibration_step_size_tuning_matches_the_reference_program(tmp_path, so, c_oracle):
    """eq_adjust_mc = .true. as in every shipped example: 13 equilibration cycles with a report every 4, then 7 more.
    The checkpoint holds the tuned mc_max_trans / mc_dv_max next to the configuration they produced."""
    vol_prob = 0.1
    transP = 0.5 / (0.5 + vol_prob)
    boxes, e_ref, ljr, ls, hist, wgt = pin.run_reference(str(tmp_path / "run"), 2, 200, 20, samplerun=False, always_switch=True,
                                                         tables=True, npt=True, vol_prob=vol_prob, transP=transP,
                                                         book_extra="eq_adjust_mc = .true.\nmonitor_int = 4\neq_mc_cycles = 13")
    ref_steps = np.frombuffer(pin.run_reference.records[2], dtype="<f8")
    st, w, hi, max_trans, dv_max, history = replay_with_step_adjustment(so, c_oracle, boxes, 20, 13, 4, transP)
    assert len(history) == 5 and history[2][1] != history[0][1]                 # tuned at cycles 4, 8, 12; left alone at 16, 20
    assert ref_steps[0] == pytest.approx(max_trans, rel=1e-12) and ref_steps[1] == pytest.approx(dv_max, rel=1e-12)
    assert max_trans != pytest.approx(1.1 * 1.8897259886, rel=1e-3)
    assert np.abs(st.h - pin.run_reference.hmatrix).max() < 1e-10
    assert np.abs(st.xyz - ljr).max() < 1e-9 and st.ls == ls
    assert np.allclose(hi, hist, rtol=1e-12, atol=1e-12) and np.allclose(w, wgt, rtol=1e-10, atol=1e-11)
