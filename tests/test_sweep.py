"""The translation-move driver.

CPU: the oracle's helpers against the reference formulas; GPU: the device-resident sweep against
the oracle's restatement of mc_water_translation on the same Philox stream -- the same molecule
picked, the same accept/reject decision and the same energies move by move, the same final
positions, order parameter and accumulated energies."""
import numpy as np
import pytest

from conftest import DE_ATOL, RTOL, load_golden


@pytest.fixture(scope="module")
def so():
    from oracle import SweepOracle
    return SweepOracle()


def test_mu_grid_matches_python_restatement(so):
    from mc_water_ls_mw_amd.sweep import MuGrid
    g = MuGrid(101, -400.0, 400.0)
    mb, bw, gp = so.mu_grid(101, -400.0, 400.0)
    assert np.allclose(mb, g.mu_bin, rtol=1e-14, atol=1e-12) and np.allclose(bw, g.binwidth, rtol=1e-13)
    assert gp[0] == pytest.approx(g.r_pos, rel=1e-15) and gp[2] == pytest.approx(g.r_neg, rel=1e-15)
    # the bins tile [mu_min, mu_max]: half-widths accumulate to the range (mc_moves.F90:571-656)
    assert bw.sum() == pytest.approx(800.0, rel=1e-12)
    assert mb[50] == 0.0 and bw[50] == 1.0
    edges = np.concatenate([[-400.0], -400.0 + np.cumsum(bw)])
    for k in range(101):                                   # a bin's centre maps back to that bin
        assert so.mu_to_bin(g, mb[k]) == k + 1
        assert edges[k] < mb[k] < edges[k + 1]


def test_eta_weight_interpolates_and_walls(so):
    from mc_water_ls_mw_amd.sweep import MuGrid
    g = MuGrid(101, -400.0, 400.0)
    w = 0.01 * g.mu_bin ** 2 / 100.0
    for k in (0, 10, 50, 77, 100):
        assert so.eta_weight(g, w, True, g.mu_bin[k]) == pytest.approx(w[k], rel=1e-13, abs=1e-15)
        assert so.eta_weight(g, w, False, g.mu_bin[k]) == w[k]
    assert so.eta_weight(g, w, True, 400.5) == np.finfo(float).max      # huge(1.0_dp): hard wall
    assert so.eta_weight(g, w, True, -400.5) == np.finfo(float).max
    mid = 0.5 * (g.mu_bin[60] + g.mu_bin[61])
    lo, hi = sorted((w[60], w[61]))
    assert lo <= so.eta_weight(g, w, True, mid) <= hi


def test_uniforms_are_a_counter_based_stream(so):
    a = so.uniforms(12345, 3, 77)
    assert np.array_equal(a, so.uniforms(12345, 3, 77))
    assert not np.array_equal(a, so.uniforms(12345, 4, 77)) and not np.array_equal(a, so.uniforms(12346, 3, 77))
    u = np.array([so.uniforms(1, 0, m) for m in range(4000)])
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01


def test_oracle_sweep_keeps_the_energy_books(so, c_oracle):
    """The accumulated model_energy (caller-side bookkeeping, mc_moves.F90:1013-1016,1087,1190) stays equal to a
    fresh full-box energy as long as the list is fresh enough -- the reference's own drift check."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.sweep import KB
    h, x = lat.ice_box("ih", (3, 2, 2), 0.1, seed=5)
    r = so.sweep(150, seed=9, walker=0, move0=0, hs=[h], xs=[x], beta=1.0 / (KB * 220.0), max_trans=0.3)
    assert 0 < r["accepted"] < 150
    iv = c_oracle.ivects(h)
    fresh = c_oracle.model_energy(r["xyz"][0], iv, *r["lists"][0])
    assert abs(r["model_energy"][0] - fresh) < 1e-10


def _farm(boxes, nlat, temperature, max_trans_ang, grid=None, weight=None):
    from mc_water_ls_mw_amd.energy import load_boxes
    from mc_water_ls_mw_amd.sweep import WalkerFarm
    em = load_boxes([b[0] for b in boxes], [b[1] for b in boxes])
    return em, WalkerFarm(em, nlat, temperature, max_trans_ang, grid=grid, weight=weight)


def _compare(log_gpu, ref, farm_state, xyz_gpu):
    assert np.array_equal(log_gpu[:, 0], ref["log"][:, 0])                     # same molecule every move
    assert np.array_equal(log_gpu[:, 1], ref["log"][:, 1])                     # same accept / reject
    for c in (2, 3, 4, 5):
        assert np.all(np.abs(log_gpu[:, c] - ref["log"][:, c]) <= RTOL * np.abs(ref["log"][:, c]) + 1e-14)
    assert np.all(np.abs((log_gpu[:, 3] - log_gpu[:, 2]) - (ref["log"][:, 3] - ref["log"][:, 2])) <= DE_ATOL)
    assert np.all(np.abs(log_gpu[:, 6] - ref["log"][:, 6]) <= 1e-7 * (1 + np.abs(ref["log"][:, 6])))   # ls_mu = beta*dE sums
    assert farm_state["accepted"] == ref["accepted"] and farm_state["ls"] == ref["ls"]
    for l, x in enumerate(xyz_gpu):
        assert np.abs(x - ref["xyz"][l]).max() < 1e-10
        assert abs(farm_state["model_energy"][l] - ref["model_energy"][l]) <= 1e-9


@pytest.mark.gpu
def test_single_box_walkers_follow_the_oracle(so):
    from mc_water_ls_mw_amd import lattice as lat
    h, x0 = lat.ice_box("ih", (3, 2, 2), 0.0)
    boxes = [(h, lat.thermalise(x0, 0.12, 40 + w)) for w in range(6)]
    em, farm = _farm(boxes, 1, 220.0, 1.1)
    try:
        for w in range(1, 7):
            farm.set_state(w, 1, 0.0)
        log = farm.sweep(120, seed=2024, move0=7, log=True)
        # a second launch continues the same chains (move numbers go on)
        log2 = farm.sweep(60, seed=2024, move0=127, log=True)
        for w in range(6):
            ref = so.sweep(180, 2024, w, 7, [h], [boxes[w][1]], farm.beta, farm.max_trans)
            _compare(np.concatenate([log[w], log2[w]]), ref, farm.state(w + 1), [farm.positions(w + 1)])
    finally:
        em.energy_deinit()


@pytest.mark.gpu
def test_lattice_switch_walkers_follow_the_oracle(so):
    """Two lattices per walker (Ic / Ih, the ice1_sample pair): fractional mapping of the move into the partner
    lattice, order parameter, multicanonical weights with interpolation."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.sweep import MuGrid
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    grid = MuGrid(101, -400.0, 400.0)
    weight = 3.0 * np.exp(-(grid.mu_bin / 150.0) ** 2) + 0.002 * np.abs(grid.mu_bin)
    boxes = []
    for w in range(4):
        boxes += [(z1["h"], lat.thermalise(z1["xyz"], 0.08, 60 + w)), (z2["h"], lat.thermalise(z2["xyz"], 0.08, 80 + w))]
    em, farm = _farm(boxes, 2, 200.0, 1.1, grid=grid, weight=weight)
    try:
        mus = []
        for w in range(1, 5):
            ls = 1 + (w % 2)
            mus.append((ls, farm.initial_mu(w)))
            farm.set_state(w, ls, mus[-1][1])
        log = farm.sweep(160, seed=77, move0=0, log=True)
        for w in range(4):
            ls, mu = mus[w]
            ref = so.sweep(160, 77, w, 0, [z1["h"], z2["h"]], [boxes[2 * w][1], boxes[2 * w + 1][1]], farm.beta,
                           farm.max_trans, grid=grid, weight=weight, ls=ls, ls_mu=mu,
                           model_energy=[em.model_energy[2 * w], em.model_energy[2 * w + 1]])
            _compare(log[w], ref, farm.state(w + 1), [farm.positions(2 * w + 1), farm.positions(2 * w + 2)])
            assert 0 < ref["accepted"] < 160
    finally:
        em.energy_deinit()


@pytest.mark.gpu
def test_1536_pair_sweep_with_list_refresh(so, c_oracle):
    """BASELINE.json configs[2]: the 1536-molecule Ic/Ih pair; sweep, rebuild the lists on the device at the
    host's cadence (mc_moves.F90:218-222), sweep on -- and the oracle does the same."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.sweep import MuGrid
    z1, z2 = load_golden("ic1536"), load_golden("ih1536")
    x1, x2 = lat.thermalise(z1["xyz"], 0.1, 7), lat.thermalise(z2["xyz"], 0.1, 8)
    grid = MuGrid(101, -8000.0, 8000.0)
    em, farm = _farm([(z1["h"], x1), (z2["h"], x2)], 2, 200.0, 1.1, grid=grid)
    try:
        mu0 = farm.initial_mu(1)
        farm.set_state(1, 1, mu0)
        log_a = farm.sweep(100, seed=5, move0=0, log=True)[0]
        em.build_neighbours_batch(1, 2)
        log_b = farm.sweep(100, seed=5, move0=100, log=True)[0]
        hs = [z1["h"], z2["h"]]
        ra = so.sweep(100, 5, 0, 0, hs, [x1, x2], farm.beta, farm.max_trans, grid=grid, ls=1, ls_mu=mu0,
                      model_energy=list(em.model_energy))
        rb = so.sweep(100, 5, 0, 100, hs, [ra["xyz"][0], ra["xyz"][1]], farm.beta, farm.max_trans, grid=grid,
                      ls=ra["ls"], ls_mu=ra["ls_mu"], model_energy=list(ra["model_energy"]))
        rb["accepted"] += ra["accepted"]
        rb["log"] = np.concatenate([ra["log"], rb["log"]])
        _compare(np.concatenate([log_a, log_b]), rb, farm.state(1), [farm.positions(1), farm.positions(2)])
    finally:
        em.energy_deinit()


def _cycle_case(so, samplerun, wl_factor, weight_fn, seed):
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.sweep import MuGrid
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    grid = MuGrid(101, -400.0, 400.0)
    weight = weight_fn(grid)
    boxes = []
    for w in range(5):
        boxes += [(z1["h"], lat.thermalise(z1["xyz"], 0.06, 160 + w)), (z2["h"], lat.thermalise(z2["xyz"], 0.06, 180 + w))]
    em, farm = _farm(boxes, 2, 200.0, 1.1, grid=grid, weight=weight)
    try:
        farm.options(record=True, samplerun=samplerun, always_switch=True, npt=False, wl_factor=wl_factor)
        mus = [farm.initial_mu(w) for w in range(1, 6)]
        for w in range(1, 6):
            farm.set_state(w, 1, mus[w - 1])
        log = farm.sweep(200, seed=seed, move0=0, log=True)
        nsw = 0
        for w in range(5):
            ref = so.cycle(200, seed, w, 0, [z1["h"], z2["h"]], [boxes[2 * w][1], boxes[2 * w + 1][1]], farm.beta,
                           farm.max_trans, grid, weight, np.zeros(101), np.zeros(101), ls=1, ls_mu=mus[w],
                           model_energy=[em.model_energy[2 * w], em.model_energy[2 * w + 1]], record=True,
                           samplerun=samplerun, always_switch=True, wl_factor=wl_factor)
            _compare(log[w], ref, farm.state(w + 1), [farm.positions(2 * w + 1), farm.positions(2 * w + 2)])
            wt, hi, uh = farm.tables(w + 1)
            assert np.allclose(hi, ref["histogram"], rtol=1e-13, atol=1e-13) and hi.sum() > 0
            assert np.allclose(wt, ref["weight"], rtol=1e-11, atol=1e-12)
            assert np.allclose(uh, ref["unbiased_hist"], rtol=1e-9, atol=1e-300)
            assert farm.switches(w + 1) == ref["switches"]
            nsw += ref["switches"]
        return nsw
    finally:
        em.energy_deinit()


@pytest.mark.gpu
def test_wang_landau_cycle_on_device_follows_the_oracle(so):
    """Weight generation: every walker grows its own weights (Wang-Landau), attempts a switch after every move."""
    _cycle_case(so, samplerun=False, wl_factor=float(np.float32(0.05)), weight_fn=lambda g: np.zeros(g.nbins), seed=31)


@pytest.mark.gpu
def test_sampling_cycle_with_switches_on_device_follows_the_oracle(so):
    nsw = _cycle_case(so, samplerun=True, wl_factor=0.0, weight_fn=lambda g: 0.03 * np.abs(g.mu_bin), seed=32)
    assert nsw >= 0


@pytest.mark.gpu
def test_farm_synchronise_sums_the_walkers_increments(so):
    """The mpi_sync_int step for a farm: all walkers end with last + the sum of every walker's increment."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.comms import WalkerComms
    from mc_water_ls_mw_amd.sweep import MuGrid
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    grid = MuGrid(101, -400.0, 400.0)
    boxes = []
    for w in range(3):
        boxes += [(z1["h"], lat.thermalise(z1["xyz"], 0.05, 260 + w)), (z2["h"], lat.thermalise(z2["xyz"], 0.05, 280 + w))]
    em, farm = _farm(boxes, 2, 200.0, 1.1, grid=grid, weight=np.zeros(101))
    try:
        farm.options(record=True, samplerun=False, always_switch=True, wl_factor=0.05)
        for w in range(1, 4):
            farm.set_state(w, 1, farm.initial_mu(w))
        comms = WalkerComms(101)
        farm.sweep(96, seed=3)
        before = [farm.tables(w) for w in range(1, 4)]
        wt, hi, uh = farm.synchronise(comms)
        assert np.allclose(hi, sum(b[1] for b in before)) and np.allclose(wt, sum(b[0] for b in before))
        assert hi.sum() > 0 and wt.max() > 0
        for w in range(1, 4):
            t = farm.tables(w)
            assert np.array_equal(t[0], wt) and np.array_equal(t[1], hi)
        farm.sweep(48, seed=3, move0=96)
        mid = [farm.tables(w) for w in range(1, 4)]
        wt2, hi2, _ = farm.synchronise(comms)
        assert np.allclose(hi2, hi + sum(m[1] - hi for m in mid))
        assert np.allclose(wt2, wt + sum(m[0] - wt for m in mid))
    finally:
        em.energy_deinit()


@pytest.mark.gpu
def test_weight_generation_farm_runs_and_keeps_its_books():
    """configs[3] in miniature on one GPU: 64 Ic/Ih walkers, 30 cycles, lists every 10, sync every 10."""
    from mc_water_ls_mw_amd.comms import WalkerComms
    from mc_water_ls_mw_amd.farm import run
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    res = run([z1["h"], z2["h"]], [z1["xyz"], z2["xyz"]], walkers=64, cycles=30, mpi_sync_int=10, comms=WalkerComms(101))
    assert 0.02 < res["acceptance"] < 0.9
    assert max(abs(d) for d in res["drift_walker1_Ha"]) < 1e-9          # the reference's own consistency check
    wt, hi, _ = res["tables"]
    assert hi.sum() > 0 and wt.max() > 0 and np.all(wt >= 0.0)


@pytest.mark.gpu
def test_farm_loop_equals_oracle_replay_including_stale_list_drift(so, c_oracle):
    """Without synchronisation a farm walker is an isolated chain: 60 cycles of farm.run (lists rebuilt every 10
    cycles on the device) against the oracle replaying the same cycles.  With 1.1 A moves molecules outrun the
    Verlet skin between rebuilds, so the accumulated energy drifts away from a fresh full-box energy -- in the
    reference's algorithm itself; device and oracle must show the same drift."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.farm import run
    from mc_water_ls_mw_amd.sweep import KB, MuGrid
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    hs, cycles, seed = [z1["h"], z2["h"]], 60, 2025
    res = run(hs, [z1["xyz"], z2["xyz"]], walkers=4, cycles=cycles, comms=None, seed=seed, wl_factor=0.05)
    grid = MuGrid(101, -400.0, 400.0)
    beta = 1.0 / (KB * 200.0)
    xs = [lat.thermalise(z1["xyz"], 0.05, 0), lat.thermalise(z2["xyz"], 0.05, 1)]     # farm.run: walker 0 of rank 0
    ivs = [c_oracle.ivects(h) for h in hs]
    lists = [c_oracle.neighbours(xs[l], ivs[l]) for l in range(2)]
    me = [c_oracle.model_energy(xs[l], ivs[l], *lists[l]) for l in range(2)]
    v = [abs(np.linalg.det(h)) for h in hs]
    mu = (me[0] - me[1]) * beta - 48.0 * np.log(v[0] / v[1])
    ls, w, hi, uh = 1, np.zeros(101), np.zeros(101), np.zeros(101)
    for cyc in range(1, cycles + 1):
        if cyc % 10 == 0:
            lists = [c_oracle.neighbours(xs[l], ivs[l]) for l in range(2)]
        r = so.cycle(48, seed, 0, (cyc - 1) * 48, hs, xs, beta, 1.1 * lat.ANG_TO_BOHR, grid, w, hi, uh, ls=ls, ls_mu=mu,
                     model_energy=me, lists=lists, record=True, samplerun=False, always_switch=True, wl_factor=0.05)
        xs, me, ls, mu, w, hi, uh = [r["xyz"][0], r["xyz"][1]], list(r["model_energy"]), r["ls"], r["ls_mu"], r["weight"], r["histogram"], r["unbiased_hist"]
    fresh = [c_oracle.model_energy(xs[l], ivs[l], *lists[l]) for l in range(2)]
    drift_oracle = [me[l] - fresh[l] for l in range(2)]
    for l in range(2):
        assert abs(res["drift_walker1_Ha"][l] - drift_oracle[l]) < 1e-9


def _npt_against_oracle(so, c_oracle, boxes, nlat, nw, temperature=200.0, max_trans_ang=1.1, pressure_atm=1.0, vol_prob=0.1,
                        dv_max_ang=0.924, seed=11, nmoves=96, mu_range=400.0, weight0=None, samplerun=False, always_switch=True,
                        wl_factor=0.05, eta_interp=True, leshift=False, minu=False, log_unbiased_norm=0.0, swetnam_alpha=None,
                        dd=None):
    """`nw` walkers of `nlat` lattices each (boxes: nlat * nw (h, xyz) pairs): two launches of `nmoves` moves with volume moves on
    the device (vol_prob = 0: translations only, NVT acceptance), lists rebuilt in between after mw_sweep_sync_cells, against
    mwo_sweep_full walker by walker.  Two lattices: the run options of mc_cycle (weights, sample run, leshift, MINU, ...)."""
    from mc_water_ls_mw_amd.sweep import MuGrid
    from oracle import FullSweepState
    grid = MuGrid(101, -mu_range, mu_range)
    p_au = pressure_atm / 2.90363081e8
    npt = vol_prob > 0.0
    weight0 = np.zeros(101) if weight0 is None else np.asarray(weight0, dtype=np.float64)
    from mc_water_ls_mw_amd.energy import load_boxes
    from mc_water_ls_mw_amd.sweep import WalkerFarm
    em = load_boxes([b[0] for b in boxes], [b[1] for b in boxes])
    farm = WalkerFarm(em, nlat, temperature, max_trans_ang, grid=grid, weight=weight0.copy(), eta_interp=eta_interp, pressure_au=p_au)
    ref_h = (0.0, 0.0)
    try:
        if nlat == 2:
            farm.options(record=True, samplerun=samplerun, always_switch=always_switch, npt=npt, wl_factor=wl_factor,
                         log_unbiased_norm=log_unbiased_norm)
            if leshift:
                ref_h = farm.starting_enthalpy(1, npt)
                farm.leshift(ref_h)
            if minu:
                farm.minu(True)
            wins = None
            if dd is not None:                                       # dd = (windows in all, overlap, eq_mc_cycles): walker w = window w mod size
                wins = [grid.window(w % dd[0], dd[0], dd[1]) for w in range(nw)]
                farm.set_windows(wins)
                farm.dd(True, dd[2])
                wrow = np.tile(weight0, (nw, 1))                     # mc_moves.F90:808-812: only the window's part of the weights
                for k, w_ in enumerate(wins):
                    wrow[k, :w_["start_bin"] - 1] = 0.0
                    wrow[k, w_["end_bin"]:] = 0.0
                farm.set_tables_range(1, weight=wrow)
                farm.set_factors(wl_factor=np.full(nw, wl_factor))
                farm.options(record=True, samplerun=samplerun, always_switch=always_switch, npt=npt, wl_factor=0.0,
                             log_unbiased_norm=log_unbiased_norm)      # (per-walker increments, as farm.run's 'dd' branch)
            if swetnam_alpha is not None:                            # (as farm.run: every walker its own increment, starting at wl_factor)
                farm.swetnam(True, swetnam_alpha, wl_factor)
                farm.set_factors(wl_factor=np.full(nw, wl_factor))
        else:
            em._chk(em.L.mw_sweep_options(0, 1, 0, 1, __import__("ctypes").c_double(grid.av_binwidth),
                                          __import__("ctypes").c_double(0.0), __import__("ctypes").c_double(0.0),
                                          __import__("ctypes").c_double(p_au)))
        if npt:
            farm.moves(trans_prob=0.5, vol_prob=vol_prob, dv_max_ang=dv_max_ang)
        mus = [farm.initial_mu(w) for w in range(1, nw + 1)]
        for w in range(1, nw + 1):
            farm.set_state(w, 1, mus[w - 1])
        e0 = em.model_energy.copy()
        log_a = farm.sweep(nmoves, seed=seed, move0=0, log=True)
        hdev = farm.sync_cells() if npt else np.array(em.hmatrix)
        em.build_neighbours_batch(1, nlat * nw)
        log_b = farm.sweep(nmoves, seed=seed, move0=nmoves, log=True)
        if npt:
            hdev = farm.sync_cells()
        nvol = 0
        transP, dvm = (farm.transP, farm.dv_max) if npt else (1.0, 0.0)
        if leshift:
            so.set_leshift(*ref_h)
        so.set_minu(minu)
        try:
            refs = []
            for w in range(nw):
                bx = boxes[nlat * w:nlat * w + nlat]
                st = FullSweepState(c_oracle, [b[0] for b in bx], [b[1] for b in bx])
                st.model_energy[:] = e0[nlat * w:nlat * w + nlat]
                st.ls_mu = mus[w]
                wt, hi, uh = weight0.copy(), np.zeros(101), np.zeros(101)
                go = grid
                if dd is not None:
                    go = grid.restricted(wins[w])
                    wt[:wins[w]["start_bin"] - 1] = 0.0
                    wt[wins[w]["end_bin"]:] = 0.0
                    so.set_dd(True, dd[2], False)
                if swetnam_alpha is not None:
                    so.set_swetnam(True, swetnam_alpha, wl_factor, -mu_range, mu_range, 0.0)
                kw = dict(record=nlat == 2, samplerun=samplerun and nlat == 2, always_switch=always_switch and nlat == 2, npt=npt,
                          wl_factor=wl_factor, pressure=p_au, eta_interp=eta_interp, log_unbiased_norm=log_unbiased_norm)
                la = so.full(st, nmoves, seed, w, 0, transP, dvm, farm.beta, farm.max_trans, go, wt, hi, uh, **kw)
                st.rebuild_lists(c_oracle)
                lb = so.full(st, nmoves, seed, w, nmoves, transP, dvm, farm.beta, farm.max_trans, go, wt, hi, uh, **kw)
                refs.append((st, wt, hi, uh, np.concatenate([la, lb]), so.get_swetnam() if swetnam_alpha is not None else None,
                             so.get_dd() if dd is not None else None))
        finally:
            so.set_leshift(0.0, 0.0)
            so.set_minu(False)
            so.set_swetnam(False)
            so.set_dd(False)
        for w in range(nw):
            st, wt, hi, uh, ref, swet, ddst = refs[w]
            if ddst is not None:                                     # walker_in_window, and "not all walkers have reached their window"
                assert bool(farm.factors(w + 1, 1)[2][0]) == ddst[0]
                assert (em.L.mw_sweep_check_flags(w + 1, 1) != 0) == ddst[1]
            if swet is not None:                                     # Swetnam's visit total and the increment it led to
                f, sh, _ = farm.factors(w + 1, 1)
                assert sh[0] == swet[0]
                # (a walker that never visits a bin -- outside the order-parameter range from its first move to its last; three
                #  seeds in five thousand of tools/fuzz_soak.py -- never recomputes its increment: it keeps the one it started with.
                #  The oracle's single set of module variables still holds the previous walker's value then.)
                assert f[0] == pytest.approx(swet[1] if swet[0] > 0.0 else wl_factor, rel=1e-9, abs=1e-12)
            dev = np.concatenate([log_a[w], log_b[w]])
            assert np.array_equal(dev[:, 0], ref[:, 0]) and np.array_equal(dev[:, 1], ref[:, 1])     # molecule, outcome flags
            nvol += int(st.nvol[0])
            assert farm.volume_moves(w + 1) == (int(st.nvol[0]), int(st.nvol[1]))
            assert np.allclose(dev[:, 2:7], ref[:, 2:7], rtol=1e-9, atol=1e-9)
            for l in range(nlat):
                assert np.abs(hdev[nlat * w + l] - st.h[l]).max() < 1e-11
                assert np.abs(farm.positions(nlat * w + l + 1) - st.xyz[l]).max() < 1e-9
            s = farm.state(w + 1)
            assert s["ls"] == st.ls and s["accepted"] == st.accepted
            assert np.allclose(s["model_energy"], st.model_energy, rtol=0, atol=1e-9)
            if nlat == 2:
                t = farm.tables(w + 1)
                assert np.allclose(t[0], wt, rtol=1e-10, atol=1e-11) and np.allclose(t[1], hi, rtol=1e-12)
                if samplerun:      # (entries are exp(eta(mu) - norm): a steep table turns mu's last digits into 1e-9 relative)
                    assert np.allclose(t[2], uh, rtol=1e-6, atol=1e-300)
        return nvol
    finally:
        em.energy_deinit()


@pytest.mark.gpu
@pytest.mark.parametrize("nlat,scale", [(2, 1.0), (1, 1.0), (2, 0.8)])
def test_npt_sweep_with_volume_moves_on_device_follows_the_oracle(so, c_oracle, nlat, scale):
    """The whole move set on the device: translations, volume moves (both cells change, positions rescaled, image
    vectors rebuilt, full-box energies with the existing lists, restore on rejection), Wang-Landau updates, lattice
    switches; lists rebuilt in between after mw_sweep_sync_cells.  Walker by walker against mwo_sweep_full.
    scale = 0.8: the cells compressed to twice the density -- a dozen neighbours inside the cutoff (the volume move's
    full-box energy overflows its in-range queue and rescans), list rows past the 32 entries the LDS copies hold."""
    from mc_water_ls_mw_amd import lattice as lat
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    boxes, nw = [], 4
    for w in range(nw):
        boxes.append((z1["h"] * scale, lat.thermalise(z1["xyz"], 0.06, 360 + w) * scale))
        if nlat == 2:
            boxes.append((z2["h"] * scale, lat.thermalise(z2["xyz"], 0.06, 380 + w) * scale))
    assert _npt_against_oracle(so, c_oracle, boxes, nlat, nw) > 10 * nw


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(8)) + [503])
def test_npt_driver_on_random_lattice_pairs(so, c_oracle, seed):
    """Seeded variety for the volume-move path: Ic / Ih cells of equal molecule count, replicated, scaled between twice and
    0.8 times the ice density, sheared, at random temperature, pressure (up to 10^4 atm: cells that do shrink) and step sizes.
    Seed 503 (found by tools/fuzz_soak.py): a compressed pair whose order parameter starts at -666, OUTSIDE the +-400 range --
    eta_weight is huge(1.0_dp) there, mc_lattice_switch adds and subtracts it one after the other (mc_moves.F90:1561-1563), the
    energy terms are absorbed and the reference switches lattice on every attempt."""
    from mc_water_ls_mw_amd import lattice as lat
    rng = np.random.default_rng(4200 + seed)
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    reps = [(1, 1, 1), (1, 1, 2), (2, 1, 1), (1, 2, 2)][int(rng.integers(0, 4))]
    nlat = 2 if rng.random() < 0.75 else 1
    scale = float(rng.uniform(0.8, 1.08))
    boxes, nw = [], 2
    shear = np.eye(3) + (rng.uniform(-0.04, 0.04, (3, 3)) * (1 - np.eye(3)) if rng.random() < 0.5 else 0.0)
    for w in range(nw):
        for l, z in enumerate((z1, z2)[:nlat]):
            h, x = lat.replicate(z["h"], z["xyz"], reps)
            x = lat.thermalise(x, float(rng.uniform(0.02, 0.1)), 900 + 10 * w + l)
            boxes.append((np.ascontiguousarray(h @ shear) * scale, np.ascontiguousarray(x @ shear) * scale))
    nvol = _npt_against_oracle(so, c_oracle, boxes, nlat, nw, temperature=float(rng.uniform(150.0, 350.0)),
                               max_trans_ang=float(rng.uniform(0.3, 1.1)), pressure_atm=float(10.0 ** rng.uniform(0.0, 4.0)),
                               vol_prob=float(rng.uniform(0.05, 0.3)), dv_max_ang=float(rng.uniform(0.2, 1.2)), seed=50 + seed, nmoves=80)
    assert nvol > 4


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(12)) + [3544])      # (3544: a walker that never visits a bin, with Swetnam's rule on -- found by tools/fuzz_soak.py)
def test_driver_run_options_on_random_pairs(so, c_oracle, seed):
    """The run options of mc_cycle in random combination on the 48-molecule Ic / Ih pair (scaled 0.85 .. 1.05): NVT or NPT, a
    weight table of random size (bumps of a few kT up to 10^8 kT, where the switch's eta terms absorb the energies), an order-
    parameter range that may leave the walkers outside it, weight generation or a sample run, leshift, MINU, interpolation off."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.sweep import MuGrid
    rng = np.random.default_rng(6100 + seed)
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    scale = float(rng.uniform(0.85, 1.05))
    boxes, nw = [], 2
    for w in range(nw):
        for l, z in enumerate((z1, z2)):
            boxes.append((z["h"] * scale, lat.thermalise(z["xyz"], float(rng.uniform(0.02, 0.1)), 700 + 10 * w + l) * scale))
    mu_range = float(rng.choice([60.0, 400.0, 400.0, 3000.0]))
    grid = MuGrid(101, -mu_range, mu_range)
    amp = float(10.0 ** rng.uniform(0.0, 8.0)) if rng.random() < 0.5 else float(rng.uniform(0.0, 20.0))
    weight0 = amp * (np.exp(-(grid.mu_bin / (0.3 * mu_range)) ** 2) + 0.3 * rng.random(101))
    if rng.random() < 0.3:                                        # (tables with negative entries: what many walkers exchanging with the
        weight0 = weight0 - float(rng.uniform(0.0, 2.0)) * amp    #  reference's own arithmetic end up with, WalkerFarm.synchronise)
    samplerun = bool(rng.random() < 0.4)
    npt = bool(rng.random() < 0.5)
    _npt_against_oracle(so, c_oracle, boxes, 2, nw, temperature=float(rng.uniform(150.0, 300.0)), max_trans_ang=float(rng.uniform(0.3, 1.1)),
                        pressure_atm=float(10.0 ** rng.uniform(0.0, 3.0)), vol_prob=float(rng.uniform(0.05, 0.3)) if npt else 0.0,
                        dv_max_ang=float(rng.uniform(0.2, 1.0)), seed=90 + seed, nmoves=80, mu_range=mu_range, weight0=weight0,
                        samplerun=samplerun, always_switch=bool(rng.random() < 0.8), wl_factor=0.0 if samplerun else float(rng.uniform(0.001, 0.5)),
                        eta_interp=bool(rng.random() < 0.8), leshift=bool(rng.random() < 0.4), minu=bool(rng.random() < 0.3),
                        log_unbiased_norm=float(rng.uniform(0.0, 5.0)) if samplerun else 0.0,
                        swetnam_alpha=float(10.0 ** rng.uniform(-3.0, 0.0)) if (not samplerun and rng.random() < 0.3) else None)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_driver_window_decomposition_on_random_pairs(so, c_oracle, seed):
    """parallel_strategy = 'dd' in the driver on its own (mc_moves.F90:181-210,243-248,659-709): each walker confined to one of 2-5
    windows of the order-parameter range (which its start may or may not lie in), an equilibration period of 0-3 cycles during which
    a walker outside its window carries no weight and attempts no switch, the flag for one still outside at its end."""
    from mc_water_ls_mw_amd import lattice as lat
    rng = np.random.default_rng(7300 + seed)
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    scale = float(rng.uniform(0.95, 1.03))
    boxes, nw = [], 3
    for w in range(nw):
        for l, z in enumerate((z1, z2)):
            boxes.append((z["h"] * scale, lat.thermalise(z["xyz"], float(rng.uniform(0.02, 0.1)), 600 + 10 * w + l) * scale))
    mu_range = float(rng.choice([60.0, 150.0, 400.0]))
    weight0 = float(rng.uniform(0.0, 10.0)) * rng.random(101)
    samplerun, npt = bool(rng.random() < 0.3), bool(rng.random() < 0.5)
    _npt_against_oracle(so, c_oracle, boxes, 2, nw, temperature=float(rng.uniform(150.0, 300.0)), max_trans_ang=float(rng.uniform(0.3, 1.1)),
                        pressure_atm=1.0, vol_prob=float(rng.uniform(0.05, 0.2)) if npt else 0.0, dv_max_ang=float(rng.uniform(0.2, 0.9)),
                        seed=300 + seed, nmoves=80, mu_range=mu_range, weight0=weight0, samplerun=samplerun, always_switch=True,
                        wl_factor=0.0 if samplerun else float(rng.uniform(0.005, 0.3)), leshift=bool(rng.random() < 0.6),
                        dd=(int(rng.integers(2, 6)), int(rng.integers(1, 4)), int(rng.integers(0, 4))))


@pytest.mark.gpu
def test_chain_synchronisation_follows_the_oracle(so, c_oracle):
    """NPT sweeps with volume moves, then mc_check_chain_synchronisation: the farm (fractional reference positions kept
    on the host, bulk transfers) against mwo_chain_sync (ref_ljr carried through every volume move), twice in a row."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.energy import load_boxes
    from mc_water_ls_mw_amd.sweep import MuGrid, WalkerFarm
    from oracle import FullSweepState
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    grid = MuGrid(101, -400.0, 400.0)
    nw, p_au = 3, 1.0 / 2.90363081e8
    boxes = []
    for w in range(nw):
        boxes += [(z1["h"], lat.thermalise(z1["xyz"], 0.06, 460 + w)), (z2["h"], lat.thermalise(z2["xyz"], 0.06, 480 + w))]
    em = load_boxes([b[0] for b in boxes], [b[1] for b in boxes])
    farm = WalkerFarm(em, 2, 200.0, 1.1, grid=grid, weight=np.zeros(101), pressure_au=p_au)
    try:
        farm.options(record=True, samplerun=False, always_switch=True, npt=True, wl_factor=0.05)
        farm.moves(trans_prob=0.5, vol_prob=0.1, dv_max_ang=0.924)
        farm.set_reference()
        mus = [farm.initial_mu(w) for w in range(1, nw + 1)]
        for w in range(1, nw + 1):
            farm.set_state(w, 1, mus[w - 1])
        e0 = em.model_energy.copy()
        sts = []
        for w in range(nw):
            st = FullSweepState(c_oracle, [boxes[2 * w][0], boxes[2 * w + 1][0]], [boxes[2 * w][1], boxes[2 * w + 1][1]])
            st.model_energy[:] = e0[2 * w:2 * w + 2]
            st.ls_mu = mus[w]
            st.tabs = (np.zeros(101), np.zeros(101), np.zeros(101))
            sts.append(st)
        kw = dict(record=True, samplerun=False, always_switch=True, npt=True, wl_factor=0.05, pressure=p_au)
        for rnd in range(2):
            farm.sweep(96, seed=21, move0=96 * rnd)
            farm.chain_synchronise()
            hdev = farm.sync_cells()
            for w, st in enumerate(sts):
                so.full(st, 96, 21, w, 96 * rnd, farm.transP, farm.dv_max, farm.beta, farm.max_trans, grid, *st.tabs, **kw)
                so.chain_sync(st, farm.beta, p_au)
                s = farm.state(w + 1)
                assert np.abs(hdev[2 * w + 1] - st.h[1]).max() < 1e-10 and np.abs(hdev[2 * w] - st.h[0]).max() < 1e-11
                assert np.abs(farm.positions(2 * w + 2) - st.xyz[1]).max() < 1e-9
                assert np.abs(farm.positions(2 * w + 1) - st.xyz[0]).max() < 1e-9
                assert np.allclose(s["model_energy"], st.model_energy, rtol=0, atol=1e-9)
                assert abs(s["ls_mu"] - st.ls_mu) < 1e-6 * (1 + abs(st.ls_mu)) and s["ls"] == st.ls
    finally:
        em.energy_deinit()


@pytest.mark.gpu
def test_regauged_synchronisation_is_one_shared_table():
    """Many walkers per GPU: WalkerFarm.synchronise(regauge=True) must give what ONE table updated by all walkers
    would hold.  A Wang-Landau update adds wl_factor * av_binwidth / binwidth(k) to the weight and av_binwidth /
    binwidth(k) to the histogram of the same bin, so over any interval  delta weight = wl_factor * delta histogram
    up to the uniform gauge constant -- for the sum over walkers too.  The reference's own delta scheme (regauge=False)
    multiplies that constant by the number of walkers at every synchronisation instead (see the docstring)."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.comms import WalkerComms
    from mc_water_ls_mw_amd.energy import load_boxes
    from mc_water_ls_mw_amd.sweep import MuGrid, WalkerFarm
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    nw, f = 96, 0.05
    hs, xs = [], []
    for w in range(nw):
        hs += [z1["h"], z2["h"]]
        xs += [lat.thermalise(z1["xyz"], 0.05, 2 * w), lat.thermalise(z2["xyz"], 0.05, 2 * w + 1)]
    em = load_boxes(hs, xs)
    grid = MuGrid(101, -400.0, 400.0)
    start = 2.0 + 0.01 * np.abs(grid.mu_bin)                   # no bin at weight 0: every walker subtracts a minimum
    farm = WalkerFarm(em, 2, 200.0, 1.1, grid=grid, weight=start)
    comms = WalkerComms(101)
    comms.eta_last_sync[:] = start                             # the tables every walker starts from
    try:
        farm.options(record=True, samplerun=False, always_switch=True, wl_factor=f)
        for w in range(1, nw + 1):
            farm.set_state(w, 1, farm.initial_mu(w))
        prev_w, prev_h = start.copy(), np.zeros(101)
        for rnd in range(4):
            farm.sweep(48 * 5, seed=99, move0=rnd * 48 * 5)
            wt, hi, _ = farm.synchronise(comms, regauge=True)
            d = (wt - prev_w) - f * (hi - prev_h)
            assert hi.sum() > prev_h.sum()
            assert np.abs(d - d[0]).max() < 1e-9 * max(1.0, np.abs(wt).max())     # uniform: one shared table
            assert wt.min() == 0.0 and wt.max() < 1e4                               # in the reference's gauge, bounded
            for w in (1, nw):
                assert np.array_equal(farm.tables(w)[0], wt)
            prev_w, prev_h = wt.copy(), hi.copy()
    finally:
        em.energy_deinit()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_random_single_box_walkers_follow_the_oracle(seed, so):
    """Seeded random systems through every build of the driver: thin cells (a molecule neighbours its own image: the
    plain routine inside the sweep), compressed lattices (rows beyond 32 entries: the global list instead of LDS rows),
    dilute and sheared ones, sizes that do and do not fit the LDS-resident variants."""
    import test_gpu_fuzz as fz
    rng = np.random.default_rng(9100 + seed)
    for _ in range(50):
        kind, h, x = fz.random_system(rng)
        if kind != "gas" and len(x) >= 8:
            break
    temperature, max_trans = float(rng.uniform(150.0, 400.0)), float(rng.uniform(0.2, 1.1))
    nw = 3
    from mc_water_ls_mw_amd import lattice as lat
    boxes = [(h, lat.thermalise(x, 0.02, 500 + seed * 10 + w)) for w in range(nw)]
    from mc_water_ls_mw_amd.energy import load_boxes
    from mc_water_ls_mw_amd.sweep import WalkerFarm
    em = load_boxes([b[0] for b in boxes], [b[1] for b in boxes], maxneigh=64)
    farm = WalkerFarm(em, 1, temperature, max_trans)
    try:
        for w in range(1, nw + 1):
            farm.set_state(w, 1, 0.0)
        nmoves = 90
        log = farm.sweep(nmoves, seed=77 + seed, move0=3, log=True)
        for w in range(nw):
            ref = so.sweep(nmoves, 77 + seed, w, 3, [h], [boxes[w][1]], farm.beta, farm.max_trans, maxneigh=64)
            _compare(log[w], ref, farm.state(w + 1), [farm.positions(w + 1)])
    finally:
        em.energy_deinit()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["pair48_wl", "pair48_npt", "thin24", "ih64_npt", "ih288", "pair768_wl"])
@pytest.mark.parametrize("ahead", [1, 4])
def test_moment_path_of_the_driver_is_the_row_scan(case, ahead, monkeypatch):
    """Walkers entirely in LDS take the i--j--k sums of a translation from per-molecule moments (move_energy_mom_wave,
    moments_commit; MW_SWEEP_MOMENTS=0: the row scan of molint.F90:320-389 as written).  Same chain: every move's molecule and
    accept / reject / switch flags identical, its four local energies to 1e-10 relative, final positions to 1e-10 bohr -- on the
    reference's Ic / Ih pair (the 7.7 A Ih cell puts a neighbour in range through two images one trial move in twenty), with
    volume moves (the trial cell's moments), and on a cell 6 A wide in two directions (up to four images of a neighbour in range)."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.sweep import MuGrid

    big = case in ("ih288", "pair768_wl")       # walkers with their rows (and, 768, their positions) in global memory: translations only, the
                                                # engine's own moment array kept current by the driver; forced on here ("2": a few walkers)

    def run(moments):
        monkeypatch.setenv("MW_SWEEP_MOMENTS", "2" if (big and moments == "1") else moments)
        monkeypatch.setenv("MW_SWEEP_AHEAD", str(ahead))
        nlat = 2 if case.startswith("pair") else 1
        if nlat == 2:
            z1, z2 = load_golden("ic48"), load_golden("ih48")
            reps = (2, 2, 4) if big else (1, 1, 1)
            boxes = []
            for w in range(3):
                for l, z in enumerate((z1, z2)):
                    h, x = lat.replicate(z["h"], z["xyz"], reps)
                    boxes.append((h, lat.thermalise(x, 0.05, 40 + 20 * l + w)))
            mu = 6000.0 if big else 400.0
            em, farm = _farm(boxes, 2, 200.0, 1.1, grid=MuGrid(101, -mu, mu), weight=np.zeros(101))
            farm.options(record=True, samplerun=False, always_switch=True, npt=case.endswith("_npt"), wl_factor=0.05)
        else:
            if case == "thin24":
                h, x0 = lat.replicate(*lat.ice_ic_cell(2.63), (1, 3, 1))          # 6.07 x 18.2 x 6.07 A, 24 molecules
            elif case == "ih288":
                h, x0 = lat.ice_box("ih", (3, 4, 3), 0.0)
            else:
                h, x0 = lat.ice_box("ih", (2, 2, 2), 0.0)
            boxes = [(h, lat.thermalise(x0, 0.1, 300 + w)) for w in range(3)]
            em, farm = _farm(boxes, 1, 230.0, 1.1)
        try:
            if case.endswith("_npt"):
                if nlat == 1:
                    ct = __import__("ctypes")
                    em._chk(em.L.mw_sweep_options(0, 1, 0, 1, ct.c_double(1.0), ct.c_double(0.0), ct.c_double(0.0), ct.c_double(1.0 / 2.90363081e8)))
                farm.moves(trans_prob=0.5, vol_prob=0.05, dv_max_ang=0.3)
            nw = len(boxes) // nlat
            for w in range(1, nw + 1):
                farm.set_state(w, 1, farm.initial_mu(w) if nlat == 2 else 0.0)
            log = farm.sweep(600, seed=47, move0=2, log=True)
            last = em.sweep_last_launch() if hasattr(em, "sweep_last_launch") else None
            return log, [farm.positions(b) for b in range(1, len(boxes) + 1)], [farm.state(w) for w in range(1, nw + 1)], last
        finally:
            em.energy_deinit()

    got, ref = run("1"), run("0")
    assert 30 < (ref[0][0][:, 1].astype(int) & 1).sum() < 570                    # moves are accepted and rejected
    for w in range(len(ref[0])):
        assert np.array_equal(got[0][w][:, :2], ref[0][w][:, :2])                # molecule; accepted + 2 switched (+ 4 volume move)
        for c in (2, 3, 4, 5):
            assert np.all(np.abs(got[0][w][:, c] - ref[0][w][:, c]) <= RTOL * np.abs(ref[0][w][:, c]) + 1e-14), (w, c)
        assert got[2][w]["accepted"] == ref[2][w]["accepted"] and got[2][w]["ls"] == ref[2][w]["ls"]
    for a, b in zip(got[1], ref[1]):
        assert np.abs(a - b).max() < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ih1000", "pair1536_wl", "ih1000_npt", "ih64", "ih64_npt", "ih288", "ih288_npt", "pair48_wl", "pair48_wl_npt"])
def test_lookahead_is_the_sequential_chain(case, monkeypatch):
    """Look-ahead (several moves of a walker evaluated at once, decided in order; mw_sweep.hip.h) changes nothing: the move
    log, the final positions and the tables of a run with 2 or 4 moves in flight are BITWISE those of the one-move-at-a-time
    run -- on boxes small enough that consecutive moves do collide (1000 molecules: a move reads ~150; positions stay in global memory from 683 molecules up), with lattice
    switches and Wang-Landau updates in between, and with volume moves cutting the rounds short."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.sweep import MuGrid

    # (walkers in global memory take the moment path when the launch fills the chip: forced on for these few walkers -- the look-ahead's
    #  dependence test by distance is then what is held to the sequential chain; with volume moves it is off by construction)
    monkeypatch.setenv("MW_SWEEP_MOMENTS", "2")

    def run(ahead):
        monkeypatch.setenv("MW_SWEEP_AHEAD", str(ahead))
        if case.startswith("ih"):                       # one lattice: 1000 molecules (global memory), 288 (positions in LDS), 64 (rows too)
            reps = {"ih1000": (5, 5, 5), "ih64": (2, 2, 2), "ih288": (3, 4, 3)}[case.split("_")[0]]
            h, x0 = lat.ice_box("ih", reps, 0.0)
            boxes = [(h, lat.thermalise(x0, 0.12, 300 + w)) for w in range(3)]
            em, farm = _farm(boxes, 1, 230.0, 1.1)
            nlat = 1
        else:
            # two lattices: 1536 molecules (global memory), or the reference's 48-molecule pair (entirely in LDS: the moment path, and with
            # volume moves at four or six in flight the full-box energy spread over the workgroup's wavefronts -- its sum must be the
            # one-wavefront routine's to the bit; a thermalised pair reaches 10-13 in-range neighbours, i.e. the records' overflow too)
            small = case.startswith("pair48")
            z1, z2 = (load_golden("ic48"), load_golden("ih48")) if small else (load_golden("ic1536"), load_golden("ih1536"))
            grid = MuGrid(101, -400.0, 400.0) if small else MuGrid(101, -8000.0, 8000.0)
            boxes = []
            for w in range(2):
                boxes += [(z1["h"], lat.thermalise(z1["xyz"], 0.1, 7 + w)), (z2["h"], lat.thermalise(z2["xyz"], 0.1, 18 + w))]
            em, farm = _farm(boxes, 2, 200.0, 1.1, grid=grid)
            farm.options(record=True, samplerun=False, always_switch=True, npt=False, wl_factor=0.05)
            nlat = 2
        try:
            if case.endswith("_npt"):
                if nlat == 1:
                    em._chk(em.L.mw_sweep_options(0, 1, 0, 1, __import__("ctypes").c_double(1.0), __import__("ctypes").c_double(0.0),
                                                  __import__("ctypes").c_double(0.0), __import__("ctypes").c_double(1.0 / 2.90363081e8)))
                else:
                    farm.options(record=True, samplerun=False, always_switch=True, npt=True, wl_factor=0.05)
                farm.moves(trans_prob=0.5, vol_prob=0.05, dv_max_ang=0.3)
            nw = len(boxes) // nlat
            for w in range(1, nw + 1):
                farm.set_state(w, 1, farm.initial_mu(w))
            log = farm.sweep(400, seed=31, move0=5, log=True)
            pos = [farm.positions(b) for b in range(1, len(boxes) + 1)]
            st = [farm.state(w) for w in range(1, nw + 1)]
            tabs = [farm.tables(w) for w in range(1, nw + 1)] if nlat == 2 else []
            return log, pos, st, tabs
        finally:
            em.energy_deinit()

    ref = run(1)
    assert 20 < ref[0][0][:, 1].astype(int).__and__(1).sum() < 380         # moves are accepted and rejected
    for ahead in (2, 4) + ((8,) if case.startswith("ih1000") else ()) + ((6,) if case.startswith("pair48") else ()):   # (eight in flight: one-lattice walkers in global memory; six: two-lattice walkers in LDS)
        got = run(ahead)
        assert np.array_equal(got[0], ref[0])
        for a, b in zip(got[1], ref[1]):
            assert np.array_equal(a, b)
        assert got[2] == ref[2]
        for ta, tb in zip(got[3], ref[3]):
            for a, b in zip(ta, tb):
                assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_lookahead_with_random_run_options(seed, monkeypatch):
    """Look-ahead under the run options of mc_cycle in random combination: weights of any size, walkers inside or outside the
    order-parameter range, weight generation or sample run, switch attempts, leshift, MINU, NVT or volume moves.  On a 768-molecule
    Ic / Ih pair (walkers in global memory, moves that do collide) and, odd seeds, on the 48-molecule pair itself and its 192-molecule replica (walkers
    entirely / with their positions in LDS, where a move reads most of the box: any accepted move ends the round).  2 and 4 moves in flight = the one-at-a-time
    chain, bit for bit."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.sweep import MuGrid
    rng = np.random.default_rng(8800 + seed)
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    scale = float(rng.uniform(0.9, 1.04))
    small = seed % 2 == 1
    reps = ((1, 1, 1) if seed % 4 == 1 else (1, 2, 2)) if small else (2, 2, 4)      # 48: rows in LDS too; 192: positions only
    boxes, nw = [], 2
    for w in range(nw):
        for l, z in enumerate((z1, z2)):
            h, x = lat.replicate(z["h"], z["xyz"], reps)
            boxes.append((h * scale, lat.thermalise(x, float(rng.uniform(0.03, 0.12)), 800 + 10 * w + l) * scale))
    assert small or len(boxes[0][1]) * 24 * 2 > 16 * 1024         # (768 molecules: positions stay in global memory)
    mu_range = float(rng.choice([300.0, 3000.0, 20000.0]))
    grid = MuGrid(101, -mu_range, mu_range)
    amp = float(10.0 ** rng.uniform(0.0, 8.0)) if rng.random() < 0.4 else float(rng.uniform(0.0, 20.0))
    weight0 = amp * (np.exp(-(grid.mu_bin / (0.3 * mu_range)) ** 2) + 0.3 * rng.random(101))
    samplerun, npt = bool(rng.random() < 0.4), bool(rng.random() < 0.5)
    opts = dict(record=True, samplerun=samplerun, always_switch=bool(rng.random() < 0.8), npt=npt,
                wl_factor=0.0 if samplerun else float(rng.uniform(0.001, 0.5)), log_unbiased_norm=float(rng.uniform(0.0, 5.0)) if samplerun else 0.0)
    temperature, max_trans = float(rng.uniform(150.0, 300.0)), float(rng.uniform(0.3, 1.1))
    p_au = float(10.0 ** rng.uniform(0.0, 3.0)) / 2.90363081e8
    leshift, minu = bool(rng.random() < 0.4), bool(rng.random() < 0.3)
    vol_prob, dv = float(rng.uniform(0.02, 0.15)), float(rng.uniform(0.1, 0.6))
    eta_interp = bool(rng.random() < 0.8)

    def run(ahead):
        from mc_water_ls_mw_amd.energy import load_boxes
        from mc_water_ls_mw_amd.sweep import WalkerFarm
        monkeypatch.setenv("MW_SWEEP_AHEAD", str(ahead))
        em = load_boxes([b[0] for b in boxes], [b[1] for b in boxes])
        farm = WalkerFarm(em, 2, temperature, max_trans, grid=grid, weight=weight0.copy(), eta_interp=eta_interp, pressure_au=p_au)
        try:
            farm.options(**opts)
            if leshift:
                farm.leshift(farm.starting_enthalpy(1, npt))
            if minu:
                farm.minu(True)
            if npt:
                farm.moves(trans_prob=0.5, vol_prob=vol_prob, dv_max_ang=dv)
            for w in range(1, nw + 1):
                farm.set_state(w, 1 + (w + seed) % 2, farm.initial_mu(w))
            log = farm.sweep(240, seed=70 + seed, move0=3, log=True)
            return (log, [farm.positions(b) for b in range(1, 2 * nw + 1)], [farm.state(w) for w in range(1, nw + 1)],
                    [farm.tables(w) for w in range(1, nw + 1)], [farm.volume_moves(w) for w in range(1, nw + 1)])
        finally:
            em.energy_deinit()

    ref = run(1)
    for ahead in (2, 4) + ((6,) if seed % 4 == 1 else ()):         # (six in flight: the walkers entirely in LDS)
        got = run(ahead)
        assert np.array_equal(got[0], ref[0], equal_nan=True)
        assert all(np.array_equal(a, b) for a, b in zip(got[1], ref[1])) and got[2] == ref[2] and got[4] == ref[4]
        for ta, tb in zip(got[3], ref[3]):
            assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ta, tb))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense", "thin_sheared", "ice_sheared"])
def test_lookahead_on_awkward_boxes(kind, so, monkeypatch):
    """Look-ahead where the fused routine declines and the plain one steps in (compressed lattices: rows beyond 32 entries,
    more than 24 in-range neighbours -- the plain routine keeps no account of what it read, so such a move depends on every
    earlier move of its round), in cells thin enough that a molecule neighbours its own image, and in sheared cells: 4 moves
    in flight give the one-at-a-time chain bit for bit, and that chain is the oracle's."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.energy import load_boxes
    from mc_water_ls_mw_amd.sweep import WalkerFarm
    rng = np.random.default_rng({"dense": 5, "thin_sheared": 6, "ice_sheared": 7}[kind])
    if kind == "dense":
        h, x = lat.ice_ih_cell(2.3)
        h, x = lat.replicate(h, x, (5, 5, 4))                    # 800 molecules, ~40 list neighbours each
    elif kind == "thin_sheared":
        h, x = lat.ice_ic_cell(2.75)
        h, x = lat.replicate(h, x, (1, 10, 9))                   # one cell thick along a: images of a molecule among its neighbours
    else:
        h, x = lat.ice_ic_cell(2.8)
        h, x = lat.replicate(h, x, (5, 5, 4))                    # 800 molecules
    if kind != "dense":
        shear = np.eye(3) + rng.uniform(-0.05, 0.05, (3, 3)) * (1 - np.eye(3))
        h, x = h @ shear, x @ shear
    assert len(x) * 24 > 16 * 1024                               # positions stay in global memory: the look-ahead builds
    boxes = [(np.ascontiguousarray(h), lat.thermalise(np.ascontiguousarray(x), 0.05, 40 + w)) for w in range(2)]
    nmoves = 150

    def run(ahead):
        monkeypatch.setenv("MW_SWEEP_AHEAD", str(ahead))
        em = load_boxes([b[0] for b in boxes], [b[1] for b in boxes], maxneigh=64)
        farm = WalkerFarm(em, 1, 260.0, 0.9)
        try:
            farm.set_states(1, np.zeros(2))
            log = farm.sweep(nmoves, seed=12, move0=0, log=True)
            return log, [farm.positions(b) for b in (1, 2)], [farm.state(w) for w in (1, 2)], farm
        finally:
            em.energy_deinit()

    log1, pos1, st1, farm = run(1)
    log4, pos4, st4, _ = run(4)
    assert np.array_equal(log4, log1) and all(np.array_equal(a, b) for a, b in zip(pos4, pos1)) and st4 == st1
    assert 5 < int(log1[0][:, 1].sum()) < nmoves - 5
    for w in range(2):
        ref = so.sweep(nmoves, 12, w, 0, [boxes[w][0]], [boxes[w][1]], farm.beta, farm.max_trans, maxneigh=64)
        _compare(log1[w], ref, st1[w], [pos1[w]])


@pytest.mark.gpu
@pytest.mark.parametrize("nlat", [1, 2])
@pytest.mark.parametrize("residency", [0, 1, 2])
@pytest.mark.parametrize("npt", [False, True])
def test_every_build_of_the_driver_runs_the_same_chain(nlat, residency, npt, monkeypatch):
    """The 40 instantiations of k_sweep -- lattices x where a walker's data live x with / without volume moves x 1, 2 or 4 moves in
    flight, 8 for one-lattice walkers in global memory, 6 for two-lattice walkers entirely in LDS -- each actually launched (mw_sweep_last_launch says which build a launch
    took) and, for the same walkers, 2, 4 (and 8) moves in flight reproduce the one-move-at-a-time chain bit for bit."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.energy import load_boxes
    from mc_water_ls_mw_amd.sweep import MuGrid, WalkerFarm
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    reps = {0: (2, 2, 4), 1: (1, 2, 2), 2: (1, 1, 1)}[residency]          # 768 / 192 / 48 molecules per lattice
    boxes, nw = [], 2
    for w in range(nw):
        for l, z in enumerate((z1, z2)[:nlat]):
            h, x = lat.replicate(z["h"], z["xyz"], reps)
            boxes.append((h, lat.thermalise(x, 0.08, 40 + 10 * w + l)))
    grid = MuGrid(101, -3000.0, 3000.0) if nlat == 2 else None

    def run(ahead):
        monkeypatch.setenv("MW_SWEEP_AHEAD", str(ahead))
        em = load_boxes([b[0] for b in boxes], [b[1] for b in boxes])
        farm = WalkerFarm(em, nlat, 220.0, 1.0, grid=grid, pressure_au=1.0 / 2.90363081e8)
        try:
            if nlat == 2:
                farm.options(record=True, samplerun=False, always_switch=True, npt=npt, wl_factor=0.05)
            elif npt:
                em._chk(em.L.mw_sweep_options(0, 1, 0, 1, __import__("ctypes").c_double(1.0), __import__("ctypes").c_double(0.0),
                                              __import__("ctypes").c_double(0.0), __import__("ctypes").c_double(1.0 / 2.90363081e8)))
            if npt:
                farm.moves(trans_prob=0.5, vol_prob=0.08, dv_max_ang=0.4)
            for w in range(1, nw + 1):
                farm.set_state(w, 1, farm.initial_mu(w))
            log = farm.sweep(200, seed=5, move0=0, log=True)
            what = farm.last_launch()
            assert (what["nlat"], what["ahead"], what["residency"], what["volume_moves"]) == (nlat, ahead, residency, int(npt)), what
            assert (what["row_stride"] > 0) == (residency == 2)
            return log, [farm.positions(b) for b in range(1, len(boxes) + 1)], [farm.state(w) for w in range(1, nw + 1)]
        finally:
            em.energy_deinit()

    ref = run(1)
    assert 5 < int((ref[0][0][:, 1].astype(int) & 1).sum()) < 195
    for ahead in (2, 4) + ((8,) if nlat == 1 and residency == 0 else ()) + ((6,) if nlat == 2 and residency == 2 else ()):
        got = run(ahead)
        assert np.array_equal(got[0], ref[0], equal_nan=True)
        assert all(np.array_equal(a, b) for a, b in zip(got[1], ref[1])) and got[2] == ref[2]
