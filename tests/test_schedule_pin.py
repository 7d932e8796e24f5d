"""CPU: pin the Wang-Landau schedule layer -- oracle/schedule.py (mc_check_flatness, the 1/t clamp, log_unbiased_norm,
mc_compute_deltaG_from_hist) -- against the REFERENCE PROGRAM (oracle/_ref/mc_water_ref_rng, build container only), then
hold the product's mc_water_ls_mw_amd.schedule to the pinned oracle on synthetic tables (runs everywhere).

The reference runs examples/ice1_gen_weights in miniature (48-molecule Ic/Ih pair, Wang-Landau updates and a switch
attempt after every move) with flat_chk_int = 8; its final checkpoint holds wl_factor, histogram, weights and
wl_invt_active, wlf.dat the history of the increment, mc.log the delta G estimate of a sample run."""
import os
import re
import struct

import numpy as np
import pytest

from conftest import ROOT

import test_sweep_pin as pin

needs_ref = pytest.mark.skipif(not os.path.exists(pin.RNG), reason="oracle/_ref/mc_water_ref_rng not built")


@pytest.fixture(scope="module")
def so():
    from oracle import SweepOracle
    return SweepOracle()


def replay_with_schedule(so, c_oracle, boxes, cycles, grid, st, chk_int, samplerun=False, weight=None, lun=0.0,
                         temperature=200.0):
    from mc_water_ls_mw_amd.lattice import ANG_TO_BOHR
    from mc_water_ls_mw_amd.sweep import KB
    from oracle import schedule as osch
    beta = 1.0 / (KB * temperature)
    hs = [b[0] for b in boxes]
    xs = [np.array(b[1]) for b in boxes]
    ivs = [c_oracle.ivects(h) for h in hs]
    lists = [c_oracle.neighbours(xs[l], ivs[l]) for l in range(2)]
    me = [c_oracle.model_energy(xs[l], ivs[l], *lists[l]) for l in range(2)]
    p = 1.0 / pin.AUP_TO_ATM
    v = [abs(np.linalg.det(h)) for h in hs]
    mu = (me[0] + p * v[0] - me[1] - p * v[1]) * beta - 48.0 * np.log(v[0] / v[1])
    ls, hi, uh = 1, np.zeros(grid.nbins), np.zeros(grid.nbins)
    w = np.zeros(grid.nbins) if weight is None else np.array(weight, dtype=float)
    events = []
    for cyc in range(1, cycles + 1):
        if cyc % 10 == 0:
            lists = [c_oracle.neighbours(xs[l], ivs[l]) for l in range(2)]
        f = 0.0 if samplerun else osch.cycle_factor(st, cyc, 48, grid.nbins)
        r = so.cycle(48, pin.SEED, 0, (cyc - 1) * 48, hs, xs, beta, 1.1 * ANG_TO_BOHR, grid, w, hi, uh, ls=ls, ls_mu=mu,
                     model_energy=me, lists=lists, record=True, samplerun=samplerun, always_switch=True, npt=False,
                     wl_factor=f, log_unbiased_norm=lun, pressure=p)
        xs = [r["xyz"][0], r["xyz"][1]]
        me, ls, mu, w, hi, uh = list(r["model_energy"]), r["ls"], r["ls_mu"], r["weight"], r["histogram"], r["unbiased_hist"]
        if not samplerun and cyc % chk_int == 0:
            events.append((cyc, osch.flatness_step(st, cyc, 48, hi, w)))
    return np.array(xs), w, hi, uh, events


def checkpoint_tables(samplerun=False):
    recs = pin.run_reference.records
    out = dict(wl_factor=struct.unpack("<d", recs[3])[0], hist=np.frombuffer(recs[4], dtype="<f8").copy(),
               weight=np.frombuffer(recs[5], dtype="<f8").copy(), invt=bool(struct.unpack("<i", recs[6])[0]))
    if samplerun:
        out["uhist"] = np.frombuffer(recs[7], dtype="<f8").copy()
    return out


F0 = float(np.float32(0.05))     # userparams.f90:32: a single-precision literal


@needs_ref
def test_first_reset_then_halving_matches_the_reference_program(tmp_path, so, c_oracle):
    """wl_schedule = 1 ("every bin visited wl_minhist times") with wl_minhist = -1: the first check resets the
    histogram (firstcycle, minimum 0 > -1), every later one finds it 'flat': weights shifted, wlf.dat, histogram
    reset, increment halved -- four times in 40 cycles."""
    from mc_water_ls_mw_amd.sweep import MuGrid
    from mc_water_ls_mw_amd import io as mwio
    from oracle import schedule as osch
    grid = MuGrid(101, -400.0, 400.0)
    d = str(tmp_path / "run")
    boxes, e_ref, ljr, ls, hist, wgt = pin.run_reference(d, 2, 200, 40, samplerun=False, always_switch=True, tables=True,
                                                         mc_extra="wl_schedule = 1\nwl_minhist = -1", book_extra="flat_chk_int = 8")
    ref = checkpoint_tables()
    st = osch.new_state(F0, schedule=1, minhist=-1)
    xs, w, hi, uh, events = replay_with_schedule(so, c_oracle, boxes, 40, grid, st, 8)
    assert [e[1] for e in events] == ["first reset", "halved", "halved", "halved", "halved"]
    assert ref["wl_factor"] == st["wl_factor"] == F0 / 16 and not ref["invt"]
    assert np.abs(xs - ljr).max() < 1e-10
    assert np.allclose(hi, ref["hist"], rtol=0, atol=1e-12) and ref["hist"].sum() == 0.0     # reset at cycle 40
    assert np.allclose(w, ref["weight"], rtol=1e-11, atol=1e-12) and ref["weight"].max() > 0
    wlf = mwio.read_wlf(d)
    assert [c for c, _ in wlf] == [c for c, _ in st["wlf"]] == [16, 16, 24, 24, 32, 32, 40, 40]
    assert np.allclose([f for _, f in wlf], [f for _, f in st["wlf"]], rtol=1e-11)
    # the tagged tables of the last flat histogram: weights after the shift
    tag = "%.12f" % (F0 / 8)
    _, mu_t, w_t = mwio.read_table(os.path.join(d, "eta_weights.dat_" + tag))
    assert np.allclose(w_t, w, rtol=1e-11, atol=1e-12) and np.allclose(mu_t, grid.mu_bin)


@needs_ref
def test_switch_to_one_over_t_matches_the_reference_program(tmp_path, so, c_oracle):
    """wl_schedule = 0 with a huge tolerance (flat at the first check) and wl_useinvt: the halved increment is below
    nbins / (cycle nwater), so the run switches to the 1/t rule, which then shrinks the increment every cycle."""
    from mc_water_ls_mw_amd.sweep import MuGrid
    from oracle import schedule as osch
    grid = MuGrid(101, -400.0, 400.0)
    boxes, e_ref, ljr, ls, hist, wgt = pin.run_reference(str(tmp_path / "run"), 2, 200, 40, samplerun=False, always_switch=True,
                                                         tables=True, mc_extra="wl_flattol = 1.0e9\nwl_useinvt = .true.",
                                                         book_extra="flat_chk_int = 8")
    ref = checkpoint_tables()
    st = osch.new_state(F0, schedule=0, flattol=1.0e9, useinvt=True)
    xs, w, hi, uh, events = replay_with_schedule(so, c_oracle, boxes, 40, grid, st, 8)
    assert [e[1] for e in events] == ["halved", "invt", "invt", "invt", "invt"]
    assert ref["invt"] and st["invt_active"]
    assert ref["wl_factor"] == pytest.approx(101.0 / (40 * 48), rel=1e-14) and st["wl_factor"] == ref["wl_factor"]
    assert np.abs(xs - ljr).max() < 1e-10
    assert np.allclose(hi, ref["hist"], rtol=1e-12, atol=1e-12) and np.allclose(w, ref["weight"], rtol=1e-11, atol=1e-12)


@needs_ref
def test_histogram_that_is_not_flat_leaves_everything_alone(tmp_path, so, c_oracle):
    """The shipped settings (wl_schedule 0, wl_flattol 0.05; schedule 2 likewise): 1920 visits over 101 bins are nowhere
    near flat, so the checks change nothing."""
    from mc_water_ls_mw_amd.sweep import MuGrid
    from oracle import schedule as osch
    grid = MuGrid(101, -400.0, 400.0)
    for sched in (0, 2):
        boxes, e_ref, ljr, ls, hist, wgt = pin.run_reference(str(tmp_path / f"run{sched}"), 2, 200, 24, samplerun=False,
                                                             always_switch=True, tables=True, mc_extra=f"wl_schedule = {sched}",
                                                             book_extra="flat_chk_int = 8")
        ref = checkpoint_tables()
        st = osch.new_state(F0, schedule=sched)
        xs, w, hi, uh, events = replay_with_schedule(so, c_oracle, boxes, 24, grid, st, 8)
        assert [e[1] for e in events] == ["checked"] * 3 and ref["wl_factor"] == F0 == st["wl_factor"]
        assert np.allclose(hi, ref["hist"], rtol=1e-12, atol=1e-12) and np.allclose(w, ref["weight"], rtol=1e-11, atol=1e-12)


@needs_ref
def test_unbiased_histogram_and_delta_g_match_the_reference_program(tmp_path, so, c_oracle):
    """examples/ice1_sample in miniature with deltaG_int = 60: the unbiased histogram (normalised through
    log_unbiased_norm) from the checkpoint and the delta G lines of mc.log."""
    from mc_water_ls_mw_amd.sweep import KB, MuGrid
    from oracle import schedule as osch
    grid = MuGrid(101, -400.0, 400.0)
    weight = 0.5 * np.abs(grid.mu_bin)             # steep enough to walk from mu = -330 across 0 within 60 cycles
    d = str(tmp_path / "run")
    boxes, e_ref, ljr, ls, hist, wgt = pin.run_reference(d, 2, 200, 60, weight=weight, grid=grid, samplerun=True,
                                                         always_switch=True, tables=True, book_extra="deltaG_int = 60",
                                                         run_env=dict(MW_WRAP_SIZE="1"))     # G11: serial comms has size = 0
    ref = checkpoint_tables(samplerun=True)
    lun = osch.unbiased_norm(weight, grid.av_binwidth, 60, 1, 1, 48)
    xs, w, hi, uh, _ = replay_with_schedule(so, c_oracle, boxes, 60, grid, None, 10 ** 9, samplerun=True, weight=weight, lun=lun)
    assert np.abs(xs - ljr).max() < 1e-10
    assert ref["uhist"][:50].sum() > 0 and ref["uhist"][51:].sum() > 0           # both lattices' halves were visited
    assert np.allclose(uh, ref["uhist"], rtol=1e-9, atol=0)
    log = open(os.path.join(d, "mc.log")).read()
    m = re.search(r"G\(lattice2\) - G\(lattice1\) =\s*([-+0-9.Ee]+)\s*kT/molecule", log)
    assert m, log[-1500:]
    dg = osch.delta_g(uh, grid.binwidth)
    assert float(m.group(1)) == pytest.approx(dg / 48.0, abs=2e-8)          # F15.8 in the log
    mj = re.search(r"G\(lattice2\) - G\(lattice1\) =\s*([-+0-9.Ee]+)\s*meV/molecule", log)
    assert float(mj.group(1)) == pytest.approx(KB * 200.0 * pin.HART_TO_EV * 1000.0 * dg / 48.0, abs=2e-8)


# ---- the product layer against the pinned oracle (no reference needed) --------------------------------------------
class TableFarm:
    """What WangLandauSchedule needs from a WalkerFarm, on host arrays: `nw` walkers with their own tables."""

    def __init__(self, nbins, nw, rng):
        from mc_water_ls_mw_amd.comms import WalkerComms
        from mc_water_ls_mw_amd.sweep import MuGrid
        self.grid = MuGrid(nbins, -400.0, 400.0)
        self.nw = nw
        self.w = rng.random((nw, nbins)) * 3.0
        self.h = np.zeros((nw, nbins))
        self.c = WalkerComms(nbins)

    def allreduce_hist(self, comms=None):
        last = self.c.hist_last_sync
        total = last + (self.h - last[None, :]).sum(axis=0)
        self.c.allreduce_hist(total)
        self.h[:] = total
        return total

    def reset_histogram(self, comms=None):
        self.h[:] = 0.0
        self.c.set_histogram(np.zeros(self.h.shape[1]))

    def shift_weights(self):
        self.w -= self.w[:, self.w.shape[1] // 2][:, None]

    def tables(self, walker):
        return self.w[walker - 1], self.h[walker - 1], None


@pytest.mark.parametrize("sched,flattol,minhist,useinvt", [(0, 0.3, 20, False), (1, 0.05, 3, False), (2, 0.3, 20, True),
                                                             (0, 0.05, 2, True)])
def test_product_schedule_follows_the_pinned_oracle(tmp_path, sched, flattol, minhist, useinvt):
    """One walker (= one rank of the reference): random visit increments between checks; the product's decisions,
    increment, weights, histogram and wlf.dat must equal the oracle's at every check."""
    from mc_water_ls_mw_amd.schedule import WangLandauSchedule
    from mc_water_ls_mw_amd import io as mwio
    from oracle import schedule as osch
    rng = np.random.default_rng(11 + sched)
    nbins, nwater = 21, 48
    farm = TableFarm(nbins, 1, rng)
    ws = WangLandauSchedule(nbins, 0.05, wl_schedule=sched, wl_flattol=flattol, wl_minhist=minhist, wl_useinvt=useinvt,
                            outdir=str(tmp_path))
    st = osch.new_state(0.05, schedule=sched, flattol=flattol, minhist=minhist, useinvt=useinvt)
    ow, oh = farm.w[0].copy(), np.zeros(nbins)
    seen = set()
    for cyc in range(1, 401):
        f1, f2 = ws.move_factor(cyc, nwater), osch.cycle_factor(st, cyc, nwater, nbins)
        assert f1 == f2
        k = rng.integers(0, nbins, size=nwater)
        inc = np.bincount(k, minlength=nbins).astype(float)
        farm.h[0] += inc; oh += inc
        farm.w[0] += f1 * inc; ow += f2 * inc
        if cyc % 10 == 0:
            got = ws.check_flatness(cyc, nwater, farm)
            want = osch.flatness_step(st, cyc, nwater, oh, ow)
            seen.add(want)
            assert got["action"] == want, (cyc, got, want)
            assert ws.wl_factor == st["wl_factor"] and ws.invt_active == st["invt_active"]
            assert np.array_equal(farm.h[0], oh) and np.array_equal(farm.w[0], ow)
    assert "halved" in seen and (useinvt or "invt" not in seen)
    if sched == 2:
        assert "invt" in seen                        # this parameter set reaches the 1/t regime
    if st["wlf"]:
        wlf = mwio.read_wlf(str(tmp_path))
        assert [c for c, _ in wlf] == [c for c, _ in st["wlf"]]
        assert np.allclose([f for _, f in wlf], [f for _, f in st["wlf"]], rtol=1e-11)


def test_product_schedule_sums_the_walkers_of_a_farm():
    """Four walkers with their own histograms: the check sees the sum (comms_allreduce_hist), every walker is reset and
    shifted, and the baseline of the delta scheme is re-based (comms_set_histogram)."""
    from mc_water_ls_mw_amd.schedule import WangLandauSchedule
    rng = np.random.default_rng(5)
    farm = TableFarm(11, 4, rng)
    ws = WangLandauSchedule(11, 0.05, wl_schedule=1, wl_minhist=8)
    farm.h[:] = 3.0                                  # 12 visits per bin over the four walkers: more than wl_minhist
    assert ws.check_flatness(10, 48, farm)["action"] == "first reset" and farm.h.sum() == 0
    farm.h[:] = 1.0                                  # 4 per bin in total: not enough
    assert ws.check_flatness(20, 48, farm)["action"] == "checked" and np.all(farm.h == 4.0)
    farm.h += 1.5                                    # each walker adds 1.5: 4 + 6 = 10 per bin
    w_before = farm.w.copy()
    out = ws.check_flatness(30, 48, farm)
    assert out["action"] == "halved" and ws.wl_factor == 0.025 and not ws.firstcycle
    assert farm.h.sum() == 0 and np.all(farm.c.hist_last_sync == 0)
    assert np.allclose(farm.w, w_before - w_before[:, 5][:, None]) and np.all(farm.w[:, 5] == 0)


def test_log_unbiased_norm_and_delta_g_follow_the_pinned_oracle():
    from mc_water_ls_mw_amd import schedule as ps
    from mc_water_ls_mw_amd.sweep import MuGrid
    from oracle import schedule as osch
    rng = np.random.default_rng(3)
    grid = MuGrid(101, -400.0, 400.0)
    for scale in (0.02, 5.0, 300.0):                 # the last one would overflow a plain sum of exp(weight)
        w = scale * np.abs(grid.mu_bin) / 100.0
        a = ps.log_unbiased_norm(w, grid.av_binwidth, 500000, 1000, 8, 48)
        assert a == osch.unbiased_norm(w, grid.av_binwidth, 500000, 1000, 8, 48) and np.isfinite(a)
    uh = rng.random(101) * np.exp(-((grid.mu_bin - 30.0) / 90.0) ** 2)
    dg, per, normp = ps.delta_g_from_hist(uh, grid.binwidth, 48, 200.0)
    assert dg == pytest.approx(osch.delta_g(uh, grid.binwidth), rel=1e-13)
    assert per["kT"] == dg / 48 and np.isclose((normp * grid.binwidth).sum(), 1.0)


def test_fortran_e_format():
    from mc_water_ls_mw_amd.io import fortran_e
    assert fortran_e(0.05) == "  0.500000000000E-01" and fortran_e(1234.5) == "  0.123450000000E+04"
    assert fortran_e(-2.5e-7) == " -0.250000000000E-06" and float(fortran_e(0.9999999999999)) == pytest.approx(1.0)
