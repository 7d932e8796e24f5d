"""CPU: pin the oracle of the translation-move driver (oracle/mw_oracle.c, mwo_sweep_translation,
mwo_eta_weight, mwo_mu_to_bin, mwo_mu_grid, mwo_recipmatrix) against the REFERENCE PROGRAM itself.

oracle/_ref/mc_water_ref_rng (build container only) is the reference's unmodified main/mc_moves/io/...
with two link-time interpositions: the G2 stack scrub and a random_uniform_random that returns the
oracle's counter-based Philox stream (oracle/ref_wrap_rng.c).  With volume and switch moves off, its
mc_cycle is then exactly "nwater translation moves, lists rebuilt every list_update_int cycles"; the
oracle replays that and must end at the same positions (read from the reference's checkpoint file, full
double precision) having passed through the same energies (thermo file, 6 digits) -- single box, and the
two-lattice Ic/Ih system with interpolated multicanonical weights (eta_weights.dat)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

RNG = os.path.join(ROOT, "oracle", "_ref", "mc_water_ref_rng")
SEED = 424242          # oracle/ref_wrap_rng.c
HART_TO_EV = 27.211396181
AUP_TO_ATM = 2.90363081e8

pytestmark = pytest.mark.skipif(not os.path.exists(RNG), reason="oracle/_ref/mc_water_ref_rng not built")


def namelist(num_lattices, temperature, cycles, samplerun, always_switch=False, npt=False, vol_prob=None, latt_sync=None,
             mc_extra="", book_extra="", par_extra=""):
    volume_lines = "allow_vol        = .false." if not npt else f"mc_vol_prob      = {vol_prob}"
    return f"""&potential
model_type = "mW"
/
&thermal
temperature = {temperature}
pressure    = 1.0
/
&MonteCarlo
mc_ensemble      = '{'npt' if npt else 'nvt'}'
mc_max_trans     = 1.1
mc_dv_max        = 0.924
nbins            = 101
mu_max           = +400
mu_min           = -400
mc_always_switch = .{'true' if always_switch else 'false'}.
allow_switch     = .false.
{volume_lines}
eta_interp       = .true.
samplerun        = .{'true' if samplerun else 'false'}.
{mc_extra}
/
&config
num_lattices = {num_lattices}
nwater       = 48
method       = 'xmol'
ls           = 1
/
&bookkeeping
list_update_int  = 10
traj_output_int  = 100000
file_output_int  = 1
max_mc_cycles    = {cycles}
eq_mc_cycles     = 1
eq_adjust_mc     = .false.
chkpt_dump_int   = {cycles}
{'' if latt_sync is None else f'latt_sync_int    = {latt_sync}'}
{book_extra}
timer_qtime      = 172800
timer_closetime  = 1800
/
{'' if not par_extra else '&parallelisation' + chr(10) + par_extra + chr(10) + '/'}
"""


def read_records(path):
    recs, data = [], open(path, "rb").read()
    off = 0
    while off < len(data):
        (n,) = struct.unpack_from("<i", data, off)
        recs.append(data[off + 4:off + 4 + n])
        off += 8 + n
    return recs


def run_reference(d, num_lattices, temperature, cycles, weight=None, grid=None, samplerun=None, always_switch=False,
                  tables=False, npt=False, vol_prob=None, transP=1.0, latt_sync=None, mc_extra="", book_extra="", run_env={},
                  par_extra="", program=None):
    from mc_water_ls_mw_amd import lattice as lat
    os.makedirs(d)
    samplerun = (weight is not None) if samplerun is None else samplerun
    open(os.path.join(d, "ice.input"), "w").write(namelist(num_lattices, temperature, cycles, samplerun, always_switch, npt, vol_prob, latt_sync, mc_extra, book_extra, par_extra))
    z1 = load_golden("ic48_t015")
    h1, x1 = lat.read_xmol(_write(d, "input001.xmol", z1))
    boxes = [(h1, x1)]
    if num_lattices == 2:
        z2 = load_golden("ih48_t020")
        boxes.append(lat.read_xmol(_write(d, "input002.xmol", z2)))
    if weight is not None:
        with open(os.path.join(d, "eta_weights.dat"), "w") as fh:       # format of mc_moves.F90:738-770
            fh.write("#Current energy increment =   0.500000007451E-01\n")
            for mu, w in zip(grid.mu_bin, weight):
                fh.write(f"  {float(mu)!r}        {float(w)!r}\n")
    env = dict(os.environ, MW_WRAP_SWITCH="1" if always_switch else "0", MW_WRAP_TRANSP=repr(float(transP)), **run_env)
    out = subprocess.run([program or RNG, "ice.input"], cwd=d, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-800:], out.stderr[-800:])
    therm = [f for f in os.listdir(d) if f.endswith("_therm.dat")][0]
    e_ev = np.array([float(ln.split()[1]) for ln in open(os.path.join(d, therm))])
    chk = [f for f in os.listdir(d) if f.startswith("checkpoint")]
    recs = read_records(os.path.join(d, sorted(chk)[-1]))
    ljr = np.frombuffer(recs[-2], dtype="<f8").reshape(num_lattices, 48, 3)      # ljr(3,1,N,L), column-major
    run_reference.hmatrix = np.frombuffer(recs[-4], dtype="<f8").reshape(num_lattices, 3, 3).copy()   # hmatrix(3,3,L)
    run_reference.ref_ljr = np.frombuffer(recs[-3], dtype="<f8").reshape(num_lattices, 48, 3).copy()
    ls = struct.unpack("<i", recs[-1])[0]
    assert struct.unpack("<i", recs[1])[0] == cycles
    run_reference.records = recs
    if tables:      # records: nwater, cycle, (max_trans, dv_max), wl_factor, histogram, weight, wl_invt_active, [uhist], ...
        hist = np.frombuffer(recs[4], dtype="<f8").copy()
        wgt = np.frombuffer(recs[5], dtype="<f8").copy()
        return boxes, e_ev, np.array(ljr), ls, hist, wgt
    return boxes, e_ev, np.array(ljr), ls


def _write(d, name, z):
    from mc_water_ls_mw_amd import lattice as lat
    p = os.path.join(d, name)
    lat.write_xmol(p, z["h"], z["xyz"])
    return p


def replay(so, c_oracle, boxes, temperature, cycles, grid=None, weight=None):
    """mc_cycle with translations only: lists rebuilt when mod(cycle, 10) == 0, then nwater moves."""
    from mc_water_ls_mw_amd.sweep import KB
    from mc_water_ls_mw_amd.lattice import ANG_TO_BOHR
    beta = 1.0 / (KB * temperature)
    hs = [b[0] for b in boxes]
    xs = [np.array(b[1]) for b in boxes]
    ivs = [c_oracle.ivects(h) for h in hs]
    lists = [c_oracle.neighbours(xs[l], ivs[l]) for l in range(len(xs))]
    me = [c_oracle.model_energy(xs[l], ivs[l], *lists[l]) for l in range(len(xs))]
    ls, mu = 1, 0.0
    if len(xs) == 2:                                            # mc_moves.F90:856-859
        p = 1.0 / AUP_TO_ATM
        v = [abs(np.linalg.det(h)) for h in hs]
        mu = me[0] + p * v[0] - me[1] - p * v[1]
        mu = mu * beta - 48.0 * np.log(v[0] / v[1])
    energies = []
    for cyc in range(1, cycles + 1):
        if cyc % 10 == 0:                                       # mc_moves.F90:217-222
            lists = [c_oracle.neighbours(xs[l], ivs[l]) for l in range(len(xs))]
        r = so.sweep(48, SEED, 0, (cyc - 1) * 48, hs, xs, beta, 1.1 * ANG_TO_BOHR, grid=grid, weight=weight,
                     ls=ls, ls_mu=mu, model_energy=me, lists=lists)
        xs = [r["xyz"][l] for l in range(len(xs))]
        me, ls, mu = list(r["model_energy"]), r["ls"], r["ls_mu"]
        energies.append(me[ls - 1] * HART_TO_EV)
    return np.array(xs), np.array(energies), ls


@pytest.fixture(scope="module")
def so():
    from oracle import SweepOracle
    return SweepOracle()


def test_single_box_translation_moves_match_the_reference_program(tmp_path, so, c_oracle):
    boxes, e_ref, ljr, ls = run_reference(str(tmp_path / "run"), 1, 220, 40)
    xs, e_or, _ = replay(so, c_oracle, boxes, 220.0, 40)
    assert np.abs(xs - ljr).max() < 1e-11                       # 1920 trial moves later: the same configuration
    assert np.allclose(e_or, e_ref, rtol=2e-6, atol=2e-6)       # the thermo file holds 6-7 digits
    assert np.abs(ljr[0] - boxes[0][1]).max() > 0.5             # ... and it did move


def test_two_lattice_moves_and_weights_match_the_reference_program(tmp_path, so, c_oracle):
    from mc_water_ls_mw_amd.sweep import MuGrid
    grid = MuGrid(101, -400.0, 400.0)
    weight = 3.0 * np.exp(-((grid.mu_bin - 60.0) / 120.0) ** 2) + 0.004 * np.abs(grid.mu_bin)
    boxes, e_ref, ljr, ls = run_reference(str(tmp_path / "run"), 2, 200, 40, weight=weight, grid=grid)
    xs, e_or, ls_or = replay(so, c_oracle, boxes, 200.0, 40, grid=grid, weight=weight)
    assert ls == ls_or == 1
    assert np.abs(xs - ljr).max() < 1e-10
    assert np.allclose(e_or, e_ref, rtol=2e-6, atol=2e-6)
    assert np.abs(ljr[1] - boxes[1][1]).max() > 0.5


def replay_cycle(so, c_oracle, boxes, temperature, cycles, grid, weight, samplerun, wl_factor):
    """mc_cycle with translations, mc_update_wl_bins and a lattice-switch attempt after every move."""
    from mc_water_ls_mw_amd.lattice import ANG_TO_BOHR
    from mc_water_ls_mw_amd.sweep import KB
    beta = 1.0 / (KB * temperature)
    hs = [b[0] for b in boxes]
    xs = [np.array(b[1]) for b in boxes]
    ivs = [c_oracle.ivects(h) for h in hs]
    lists = [c_oracle.neighbours(xs[l], ivs[l]) for l in range(2)]
    me = [c_oracle.model_energy(xs[l], ivs[l], *lists[l]) for l in range(2)]
    p = 1.0 / AUP_TO_ATM
    v = [abs(np.linalg.det(h)) for h in hs]
    mu = me[0] + p * v[0] - me[1] - p * v[1]
    mu = mu * beta - 48.0 * np.log(v[0] / v[1])
    ls, w, hi, uh = 1, np.array(weight, dtype=float), np.zeros(grid.nbins), np.zeros(grid.nbins)
    energies, nsw = [], 0
    for cyc in range(1, cycles + 1):
        if cyc % 10 == 0:
            lists = [c_oracle.neighbours(xs[l], ivs[l]) for l in range(2)]
        r = so.cycle(48, SEED, 0, (cyc - 1) * 48, hs, xs, beta, 1.1 * ANG_TO_BOHR, grid, w, hi, uh, ls=ls, ls_mu=mu,
                     model_energy=me, lists=lists, record=True, samplerun=samplerun, always_switch=True, npt=False,
                     wl_factor=wl_factor, pressure=p)
        xs = [r["xyz"][0], r["xyz"][1]]
        me, ls, mu, w, hi, uh = list(r["model_energy"]), r["ls"], r["ls_mu"], r["weight"], r["histogram"], r["unbiased_hist"]
        nsw += r["switches"]
        energies.append(me[ls - 1] * HART_TO_EV)
    return np.array(xs), np.array(energies), ls, w, hi, nsw


def test_wang_landau_weight_generation_matches_the_reference_program(tmp_path, so, c_oracle):
    """examples/ice1_gen_weights in miniature: no weights file, Wang-Landau updates of the visited bin after
    every move (wl_factor = 0.05 as a single-precision literal, userparams.f90:32), lattice-switch attempt after
    every move.  The reference's checkpoint holds histogram and weights: both must be reproduced."""
    from mc_water_ls_mw_amd.sweep import MuGrid
    grid = MuGrid(101, -400.0, 400.0)
    boxes, e_ref, ljr, ls, hist, wgt = run_reference(str(tmp_path / "run"), 2, 200, 40, samplerun=False,
                                                     always_switch=True, tables=True)
    xs, e_or, ls_or, w, hi, nsw = replay_cycle(so, c_oracle, boxes, 200.0, 40, grid, np.zeros(101), False,
                                               float(np.float32(0.05)))
    assert np.abs(xs - ljr).max() < 1e-10 and ls == ls_or
    assert np.allclose(e_or, e_ref, rtol=2e-6, atol=2e-6)
    assert hist.sum() > 0 and np.allclose(hi, hist, rtol=1e-12, atol=1e-12)
    assert wgt.max() > 0 and np.allclose(w, wgt, rtol=1e-11, atol=1e-12)


def test_sampling_run_with_switches_matches_the_reference_program(tmp_path, so, c_oracle):
    """examples/ice1_sample in miniature: fixed weights, histogram accumulation, switch attempt after every move."""
    from mc_water_ls_mw_amd.sweep import MuGrid
    grid = MuGrid(101, -400.0, 400.0)
    weight = 0.02 * np.abs(grid.mu_bin)          # pulls mu towards 0, where lattice switches are accepted
    boxes, e_ref, ljr, ls, hist, wgt = run_reference(str(tmp_path / "run"), 2, 200, 40, weight=weight, grid=grid,
                                                     samplerun=True, always_switch=True, tables=True)
    xs, e_or, ls_or, w, hi, nsw = replay_cycle(so, c_oracle, boxes, 200.0, 40, grid, weight, True, 0.0)
    assert np.abs(xs - ljr).max() < 1e-10 and ls == ls_or
    assert np.allclose(e_or, e_ref, rtol=2e-6, atol=2e-6)
    assert np.allclose(hi, hist, rtol=1e-12, atol=1e-12) and np.allclose(w, wgt, rtol=1e-13)


def test_npt_run_with_volume_moves_matches_the_reference_program(tmp_path, so, c_oracle):
    """The whole move set of mc_cycle: translations, volume moves (cell change, all positions rescaled through
    fractional coordinates, full-box energies with the existing lists, restore on rejection), Wang-Landau updates and
    a lattice-switch attempt after every move -- NPT, two lattices, 40 cycles with ~1 move in 6 a volume move."""
    from mc_water_ls_mw_amd.lattice import ANG_TO_BOHR
    from mc_water_ls_mw_amd.sweep import KB, MuGrid
    from oracle import FullSweepState
    grid = MuGrid(101, -400.0, 400.0)
    vol_prob = 0.1
    transP = 0.5 / (0.5 + vol_prob + 0.0)                      # mc_moves.F90:157-166
    boxes, e_ref, ljr, ls, hist, wgt = run_reference(str(tmp_path / "run"), 2, 200, 40, samplerun=False, always_switch=True,
                                                     tables=True, npt=True, vol_prob=vol_prob, transP=transP)
    beta, p = 1.0 / (KB * 200.0), 1.0 / AUP_TO_ATM
    st = FullSweepState(c_oracle, [b[0] for b in boxes], [b[1] for b in boxes])
    mu = st.model_energy[0] + p * st.volume[0] - st.model_energy[1] - p * st.volume[1]
    st.ls_mu = mu * beta - 48.0 * np.log(st.volume[0] / st.volume[1])
    w, hi, uh = np.zeros(101), np.zeros(101), np.zeros(101)
    energies = []
    for cyc in range(1, 41):
        if cyc % 10 == 0:                                      # compute_neighbours also rebuilds the image vectors
            st.rebuild_lists(c_oracle)
        so.full(st, 48, SEED, 0, (cyc - 1) * 48, transP, 0.924 * ANG_TO_BOHR, beta, 1.1 * ANG_TO_BOHR, grid, w, hi, uh,
                record=True, samplerun=False, always_switch=True, npt=True, wl_factor=float(np.float32(0.05)), pressure=p)
        energies.append(st.model_energy[st.ls - 1] * HART_TO_EV)
    assert st.nvol[0] > 100 and 0 < st.nvol[1] < st.nvol[0]                   # volume moves happened, some accepted
    assert np.abs(st.h - run_reference.hmatrix).max() < 1e-10                 # the same cell ...
    assert np.abs(st.xyz - ljr).max() < 1e-9 and st.ls == ls                  # ... the same configuration
    assert np.allclose(np.array(energies), e_ref, rtol=2e-6, atol=2e-6)
    assert np.allclose(hi, hist, rtol=1e-12, atol=1e-12) and np.allclose(w, wgt, rtol=1e-10, atol=1e-11)


def test_chain_synchronisation_matches_the_reference_program(tmp_path, so, c_oracle):
    """NPT two-lattice run with latt_sync_int = 5: every fifth cycle the reference re-imposes lattice 2 from lattice 1
    (mc_check_chain_synchronisation).  The oracle -- volume moves carrying ref_ljr, mwo_chain_sync -- must end at the
    reference's checkpointed ljr, ref_ljr and cells."""
    from mc_water_ls_mw_amd.lattice import ANG_TO_BOHR
    from mc_water_ls_mw_amd.sweep import KB, MuGrid
    from oracle import FullSweepState
    grid = MuGrid(101, -400.0, 400.0)
    vol_prob = 0.1
    transP = 0.5 / (0.5 + vol_prob)
    boxes, e_ref, ljr, ls, hist, wgt = run_reference(str(tmp_path / "run"), 2, 200, 30, samplerun=False, always_switch=True,
                                                     tables=True, npt=True, vol_prob=vol_prob, transP=transP, latt_sync=5)
    beta, p = 1.0 / (KB * 200.0), 1.0 / AUP_TO_ATM
    st = FullSweepState(c_oracle, [b[0] for b in boxes], [b[1] for b in boxes])
    mu = st.model_energy[0] + p * st.volume[0] - st.model_energy[1] - p * st.volume[1]
    st.ls_mu = mu * beta - 48.0 * np.log(st.volume[0] / st.volume[1])
    w, hi, uh = np.zeros(101), np.zeros(101), np.zeros(101)
    energies = []
    for cyc in range(1, 31):
        if cyc % 10 == 0:
            st.rebuild_lists(c_oracle)
        so.full(st, 48, SEED, 0, (cyc - 1) * 48, transP, 0.924 * ANG_TO_BOHR, beta, 1.1 * ANG_TO_BOHR, grid, w, hi, uh,
                record=True, samplerun=False, always_switch=True, npt=True, wl_factor=float(np.float32(0.05)), pressure=p)
        if cyc % 5 == 0:                                       # mc_moves.F90:297-300 (after the moves of the cycle)
            so.chain_sync(st, beta, p)
        energies.append(st.model_energy[st.ls - 1] * HART_TO_EV)
    assert np.abs(st.h - run_reference.hmatrix).max() < 1e-10
    assert np.abs(st.ref_xyz - run_reference.ref_ljr).max() < 1e-9
    assert np.abs(st.xyz - ljr).max() < 1e-9 and st.ls == ls
    assert np.allclose(np.array(energies), e_ref, rtol=2e-6, atol=2e-6)
    assert np.allclose(w, wgt, rtol=1e-10, atol=1e-11)
