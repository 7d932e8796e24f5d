"""CPU: pin the -DMINU variant of the move routines (mc_moves.F90:1119-1140,1168-1170,1385-1401,1426-1429: an accepted
translation or volume move also takes the walker to the lattice of lower enthalpy, with the switch's terms in the
acceptance) against the REFERENCE PROGRAM compiled that way (oracle/_ref/mc_water_ref_rng_minu: mc_water_ref_rng with
mc_moves.F90 built -DMINU, oracle/Makefile).  MINU is a compile-time choice of the reference, off in all its examples;
here it is a run option of the oracle and of the device driver.  leshift is switched on in these runs: without it the
enthalpy gap between the two 48-molecule ices (~110 kT) leaves nothing for MINU to decide."""
import os

import numpy as np
import pytest

import test_sweep_pin as pin

MINU = os.path.join(pin.ROOT, "oracle", "_ref", "mc_water_ref_rng_minu")
pytestmark = pytest.mark.skipif(not os.path.exists(MINU), reason="oracle/_ref/mc_water_ref_rng_minu not built")

F0 = float(np.float32(0.05))     # userparams.f90:32: a single-precision literal


@pytest.fixture(scope="module")
def so():
    from oracle import SweepOracle
    s = SweepOracle()
    yield s
    s.set_leshift(0.0, 0.0); s.set_minu(False)


def replay(so, c_oracle, boxes, cycles, npt, transP, always_switch, minu=True):
    from mc_water_ls_mw_amd.lattice import ANG_TO_BOHR
    from mc_water_ls_mw_amd.sweep import KB, MuGrid
    from oracle import FullSweepState
    grid = MuGrid(101, -400.0, 400.0)
    beta, p = 1.0 / (KB * 200.0), 1.0 / pin.AUP_TO_ATM
    st = FullSweepState(c_oracle, [b[0] for b in boxes], [b[1] for b in boxes])
    ref = [st.model_energy[l] + (p * st.volume[l] if npt else 0.0) for l in range(2)]          # main.f90:146-147
    mu = st.model_energy[0] + p * st.volume[0] - st.model_energy[1] - p * st.volume[1]
    mu = mu - (ref[0] - ref[1])                                                                 # main.f90:173
    st.ls_mu = mu * beta - 48.0 * np.log(st.volume[0] / st.volume[1])
    w, hi, uh = np.zeros(101), np.zeros(101), np.zeros(101)
    so.set_leshift(ref[0], ref[1]); so.set_minu(minu)
    visited = set()
    try:
        for cyc in range(1, cycles + 1):
            if cyc % 10 == 0:
                st.rebuild_lists(c_oracle)
            so.full(st, 48, pin.SEED, 0, (cyc - 1) * 48, transP, 0.924 * ANG_TO_BOHR, beta, 1.1 * ANG_TO_BOHR, grid, w, hi, uh,
                    record=True, samplerun=False, always_switch=always_switch, npt=npt, wl_factor=F0, pressure=p)
            visited.add(st.ls)
    finally:
        so.set_leshift(0.0, 0.0); so.set_minu(False)
    return st, w, hi, visited


def test_minu_translations_match_the_reference_program(tmp_path, so, c_oracle):
    """NVT, no explicit switch moves: every change of lattice in this run is MINU's."""
    boxes, e_ref, ljr, ls, hist, wgt = pin.run_reference(str(tmp_path / "run"), 2, 200, 30, samplerun=False, always_switch=False,
                                                         tables=True, mc_extra="leshift = .true.", program=MINU)
    st, w, hi, visited = replay(so, c_oracle, boxes, 30, False, 1.0, False)
    assert 2 in visited                                                         # the run starts in lattice 1: MINU moved it
    assert np.abs(st.xyz - ljr).max() < 1e-10 and st.ls == ls
    assert hist.sum() > 0 and np.allclose(hi, hist, rtol=1e-12, atol=1e-12) and np.allclose(w, wgt, rtol=1e-11, atol=1e-12)
    st0, w0, hi0, _ = replay(so, c_oracle, boxes, 30, False, 1.0, False, minu=False)   # the plain routine: another trajectory
    assert np.abs(hi0 - hist).max() > 0.5


def test_minu_volume_moves_match_the_reference_program(tmp_path, so, c_oracle):
    """NPT (~1 move in 6 a volume move) with a switch attempt after every move as well: both MINU branches, the volume terms
    of the translation branch (:1131-1133) and the old-volume terms of the volume branch (:1396-1397)."""
    vol_prob = 0.1
    transP = 0.5 / (0.5 + vol_prob)
    boxes, e_ref, ljr, ls, hist, wgt = pin.run_reference(str(tmp_path / "run"), 2, 200, 16, samplerun=False, always_switch=True,
                                                         tables=True, npt=True, vol_prob=vol_prob, transP=transP,
                                                         mc_extra="leshift = .true.", program=MINU)
    st, w, hi, visited = replay(so, c_oracle, boxes, 16, True, transP, True)
    assert st.nvol[0] > 60 and st.nvol[1] > 0 and 2 in visited
    assert np.abs(st.h - pin.run_reference.hmatrix).max() < 1e-10
    assert np.abs(st.xyz - ljr).max() < 1e-9 and st.ls == ls
    assert hist.sum() > 0 and np.allclose(hi, hist, rtol=1e-12, atol=1e-12) and np.allclose(w, wgt, rtol=1e-10, atol=1e-11)
