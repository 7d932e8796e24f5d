"""CPU: the reference's file formats against files the reference program itself wrote, and a
restart of the reference program from a checkpoint written by us."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT

RNG = os.path.join(ROOT, "oracle", "_ref", "mc_water_ref_rng")
pytestmark = pytest.mark.skipif(not os.path.exists(RNG), reason="oracle/_ref/mc_water_ref_rng not built")


def _run(d, cycles, weight_grid=None, samplerun=False):
    import test_sweep_pin as P
    from mc_water_ls_mw_amd import io as mwio
    return P.run_reference(d, 2, 200, cycles, samplerun=samplerun, always_switch=True, tables=True)


def test_checkpoint_roundtrip_is_byte_identical(tmp_path):
    from mc_water_ls_mw_amd import io as mwio
    d = str(tmp_path / "run")
    boxes, e, ljr, ls, hist, wgt = _run(d, 20)
    path, c = mwio.latest_checkpoint(d)
    assert c["nwater"] == 48 and c["cycle"] == 20 and c["ls"] == ls and not c["samplerun"]
    assert np.array_equal(c["ljr"], ljr) and np.array_equal(c["histogram"], hist) and np.array_equal(c["weight"], wgt)
    assert c["hmatrix"].shape == (2, 3, 3) and np.allclose(c["hmatrix"][0], boxes[0][0], rtol=1e-12)
    out = str(tmp_path / "copy.dat")
    mwio.write_checkpoint(out, c)
    assert open(out, "rb").read() == open(path, "rb").read()


def test_reference_restarts_from_our_checkpoint(tmp_path):
    """40 cycles in one go == 20 cycles, checkpoint re-written by write_checkpoint, restart for 20 more."""
    import test_sweep_pin as P
    from mc_water_ls_mw_amd import io as mwio
    full = _run(str(tmp_path / "full"), 40)
    d = str(tmp_path / "half")
    _run(d, 20)
    path, c = mwio.latest_checkpoint(d)
    for f in os.listdir(d):
        if f.startswith("checkpoint"):
            os.remove(os.path.join(d, f))
    mwio.write_checkpoint(os.path.join(d, "checkpoint000.dat.1"), c)
    nl = P.namelist(2, 200, 20, False, True).replace("chkpt_dump_int   = 20", "chkpt_dump_int   = 40")
    open(os.path.join(d, "ice.input"), "w").write(nl)
    # the restarted run draws moves 960.. of the same stream only if its wrapper starts there: it does not (the counter
    # restarts at 0), so compare what a restart must preserve exactly: it LOADS our file and continues from cycle 20
    out = subprocess.run([RNG, "ice.input"], cwd=d, capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, MW_WRAP_SWITCH="1", MW_WRAP_TRANSP="1.0"))
    assert out.returncode == 0, out.stderr[-500:]
    _, c2 = mwio.latest_checkpoint(d)
    assert c2["cycle"] == 40                                    # continued from our cycle 20 for 20 more
    assert c2["histogram"].sum() > c["histogram"].sum()         # ... on top of our histogram


def test_table_files(tmp_path):
    from mc_water_ls_mw_amd import io as mwio
    from mc_water_ls_mw_amd.sweep import MuGrid
    g = MuGrid(101, -400.0, 400.0)
    w = 0.01 * np.abs(g.mu_bin)
    p = str(tmp_path / "eta_weights.dat")
    mwio.write_table(p, 0.05, g.mu_bin, w)
    wl, mu, val = mwio.read_table(p)
    assert wl == pytest.approx(0.05) and np.array_equal(mu, g.mu_bin) and np.array_equal(val, w)
    ref = os.path.join("/root/reference/examples/ice1_sample/eta_weights.dat")
    if os.path.exists(ref):                                     # the reference's own shipped table parses
        wl, mu, val = mwio.read_table(ref)
        assert len(mu) == 101 and wl == pytest.approx(0.05, rel=1e-6) and np.all(np.diff(mu) > 0)
