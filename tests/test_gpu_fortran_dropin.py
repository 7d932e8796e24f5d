"""GPU: the Fortran boundary.  tests/fortran/dropin_driver.f90 drives `module energy`
the way mc_moves.F90 does (old/new local energy around host-side moves with silent
reverts, caller-edited model_energy, list refresh, drift check, volume move).  It is
built twice by oracle/Makefile (build container only; the binaries travel in
oracle/_ref/): against the reference's molint.F90 and against the ISO_C_BINDING
replacement module + libmw_hip.so.  Every printed energy must agree to 1e-10."""
import os
import subprocess

import numpy as np
import pytest

from conftest import DE_ATOL, ROOT, RTOL, load_golden

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "dropin_ref")
HIP_BIN = os.path.join(ROOT, "oracle", "_ref", "dropin_hip")


def write_input(path, boxes, nmoves, refresh, seed):
    with open(path, "w") as fh:
        n = len(boxes[0][1])
        fh.write(f"{n} {len(boxes)} {nmoves} {refresh} {seed}\n")
        for h, xyz in boxes:
            fh.write(" ".join(repr(float(v)) for v in np.asarray(h).reshape(-1)) + "\n")
            for r in xyz:
                fh.write(" ".join(repr(float(v)) for v in r) + "\n")


def run(binary, inp):
    out = subprocess.run([binary, inp], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = []
    for ln in out.stdout.splitlines():
        f = ln.split()
        rows.append((f[0], [float(v) for v in f[1:]]))
    return rows


def compare(a, b):
    assert len(a) == len(b) and len(a) > 10
    for (ta, va), (tb, vb) in zip(a, b):
        assert ta == tb
        if ta in ("nivec",):
            assert va == vb
        elif ta == "move":
            assert va[:2] == vb[:2]                         # lattice, molecule
            for x, y in zip(va[2:], vb[2:]):
                assert abs(x - y) <= RTOL * abs(y) + 1e-14
            assert abs((va[3] - va[2]) - (vb[3] - vb[2])) <= DE_ATOL
        else:
            assert va[0] == vb[0]
            assert abs(va[1] - vb[1]) <= RTOL * abs(vb[1]), (ta, va, vb)


needs_bins = pytest.mark.skipif(not (os.path.exists(REF_BIN) and os.path.exists(HIP_BIN)),
                                reason="oracle/_ref/dropin_{ref,hip} not built (build container: make -C oracle dropin)")


@needs_bins
@pytest.mark.gpu
def test_two_lattice_dropin_matches_reference(tmp_path):
    """The ice1_sample system: lattice 1 = input001 (Ic), lattice 2 = input002 (Ih), thermalised."""
    z1, z2 = load_golden("ic48_t015"), load_golden("ih48_t020")
    inp = str(tmp_path / "in.txt")
    write_input(inp, [(z1["h"], z1["xyz"]), (z2["h"], z2["xyz"])], nmoves=400, refresh=50, seed=12345)
    compare(run(HIP_BIN, inp), run(REF_BIN, inp))


@needs_bins
@pytest.mark.gpu
def test_single_box_1536_dropin_matches_reference(tmp_path):
    z = load_golden("ih1536_t012")
    inp = str(tmp_path / "in.txt")
    write_input(inp, [(z["h"], z["xyz"])], nmoves=300, refresh=100, seed=777)
    compare(run(HIP_BIN, inp), run(REF_BIN, inp))


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/dropin_ref not built")
def test_reference_driver_reproduces_golden(tmp_path):
    """CPU: the driver linked with the reference's own module prints the golden full-box energy
    (pins the driver itself, so that the GPU comparison above means something)."""
    z = load_golden("ih48_t020")
    inp = str(tmp_path / "in.txt")
    write_input(inp, [(z["h"], z["xyz"])], nmoves=20, refresh=7, seed=3)
    rows = run(REF_BIN, inp)
    assert rows[0][0] == "init" and rows[0][1][1] == pytest.approx(float(z["model_energy"]), rel=1e-15)
    accum = [v for t, v in rows if t == "accum"][0][1]
    fresh = [v for t, v in rows if t == "fresh"][0][1]
    assert abs(accum - fresh) < 1e-10          # the reference's own drift check, mc_moves.F90:1099
