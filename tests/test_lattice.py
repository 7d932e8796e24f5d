"""CPU: host-side input generation (synthetic ice, xmol I/O, trial moves)."""
import numpy as np
import pytest

from mc_water_ls_mw_amd import lattice as lat


def _nearest(xyz, h):
    best = np.full(len(xyz), np.inf)
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            for k in (-1, 0, 1):
                d = xyz[None, :, :] + (i * h[0] + j * h[1] + k * h[2]) - xyz[:, None, :]
                r = np.sqrt((d * d).sum(-1))
                r[r < 1e-9] = np.inf
                best = np.minimum(best, r.min(1))
    return best


@pytest.mark.parametrize("kind", ["ih", "ic"])
def test_ideal_cells_are_tetrahedral(kind, c_oracle):
    h, xyz = lat.ice_box(kind, (2, 2, 2))
    assert len(xyz) == 64
    assert np.allclose(_nearest(xyz, h), lat.D_OO_ANG * lat.ANG_TO_BOHR, rtol=1e-12)
    iv = c_oracle.ivects(h)
    nn, jn, vn = c_oracle.neighbours(xyz, iv)
    assert set(nn) == {17 if kind == "ih" else 16}        # 4 + 12 (+1 for Ih), SURVEY.md G5
    _, counts = c_oracle.model_energy(xyz, iv, nn, jn, vn, counts=True)
    assert tuple(counts) == (4 * 64, 6 * 64)                # 10 interactions per atom on ideal ice


def test_replicate_order_and_cell():
    h, xyz = lat.ice_ic_cell()
    hh, xx = lat.replicate(h, xyz, (2, 3, 1))
    assert xx.shape == (48, 3) and np.allclose(hh, h * np.array([[2], [3], [1]]))
    assert np.allclose(xx[8 * 4:8 * 5], xyz + 1 * h[0] + 1 * h[1])   # replica (1,1,0) is block (1*3+1)*1+0 = 4


def test_xmol_roundtrip(tmp_path):
    h, xyz = lat.ice_box("ih", (1, 1, 1), 0.1, seed=3)
    p = tmp_path / "input001.xmol"
    lat.write_xmol(p, h, xyz)
    h2, xyz2 = lat.read_xmol(p)
    assert np.allclose(h2, h, atol=1e-10) and np.allclose(xyz2, xyz, atol=1e-10)


def test_trial_moves_follow_the_reference_recipe():
    h, xyz = lat.ice_box("ic", (2, 2, 2))
    imol, trial = lat.trial_moves(xyz, 5000, max_trans_ang=1.1, seed=9)
    assert imol.min() >= 1 and imol.max() <= 64 and imol.dtype == np.int32
    step = np.linalg.norm(trial - xyz[imol - 1], axis=1)
    assert step.max() <= 1.1 * lat.ANG_TO_BOHR * (1 + 1e-12)
    imol2, trial2 = lat.trial_moves(xyz, 5000, max_trans_ang=1.1, seed=9)
    assert np.array_equal(imol, imol2) and np.array_equal(trial, trial2)
