"""CPU: pin the C restatement (oracle/mw_oracle.c) against every golden vector
that tests/golden/make_golden.py produced from the reference's own Fortran, and
-- where the compiled reference is present (oracle/_ref/libmw_ref.so) -- against
the reference directly on fresh seeded inputs."""
import numpy as np
import pytest

from conftest import GOLDEN, golden_names, list_digest, load_golden

# the C oracle keeps the reference's operation order, so agreement is at rounding level
TIGHT = 1e-14

CASES = [n for n in golden_names() if "32768" not in n]


def test_constants_match_reference(c_oracle):
    ref = np.load(GOLDEN + "/constants.npz")["constants"]
    assert np.array_equal(c_oracle.constants(), ref)
    # G1: cos0 is the float32 literal widened, not -1/3 and not the decimal literal
    assert c_oracle.constants()[7] == float(np.float32(-0.33331324756))
    assert c_oracle.constants()[7] != -0.33331324756


@pytest.mark.parametrize("name", CASES)
def test_c_oracle_matches_golden(name, c_oracle):
    z = load_golden(name)
    iv = c_oracle.ivects(z["h"])
    assert np.array_equal(iv, z["ivect"])
    nn, jn, vn = c_oracle.neighbours(z["xyz"], iv)
    assert np.array_equal(nn, z["nn"])
    assert list_digest(nn, jn, vn) == str(z["list_sha256"])
    if "jn" in z:
        assert np.array_equal(jn, z["jn"]) and np.array_equal(vn, z["vn"])
    e = c_oracle.model_energy(z["xyz"], iv, nn, jn, vn)
    assert abs(e - float(z["model_energy"])) <= TIGHT * abs(float(z["model_energy"]))
    loc = c_oracle.local_energy_all(z["xyz"], iv, nn, jn, vn)
    if "local" in z:
        assert np.all(np.abs(loc - z["local"]) <= TIGHT * np.abs(z["local"]) + 1e-300)
    assert abs(loc.sum() - float(z["local_sum"])) <= 1e-12 * abs(float(z["local_sum"])) + 1e-300
    if "trial_imol" in z:
        eo, en = c_oracle.trial_moves(z["trial_imol"], z["trial_xyz"], z["xyz"], iv, nn, jn, vn)
        assert np.all(np.abs(eo - z["trial_old"]) <= TIGHT * np.abs(z["trial_old"]))
        assert np.all(np.abs(en - z["trial_new"]) <= TIGHT * np.abs(z["trial_new"]))


def test_known_answers_of_the_survey():
    """The numbers SURVEY.md 8(c) measured independently from the compiled reference."""
    assert float(load_golden("ic48")["model_energy"]) == pytest.approx(-9.39190389722995067e-01, rel=1e-15)
    assert float(load_golden("ih48")["model_energy"]) == pytest.approx(-9.40303865665672145e-01, rel=1e-15)
    assert float(load_golden("ic48")["local"][0]) == pytest.approx(-3.90572545543489205e-02, rel=1e-15)
    assert float(load_golden("ih48")["local"][0]) == pytest.approx(-3.90901396590697392e-02, rel=1e-15)
    assert float(load_golden("ic96")["model_energy"]) == pytest.approx(-1.87838077944598947e+00, rel=1e-14)
    assert float(load_golden("ic1536")["model_energy"]) == pytest.approx(-3.00540924711369186e+01, rel=1e-13)
    assert float(load_golden("ih1536")["model_energy"]) == pytest.approx(-3.00897237013013772e+01, rel=1e-13)


def test_replication_invariance_and_g4(c_oracle):
    """E/N is invariant under supercell replication; sum_i local = 2 E2 + 3 E3."""
    e48 = float(load_golden("ic48")["model_energy"])
    assert float(load_golden("ic96")["model_energy"]) == pytest.approx(2 * e48, rel=1e-13)
    assert float(load_golden("ic1536")["model_energy"]) == pytest.approx(32 * e48, rel=1e-13)
    z = load_golden("ih48_t020")
    iv = c_oracle.ivects(z["h"])
    nn, jn, vn = c_oracle.neighbours(z["xyz"], iv)
    e, counts = c_oracle.model_energy(z["xyz"], iv, nn, jn, vn, counts=True)
    loc, lc = c_oracle.local_energy_all(z["xyz"], iv, nn, jn, vn, counts=True)
    assert lc[0] == counts[0] and lc[1] == 3 * counts[1]   # every pair seen from both ends, every triplet from 3 atoms
    e3 = loc.sum() - 2 * e        # = E3
    e2 = e - e3
    assert e3 > 0 and e2 < 0
    assert loc.sum() == pytest.approx(2 * e2 + 3 * e3, rel=1e-13)


def test_edge_cases(c_oracle):
    z = load_golden("single_atom")
    assert int(z["nn"][0]) == 0 and float(z["model_energy"]) == 0.0
    z = load_golden("dimer")
    assert list(z["nn"]) == [1, 1]
    assert float(z["local_sum"]) == pytest.approx(2 * float(z["model_energy"]), rel=1e-15)   # no triplets
    z = load_golden("ih8_small")
    assert len(z["ivect"]) == 45 and int(z["nn"].max()) == 26
    # atoms neighbour their own periodic images there
    own = (z["jn"] == np.arange(1, 9)[:, None]) & (z["vn"] > 1)
    assert own.any()
    # overflow is reported, never silent (the reference overflows silently, G9)
    iv = c_oracle.ivects(z["h"])
    with pytest.raises(RuntimeError, match="overflow"):
        c_oracle.neighbours(z["xyz"], iv, maxneigh=20)


@pytest.mark.parametrize("kind,reps,sigma,seed", [("ih", (2, 1, 1), 0.12, 3), ("ic", (2, 2, 1), 0.2, 4), ("ih", (3, 2, 2), 0.15, 5)])
def test_c_oracle_against_compiled_reference(kind, reps, sigma, seed, c_oracle):
    from oracle import RefOracle
    if not RefOracle.available():
        pytest.skip("oracle/_ref/libmw_ref.so not built (needs /root/reference, build container only)")
    from mc_water_ls_mw_amd import lattice as lat
    h, xyz = lat.ice_box(kind, reps, sigma, seed=seed)
    R = RefOracle()
    R.load([h], [xyz])
    iv = c_oracle.ivects(h)
    assert np.array_equal(iv, R.ivects(1))
    nn, jn, vn = c_oracle.neighbours(xyz, iv)
    rnn, rjn, rvn = R.neighbours(1)
    assert np.array_equal(nn, rnn) and np.array_equal(jn, rjn) and np.array_equal(vn, rvn)
    assert c_oracle.model_energy(xyz, iv, nn, jn, vn) == pytest.approx(R.model_energy(1), rel=TIGHT)
    assert np.allclose(c_oracle.local_energy_all(xyz, iv, nn, jn, vn), R.local_energy_all(1), rtol=TIGHT, atol=0)
    imol, trial = lat.trial_moves(xyz, 64, seed=seed)
    eo, en = c_oracle.trial_moves(imol, trial, xyz, iv, nn, jn, vn)
    ro, rn = R.trial_moves(1, imol, trial)
    assert np.allclose(eo, ro, rtol=TIGHT, atol=0) and np.allclose(en, rn, rtol=TIGHT, atol=0)
