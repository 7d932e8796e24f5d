"""GPU: seeded random systems against the C oracle -- the edge cases no fixture was written for.  Random triclinic
cells, sizes from 2 to 400 molecules, densities from a dilute gas (ragged lists, molecules with no neighbour at all)
through ice to compressed lattices (rows beyond 32 entries, more than 24 in-range neighbours), cells narrower than the
list radius (a molecule neighbours its own images), and both single-move paths: enough requests per box for the
LDS-staged kernel with its dynamic request hand-out, and a handful per box for the L2-gather kernel.
Bars as everywhere: lists entry for entry, counts as integers, energies 1e-10 relative, Delta E 1e-10 Ha absolute."""
import numpy as np
import pytest

from conftest import DE_ATOL, RTOL

pytestmark = pytest.mark.gpu


def random_system(rng):
    from mc_water_ls_mw_amd import lattice as lat
    kind = rng.choice(["gas", "ice", "dense", "thin"])
    if kind == "gas":
        n = int(rng.integers(2, 120))
        L = rng.uniform(14.0, 40.0, 3) * lat.ANG_TO_BOHR
        h = np.diag(L) + rng.uniform(-0.1, 0.1, (3, 3)) * L.min()
        x = rng.random((n, 3)) @ h
        # keep molecules apart (r > 2.0 A): the potential explodes below that, as it does in the reference
        keep = [0]
        for i in range(1, n):
            d = x[keep] - x[i]
            if np.all(np.linalg.norm(d, axis=1) > 2.0 * lat.ANG_TO_BOHR):
                keep.append(i)
        x = x[keep]
    else:
        d_oo = {"ice": rng.uniform(2.6, 2.9), "dense": rng.uniform(2.15, 2.4), "thin": rng.uniform(2.6, 2.9)}[kind]
        cell = lat.ice_ic_cell if rng.random() < 0.5 else lat.ice_ih_cell
        h, x = cell(d_oo)
        reps = {"ice": rng.integers(2, 4, 3), "dense": rng.integers(2, 4, 3), "thin": np.array([1, int(rng.integers(2, 4)), int(rng.integers(1, 3))])}[kind]
        h, x = lat.replicate(h, x, tuple(int(r) for r in reps))
        x = lat.thermalise(x, float(rng.uniform(0.02, 0.2)), int(rng.integers(1, 10 ** 6)))
        if rng.random() < 0.3:                                   # shear the cell a little, positions with it
            shear = np.eye(3) + rng.uniform(-0.06, 0.06, (3, 3)) * (1 - np.eye(3))
            h, x = h @ shear, x @ shear
    return kind, np.ascontiguousarray(h), np.ascontiguousarray(x)


@pytest.mark.parametrize("seed", range(36))
def test_random_systems_follow_the_oracle(seed, c_oracle):
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.energy import load_boxes
    rng = np.random.default_rng(7000 + seed)
    kind, h, x = random_system(rng)
    n = len(x)
    iv = c_oracle.ivects(h)
    if len(iv) > 1000:
        pytest.skip("cell too thin for the image table")
    nn, jn, vn = c_oracle.neighbours(x, iv, 64)
    if nn.max() > 64:
        pytest.skip("denser than maxneigh = 64")
    e_ref, counts = c_oracle.model_energy(x, iv, nn, jn, vn, counts=True)
    em = load_boxes([h], [x], maxneigh=64)
    try:
        assert np.array_equal(em.ivect(1), iv)
        gnn, gjn, gvn = em.neighbours(1)
        assert np.array_equal(gnn, nn) and np.array_equal(gjn, jn) and np.array_equal(gvn, vn), kind
        assert abs(em.model_energy[0] - e_ref) <= RTOL * abs(e_ref) + 1e-14, kind
        assert em.model_energy_counts(1) == (int(counts[0]), int(counts[1]))
        ref = c_oracle.local_energy_all(x, iv, nn, jn, vn)
        loc = em.local_energy_batch(1, np.arange(1, n + 1))
        assert np.all(np.abs(loc - ref) <= RTOL * np.abs(ref) + 1e-14), kind
        # many requests: LDS-staged kernel (dynamic hand-out); a few: the L2-gather kernel
        for nreq, s in ((max(600, 3 * n), 11), (min(n, 5), 12)):
            imol, trial = lat.trial_moves(x, nreq, max_trans_ang=float(rng.uniform(0.2, 1.1)), seed=s + seed)
            eo, en = em.delta_energy_batch(1, imol, trial)
            ro, rn = c_oracle.trial_moves(imol, trial, x, iv, nn, jn, vn)
            big = np.abs(rn) > 1.0                                   # a trial position on top of a neighbour: E ~ r^-4
            assert np.all(np.abs(eo - ro) <= RTOL * np.abs(ro) + 1e-14), (kind, nreq)
            assert np.all(np.abs(en - rn) <= RTOL * np.abs(rn) + 1e-14), (kind, nreq)
            assert np.all(np.abs((en - eo) - (rn - ro))[~big] <= DE_ATOL), (kind, nreq)
    finally:
        em.energy_deinit()


@pytest.mark.parametrize("seed", range(16))
def test_random_systems_through_the_monte_carlo_driver(seed, c_oracle):
    """The same random systems through the device-resident driver (one lattice, two walkers): every residency of a walker's
    data (positions + rows in LDS, positions only, global memory with and without look-ahead), ragged and over-long rows,
    cells thin enough for self images -- the chain is the oracle's move for move."""
    from mc_water_ls_mw_amd.energy import load_boxes
    from mc_water_ls_mw_amd.sweep import WalkerFarm
    from oracle import SweepOracle
    from test_sweep import _compare
    rng = np.random.default_rng(9000 + seed)
    kind, h, x = random_system(rng)
    if len(x) < 2:
        pytest.skip("a single molecule has nothing to move against")
    iv = c_oracle.ivects(h)
    if len(iv) > 1000:
        pytest.skip("cell too thin for the image table")
    if c_oracle.neighbours(x, iv, 64)[0].max() > 60:
        pytest.skip("too dense for maxneigh = 64 once molecules move")
    so = SweepOracle()
    from mc_water_ls_mw_amd import lattice as lat
    boxes = [(h, lat.thermalise(x, 0.02, 70 + w)) for w in range(2)]
    nmoves = 160
    em = load_boxes([b[0] for b in boxes], [b[1] for b in boxes], maxneigh=64)
    farm = WalkerFarm(em, 1, float(rng.uniform(150.0, 400.0)), float(rng.uniform(0.3, 1.1)))
    try:
        farm.set_states(1, np.zeros(2))
        log = farm.sweep(nmoves, seed=31 + seed, move0=5, log=True)
        pos = [farm.positions(b) for b in (1, 2)]
        st = [farm.state(w) for w in (1, 2)]
        for w in range(2):
            ref = so.sweep(nmoves, 31 + seed, w, 5, [boxes[w][0]], [boxes[w][1]], farm.beta, farm.max_trans, maxneigh=64)
            _compare(log[w], ref, st[w], [pos[w]])
    finally:
        em.energy_deinit()
