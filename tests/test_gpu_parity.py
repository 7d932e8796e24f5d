"""GPU parity: the HIP engine, called through the C ABI, against the golden
vectors generated from the reference's own Fortran (tests/golden/make_golden.py)
and against the C oracle on the same seeded inputs.

Bars: bit-exact for index work (image vectors are compared exactly too, the
neighbour list entry for entry, interaction counts as integers); energies within
1e-10 relative (BASELINE.json north_star), move energy changes within 1e-10 Ha
absolute (the reference author's own bound, mc_moves.F90:1099).
"""
import numpy as np
import pytest

from conftest import DE_ATOL, RTOL, golden_names, list_digest, load_golden

pytestmark = pytest.mark.gpu

SMALL = [n for n in golden_names() if "4096" not in n and "32768" not in n]
BIG = [n for n in golden_names() if "4096" in n]


def _engine(z, maxneigh=50):
    from mc_water_ls_mw_amd.energy import load_boxes
    return load_boxes([z["h"]], [z["xyz"]], maxneigh=maxneigh)


def _check_case(z, c_oracle, full_lists):
    em = _engine(z)
    try:
        # image vectors: exact
        assert em.nivect[0] == len(z["ivect"])
        assert np.array_equal(em.ivect(1), z["ivect"])
        # neighbour list: same set, same order, entry for entry
        nn, jn, vn = em.neighbours(1)
        assert np.array_equal(nn, z["nn"])
        assert list_digest(nn, jn, vn) == str(z["list_sha256"])
        if full_lists and "jn" in z:
            assert np.array_equal(jn, z["jn"]) and np.array_equal(vn, z["vn"])
        # full-box energy
        e_ref = float(z["model_energy"])
        assert abs(em.model_energy[0] - e_ref) <= RTOL * abs(e_ref)
        # interaction counts as the reference enumerates them: integers, exact
        iv = np.ascontiguousarray(z["ivect"])
        onn, ojn, ovn = c_oracle.neighbours(z["xyz"], iv)
        _, counts = c_oracle.model_energy(z["xyz"], iv, onn, ojn, ovn, counts=True)
        assert em.model_energy_counts(1) == (int(counts[0]), int(counts[1]))
        # local energies
        n = int(z["n"])
        loc = em.local_energy_batch(1, np.arange(1, n + 1))
        if "local" in z:
            ref = z["local"]
            assert np.all(np.abs(loc - ref) <= RTOL * np.abs(ref) + 1e-14)
        assert abs(loc.sum() - float(z["local_sum"])) <= RTOL * abs(float(z["local_sum"])) + 1e-13
        # G4: sum_i local = 2 E2 + 3 E3 with model = E2 + E3 -> both must be reproducible from the two numbers
        # trial moves: old/new local energy and their difference
        if "trial_imol" in z:
            eo, en = em.delta_energy_batch(1, z["trial_imol"], z["trial_xyz"])
            assert np.all(np.abs(eo - z["trial_old"]) <= RTOL * np.abs(z["trial_old"]) + 1e-14)
            assert np.all(np.abs(en - z["trial_new"]) <= RTOL * np.abs(z["trial_new"]) + 1e-14)
            assert np.all(np.abs((en - eo) - (z["trial_new"] - z["trial_old"])) <= DE_ATOL)
    finally:
        em.energy_deinit()


@pytest.mark.parametrize("name", SMALL)
def test_small_cases_match_reference(name, c_oracle):
    _check_case(load_golden(name), c_oracle, full_lists=True)


@pytest.mark.parametrize("name", BIG)
def test_4096_boxes_match_reference(name, c_oracle):
    _check_case(load_golden(name), c_oracle, full_lists=False)


def test_32768_stress_box_list_rebuild_and_full_energy():
    """BASELINE.json configs[4]: 32768-molecule ice Ih, neighbour-list rebuild + full energy
    (positions do not fit LDS: the full-box kernel gathers from L2)."""
    z = load_golden("ih32768_t015")
    em = _engine(z)
    try:
        nn, jn, vn = em.neighbours(1)
        assert np.array_equal(nn, z["nn"])
        assert list_digest(nn, jn, vn) == str(z["list_sha256"])
        e_ref = float(z["model_energy"])
        assert abs(em.model_energy[0] - e_ref) <= RTOL * abs(e_ref)
        loc = em.local_energy_batch(1, np.arange(1, int(z["n"]) + 1))
        assert abs(loc.sum() - float(z["local_sum"])) <= RTOL * abs(float(z["local_sum"]))
        # G4: sum_i local = 2 E2 + 3 E3 and model = E2 + E3, so E3 = sum_local - 2 model must be positive and small
        e3 = loc.sum() - 2 * em.model_energy[0]
        assert 0 < e3 < 0.2 * abs(e_ref)
    finally:
        em.energy_deinit()


def test_batched_walkers_match_single_box_results(c_oracle):
    """Many boxes in one launch (the bench path) give the same numbers as one box at a time."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.energy import load_boxes
    h, x0 = lat.ice_box("ih", (3, 2, 2), 0.0)
    xs = [lat.thermalise(x0, 0.15, 100 + w) for w in range(7)]
    em = load_boxes([h] * 7, xs)
    try:
        iv = c_oracle.ivects(h)
        e = em.model_energy_batch(1, 7)
        rng = np.random.default_rng(3)
        ils = rng.integers(1, 8, 500).astype(np.int32)
        imol = rng.integers(1, len(x0) + 1, 500).astype(np.int32)
        trial = np.stack([xs[b - 1][i - 1] for b, i in zip(ils, imol)]) + rng.normal(0, 0.5, (500, 3))
        eo, en = em.delta_energy_batch(ils, imol, trial)
        for w in range(7):
            nn, jn, vn = c_oracle.neighbours(xs[w], iv)
            ref = c_oracle.model_energy(xs[w], iv, nn, jn, vn)
            assert abs(e[w] - ref) <= RTOL * abs(ref)
            sel = np.nonzero(ils == w + 1)[0]
            ro, rn = c_oracle.trial_moves(imol[sel], trial[sel], xs[w], iv, nn, jn, vn)
            assert np.all(np.abs(eo[sel] - ro) <= RTOL * np.abs(ro))
            assert np.all(np.abs((en[sel] - eo[sel]) - (rn - ro)) <= DE_ATOL)
    finally:
        em.energy_deinit()


def test_persistent_workgroups_over_many_boxes(c_oracle):
    """More boxes than compute units, and not a multiple of them: the full-box kernel's workgroups are persistent (each takes
    every 256th box and reads its next box while the current one drains), so box k's energy must not depend on which
    workgroup served it, on what it served before, or on the sub-range of boxes a launch covers.  Every box has its own
    positions AND its own cell (image vectors are re-staged per box); the box is small, so a staging ticket of 768 doubles
    runs past its end into the following boxes."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.energy import load_boxes
    h0, x0 = lat.ice_box("ih", (3, 2, 2), 0.0)
    nb = 600
    rng = np.random.default_rng(11)
    scales = 1.0 + 0.01 * rng.standard_normal(nb)
    hs = [h0 * s for s in scales]
    xs = [lat.thermalise(x0, 0.15, 500 + w) * scales[w] for w in range(nb)]
    em = load_boxes(hs, xs)
    try:
        e_all = em.model_energy_batch(1, nb).copy()
        check = [0, 1, 255, 256, 257, 511, 512, 599] + [int(k) for k in rng.integers(0, nb, 12)]
        for w in check:
            iv = c_oracle.ivects(hs[w])
            nn, jn, vn = c_oracle.neighbours(xs[w], iv)
            ref = c_oracle.model_energy(xs[w], iv, nn, jn, vn)
            assert abs(e_all[w] - ref) <= RTOL * abs(ref), w
        # the same boxes through other launch shapes: a sub-range that starts in the middle, and one box at a time
        sub = em.model_energy_batch(201, 300)
        assert np.array_equal(sub, e_all[200:500])
        for w in (0, 256, 599):
            assert em.model_energy_batch(w + 1, 1)[0] == pytest.approx(e_all[w], rel=1e-13)
    finally:
        em.energy_deinit()


def _triclinic_box(reps, sigma, seed):
    from mc_water_ls_mw_amd import lattice as lat
    h, x = lat.ice_ic_cell(2.73)
    hs = h.copy()
    hs[1] += 0.23 * h[0]
    hs[2] += 0.11 * h[0] - 0.17 * h[1]
    xs = (x @ np.linalg.inv(h)) @ hs
    hh, xx = lat.replicate(hs, xs, reps)
    return hh, lat.thermalise(xx, sigma, seed)


@pytest.mark.parametrize("case", ["triclinic", "unwrapped", "dilute"])
def test_cell_grid_list_equals_reference_enumeration(case, c_oracle, monkeypatch):
    """The O(N) cell-grid builder must give the reference's list entry for entry -- also in a
    triclinic cell, with molecules that sit several cells outside the box (positions are never
    wrapped, G8; images the reference's table does not hold are not neighbours for it either),
    and in a dilute box with ragged rows.  The brute-force kernel is the cross-check."""
    from mc_water_ls_mw_amd.energy import load_boxes
    rng = np.random.default_rng(17)
    if case == "triclinic":
        h, x = _triclinic_box((4, 4, 3), 0.15, 31)
    elif case == "unwrapped":
        h, x = _triclinic_box((3, 4, 4), 0.12, 32)
        x = x + rng.integers(-2, 3, (len(x), 3)).astype(float) @ h      # whole-cell displacements
    else:
        h = np.diag([70.0, 55.0, 61.0])
        x = rng.random((300, 3)) * np.array([40.0, 30.0, 30.0])
    iv = c_oracle.ivects(h)
    onn, ojn, ovn = c_oracle.neighbours(x, iv)
    lists = {}
    for mode in ("grid", "brute"):
        if mode == "brute":
            monkeypatch.setenv("MW_FORCE_BRUTE_NEIGHBOURS", "1")
        else:
            monkeypatch.delenv("MW_FORCE_BRUTE_NEIGHBOURS", raising=False)
        em = load_boxes([h], [x])
        try:
            lists[mode] = em.neighbours(1)
            e = em.model_energy[0]
        finally:
            em.energy_deinit()
        ref = c_oracle.model_energy(x, iv, onn, ojn, ovn)
        assert abs(e - ref) <= RTOL * abs(ref) + 1e-14
    for mode in lists:
        nn, jn, vn = lists[mode]
        assert np.array_equal(nn, onn), mode
        assert np.array_equal(jn, ojn) and np.array_equal(vn, ovn), mode


def test_cell_grid_rows_with_several_candidate_batches_and_ambiguous_hits(monkeypatch):
    """The 1536-molecule Ic / Ih cells have grid cells large enough that a block's candidates come in TWO batches of
    registers (> 320), and among 64 thermal boxes some rows carry a hit too close to the list radius to call in single
    precision.  Such a row's count keeps a flag bit between the batches; round 2's kernel added that bit into the slot
    arithmetic of the second batch, dropped its hits, sorted LDS garbage in their place and -- after a run with other box
    sizes had left other garbage -- took a wild molecule index into the double-precision re-decision (a GPU memory fault,
    found in round 3).  Every list of the 64 boxes against the all-pairs builder, entry for entry."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.energy import load_boxes
    z = [load_golden("ic1536"), load_golden("ih1536")]
    hs, xs = [], []
    for w in range(32):
        for l in range(2):
            hs.append(z[l]["h"]); xs.append(lat.thermalise(z[l]["xyz"], 0.1, 1000 * l + w))
    lists, energies = {}, {}
    for mode in ("grid", "brute"):
        if mode == "brute":
            monkeypatch.setenv("MW_FORCE_BRUTE_NEIGHBOURS", "1")
        else:
            monkeypatch.delenv("MW_FORCE_BRUTE_NEIGHBOURS", raising=False)
        em = load_boxes(hs, xs)
        try:
            lists[mode] = [em.neighbours(b) for b in range(1, 65)]
            energies[mode] = em.model_energy.copy()
        finally:
            em.energy_deinit()
    for b in range(64):
        for a, r in zip(lists["grid"][b], lists["brute"][b]):
            assert np.array_equal(a, r), b
    assert np.allclose(energies["grid"], energies["brute"], rtol=1e-12, atol=0.0)


def test_single_call_local_energy_follows_host_moves(c_oracle):
    """The drop-in protocol of mc_water_translation (mc_moves.F90:1010-1190):
    old energy, move the molecule on the HOST only, new energy, silently revert
    on the host, go on to another molecule -- the engine must track all of it."""
    z = load_golden("ih48_t020")
    em = _engine(z)
    try:
        iv = np.ascontiguousarray(z["ivect"])
        nn, jn, vn = c_oracle.neighbours(z["xyz"], iv)
        xyz = np.array(z["xyz"])
        rng = np.random.default_rng(5)
        for step in range(60):
            imol = int(rng.integers(1, 49))
            e_old = em.compute_local_real_energy(imol, 1)
            assert abs(e_old - c_oracle.local_energy(imol, xyz, iv, nn, jn, vn)) <= RTOL * abs(e_old)
            disp = rng.normal(0, 0.4, 3)
            em.ljr[0, imol - 1] += disp
            xyz[imol - 1] += disp
            e_new = em.compute_local_real_energy(imol, 1)
            assert abs(e_new - c_oracle.local_energy(imol, xyz, iv, nn, jn, vn)) <= RTOL * abs(e_new)
            if step % 2:   # reject: host reverts silently
                em.ljr[0, imol - 1] -= disp
                xyz[imol - 1] -= disp
        # after all that the mirrored box must equal the host's: a full energy WITHOUT re-upload
        e_dev = em.model_energy_batch(1, 1)[0]
        # the last queried molecule may still be stale on the device by design; patch it as the next call would
        e_host = c_oracle.model_energy(xyz, iv, nn, jn, vn)
        last = em._last_imol[0]
        em.compute_local_real_energy(last, 1)
        e_dev = em.model_energy_batch(1, 1)[0]
        assert abs(e_dev - e_host) <= RTOL * abs(e_host)
        assert abs(em.compute_model_energy(1) - e_host) <= RTOL * abs(e_host)
    finally:
        em.energy_deinit()


def test_two_lattices_like_ice1_sample():
    z1, z2 = load_golden("ic48"), load_golden("ih48")
    pair = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "ls_pair48.npz"))
    from mc_water_ls_mw_amd.energy import load_boxes
    em = load_boxes([z1["h"], z2["h"]], [z1["xyz"], z2["xyz"]])
    try:
        assert np.all(np.abs(em.model_energy - pair["model_energy"]) <= RTOL * np.abs(pair["model_energy"]))
        for ils in (1, 2):
            loc = em.local_energy_batch(ils, np.arange(1, 49))
            assert np.all(np.abs(loc - pair["local"][ils - 1]) <= RTOL * np.abs(pair["local"][ils - 1]))
            assert abs(em.compute_local_real_energy(7, ils) - pair["local"][ils - 1][6]) <= RTOL * abs(pair["local"][ils - 1][6])
    finally:
        em.energy_deinit()


def test_neighbour_overflow_fails_loudly():
    from mc_water_ls_mw_amd.energy import MwError, load_boxes
    z = load_golden("ih8_small")   # nn = 25..26
    with pytest.raises(MwError, match="overflow"):
        load_boxes([z["h"]], [z["xyz"]], maxneigh=20)
    from mc_water_ls_mw_amd.energy import load_library
    load_library().mw_finalize()


def test_full_size_properties_4096(c_oracle):
    """Size-independent properties at BASELINE.json's full size (4096 molecules), no oracle energy needed:
    E/N of the ideal crystal equals E/N of a 96-molecule crystal of the same cell (replication invariance);
    sum_i local = 2 E2 + 3 E3 with model = E2 + E3 (G4) holds with E3 > 0; interaction counts of the ideal
    crystal are exactly 4N pairs + 6N triplets; old/new local energies are invariant under a whole-cell
    translation of all molecules (positions are unwrapped, images are not)."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.energy import load_boxes
    h_s, x_s = lat.ice_box("ih", (3, 2, 2), 0.0)
    h_b, x_b = lat.ice_box("ih", (8, 8, 8), 0.0)
    em = load_boxes([h_s], [x_s])
    e_small = em.model_energy[0] / len(x_s)
    em.energy_deinit()
    x_t = lat.thermalise(x_b, 0.15, 99)
    em = load_boxes([h_b, h_b, h_b], [x_b, x_t, x_t + 2 * h_b[0] - h_b[2]])
    try:
        assert em.model_energy[0] / 4096 == pytest.approx(e_small, rel=1e-13)
        assert em.model_energy_counts(1) == (4 * 4096, 6 * 4096)
        assert em.model_energy[2] == pytest.approx(em.model_energy[1], rel=1e-12)      # rigid whole-cell shift
        loc = em.local_energy_batch(2, np.arange(1, 4097))
        e3 = loc.sum() - 2 * em.model_energy[1]
        e2 = em.model_energy[1] - e3
        assert e3 > 0 and e2 < 0 and loc.sum() == pytest.approx(2 * e2 + 3 * e3, rel=1e-13)
        p, t = em.model_energy_counts(2)
        imol = np.arange(1, 4097, 7)
        eo, en = em.delta_energy_batch(2, imol, x_t[imol - 1] + 0.3)
        eo3, en3 = em.delta_energy_batch(3, imol, x_t[imol - 1] + 0.3 + 2 * h_b[0] - h_b[2])
        assert np.allclose(eo, eo3, rtol=1e-11) and np.allclose(en, en3, rtol=1e-11)
        assert np.allclose(eo, loc[imol - 1], rtol=1e-13)
    finally:
        em.energy_deinit()


def test_engine_lifecycle_and_argument_errors():
    from mc_water_ls_mw_amd.energy import EnergyModule, MwError, load_library
    z = load_golden("ic48")
    L = load_library()
    em = EnergyModule(48, 1)
    em.hmatrix[0], em.ljr[0] = z["h"], z["xyz"]
    em.energy_init()
    try:
        assert L.mw_init(0, 48, 1, 50) != 0 and b"already initialised" in L.mw_last_error()
        with pytest.raises(MwError, match="outside"):
            em.compute_local_real_energy(49, 1)
        with pytest.raises(MwError, match="outside"):
            em.compute_model_energy(2)
        with pytest.raises(MwError, match="outside"):
            em.delta_energy_batch(1, np.array([0], dtype=np.int32), np.zeros((1, 3)))
        assert em.local_energy_batch(1, np.zeros(0, dtype=np.int32)).shape == (0,)     # empty batch is fine
    finally:
        em.energy_deinit()
    em.energy_deinit()                                                                 # idempotent
    assert L.mw_is_initialised() == 0


@pytest.mark.parametrize("d_oo", [2.35, 2.2, 2.1])
def test_dense_boxes_take_the_plain_routines(d_oo, c_oracle):
    """Compressed diamond lattices: list rows of up to 35 entries (> the 32 slots of the half-wave pass) and 16-28
    in-range neighbours per molecule (> the 24 records of the wave scratch, > the 12-entry LDS queue of the full-box
    kernel): every fast path has to step aside for its plain fallback, with maxneigh = 64."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.energy import load_boxes
    h, x = lat.ice_ic_cell(d_oo)
    h, x = lat.replicate(h, x, (3, 3, 3))
    x = lat.thermalise(x, 0.05, 1)
    iv = c_oracle.ivects(h)
    nn, jn, vn = c_oracle.neighbours(x, iv, 64)
    e_ref, counts = c_oracle.model_energy(x, iv, nn, jn, vn, counts=True)
    em = load_boxes([h], [x], maxneigh=64)
    try:
        gnn, gjn, gvn = em.neighbours(1)
        assert np.array_equal(gnn, nn) and np.array_equal(gjn, jn) and np.array_equal(gvn, vn)
        assert abs(em.model_energy[0] - e_ref) <= RTOL * abs(e_ref)
        assert em.model_energy_counts(1) == (int(counts[0]), int(counts[1]))
        loc = em.local_energy_batch(1, np.arange(1, len(x) + 1))
        ref = c_oracle.local_energy_all(x, iv, nn, jn, vn)
        assert np.all(np.abs(loc - ref) <= RTOL * np.abs(ref))
        imol, trial = lat.trial_moves(x, 300, max_trans_ang=0.6, seed=8)
        eo, en = em.delta_energy_batch(1, imol, trial)
        ro, rn = c_oracle.trial_moves(imol, trial, x, iv, nn, jn, vn)
        assert np.all(np.abs(eo - ro) <= RTOL * np.abs(ro)) and np.all(np.abs(en - rn) <= RTOL * np.abs(rn))
        assert abs(em.compute_local_real_energy(17, 1) - ref[16]) <= RTOL * abs(ref[16])
    finally:
        em.energy_deinit()


def test_concurrent_host_threads_are_serialised(c_oracle):
    """SURVEY.md 8(b), threading: the reference's dormant OpenMP sections would call compute_local_real_energy for
    both lattices at once (mc_moves.F90:1006-1018).  Two host threads hammer the two boxes of an Ic/Ih pair through
    the single-call entry (ctypes releases the GIL during a call): every value must be the serial one."""
    import threading
    from mc_water_ls_mw_amd.energy import load_boxes
    z1, z2 = load_golden("ic48_t015"), load_golden("ih48_t020")
    em = load_boxes([z1["h"], z2["h"]], [z1["xyz"], z2["xyz"]])
    try:
        want = [np.array([em.compute_local_real_energy(i, ils) for i in range(1, 49)]) for ils in (1, 2)]
        got = [np.zeros((6, 48)), np.zeros((6, 48))]
        errors = []

        def worker(ils):
            try:
                for rep in range(6):
                    for i in range(1, 49):
                        got[ils - 1][rep, i - 1] = em.compute_local_real_energy(i, ils)
            except Exception as exc:          # noqa: BLE001
                errors.append(exc)

        threads = [threading.Thread(target=worker, args=(ils,)) for ils in (1, 2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors
        for ils in (0, 1):
            assert np.array_equal(got[ils], np.broadcast_to(want[ils], (6, 48)))
    finally:
        em.energy_deinit()


_SINGLE_CALL_SCRIPT = """
import json, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from conftest import load_golden
from mc_water_ls_mw_amd.energy import load_boxes
z1, z2 = load_golden("ic48_t015"), load_golden("ih48_t020")
em = load_boxes([z1["h"], z2["h"]], [z1["xyz"], z2["xyz"]])
rng = np.random.default_rng(5)
out = []
for step in range(300):                       # the caller's pattern: move a molecule on the host, ask, sometimes put it back
    ils, i = int(rng.integers(1, 3)), int(rng.integers(1, 49))
    old = em.ljr[ils - 1, i - 1].copy()
    em.ljr[ils - 1, i - 1] += rng.normal(0, 0.3, 3)
    out.append(em.compute_local_real_energy(i, ils))
    if step % 3 == 0:
        em.ljr[ils - 1, i - 1] = old          # silent revert (mc_moves.F90:1186)
    if step % 50 == 49:
        em.compute_model_energy(ils)          # an exclusive entry: stops and restarts the resident server
        out.append(float(em.model_energy[ils - 1]))
em.energy_deinit()
print(json.dumps([float(v).hex() for v in out]))
"""


def test_single_call_paths_agree(tmp_path):
    """The drop-in single call has three transports -- request lines in device memory (default where the host can address
    it), request lines in host-mapped memory (MW_SERVER_REQ=host), one kernel launch per call (MW_LOCAL_SERVER=0).  The
    same sequence of host-side moves, reverts and calls must return the same bits through the two server transports and
    the same numbers (another summation order, 1e-12) through the launch path."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "calls.py"
    script.write_text(_SINGLE_CALL_SCRIPT.format(root=ROOT, tests=os.path.join(ROOT, "tests")))
    results = {}
    for name, env in (("device", {"MW_SERVER_REQ": "device"}), ("host", {"MW_SERVER_REQ": "host"}), ("launch", {"MW_LOCAL_SERVER": "0"})):
        out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert out.returncode == 0, out.stderr[-2000:]
        results[name] = json.loads(out.stdout.strip().splitlines()[-1])
    assert len(results["device"]) == 306
    assert results["device"] == results["host"]                          # the same kernel behind both transports: the same bits
    dev = np.array([float.fromhex(v) for v in results["device"]])
    lau = np.array([float.fromhex(v) for v in results["launch"]])         # another evaluation order (plain routine): last-bit differences
    assert np.all(np.abs(dev - lau) <= 1e-12 * np.abs(dev))


@pytest.mark.parametrize("moments", ["1", "0"])
def test_moment_path_of_the_move_kernel_and_the_triplets_it_must_decline(moments, c_oracle, monkeypatch):
    """The batched single-move kernel's MOMENT path (i--j--k triplets from the per-molecule moments the full-box pass leaves behind,
    mw_move_energy.hip.h) against the oracle, next to the scanning path (MW_MOVE_MOMENTS=0) on the same requests: a 1536-molecule
    Ic box, thermal -- and with one molecule pushed ALMOST BEHIND a neighbour's neighbour, so that triplets with cos(theta) >= 0.99
    exist: the reference drops those terms (molint.F90:367-371), a moment sum cannot, and the requests that touch them must come
    out right all the same (declined to the plain routine).  Energies to 1e-10, interaction and slot counts exact."""
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.energy import load_boxes
    monkeypatch.setenv("MW_MOVE_MOMENTS", moments)
    z = load_golden("ic1536")
    h = z["h"]
    x = lat.thermalise(z["xyz"], 0.1, 77)
    iv = c_oracle.ivects(h)
    nn, jn, vn = c_oracle.neighbours(x, iv)
    # molecule a, its nearest neighbour b (central image), and a third molecule c placed on the line a -> b, 1.45 |ab| from a
    a = 100
    cands = [(np.linalg.norm(x[jn[a, s] - 1] - x[a]), jn[a, s] - 1) for s in range(nn[a]) if vn[a, s] == 1]
    dab, b = min(cands)
    c = next(k for k in range(len(x)) if k not in (a, b) and np.linalg.norm(x[k] - x[a]) > 15.0)
    x = x.copy()
    x[c] = x[a] + 1.45 * (x[b] - x[a]) + np.array([0.02, -0.01, 0.015])
    nn, jn, vn = c_oracle.neighbours(x, iv)
    em = load_boxes([h], [x])
    try:
        rng = np.random.default_rng(5)
        imol = np.concatenate([np.array([a + 1, b + 1, c + 1] * 3), rng.integers(1, len(x) + 1, 1600)]).astype(np.int32)
        trial = x[imol - 1] + rng.normal(0.0, 0.4, (len(imol), 3))
        trial[:3] = x[imol[:3] - 1]                                       # (unmoved: old == new, the dropped terms on both sides)
        eo, en = em.delta_energy_batch(1, imol, trial)
        ro, rn = c_oracle.trial_moves(imol, trial, x, iv, nn, jn, vn)
        assert np.all(np.abs(eo - ro) <= RTOL * np.abs(ro)) and np.all(np.abs((en - eo) - (rn - ro)) <= DE_ATOL)
        # the pushed molecule really does make such triplets: with the 0.99 rule ignored, b's local energy would be off by far more than the tolerance
        _, counts_b = c_oracle.local_energy(b + 1, x, iv, nn, jn, vn, counts=True)
        io, so, inw, sn = em.moves_counts()
        ref_i = sum(int(c_oracle.local_energy(int(i), x, iv, nn, jn, vn, counts=True)[1].sum()) for i in imol[:64])
        eo64, _ = em.delta_energy_batch(1, imol[:64], x[imol[:64] - 1])
        assert em.moves_counts()[0] == ref_i and em.moves_counts()[2] == ref_i          # old = new: the same interactions, counted exactly
        assert np.all(np.abs(eo64 - ro[:64]) <= RTOL * np.abs(ro[:64]))
    finally:
        em.energy_deinit()
