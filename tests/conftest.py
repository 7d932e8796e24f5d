import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# mw_sweep_sync_cells no longer re-uploads the image vectors the device's volume moves built: under test, every call reads the
# device's tables back and fails unless they equal the host's own compute_ivects bit for bit (read once, at the first call).
os.environ.setdefault("MW_SYNC_CELLS_VERIFY", "1")

#: parity bar of BASELINE.json's north_star: energies within 1e-10 relative of the Fortran reference
RTOL = 1e-10
#: the reference author's own absolute tolerance on a move's energy change (mc_moves.F90:1099), Hartree
DE_ATOL = 1e-10


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                  if os.path.basename(p) not in ("constants.npz", "ls_pair48.npz") and not os.path.basename(p).startswith("chain_"))   # (chain_*: the driver's golden chains, tests/test_chain_fixtures.py)


def load_golden(name):
    from mc_water_ls_mw_amd import lattice as lat
    z = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    if "xyz" not in z:   # big synthetic boxes are regenerated from their recipe and checked by digest
        kind = name[:2]
        reps = {"4096": (8, 8, 8), "32768": (16, 16, 16)}[name[2:].split("_")[0]]
        sigma = 0.15 if name.endswith("t015") else 0.0
        h, xyz = lat.ice_box(kind, reps, sigma, seed=20250228)
        import hashlib
        assert hashlib.sha256(np.ascontiguousarray(xyz).tobytes()).hexdigest() == str(z["xyz_sha256"]), \
            f"{name}: regenerated positions differ from the ones the golden vector was made from"
        assert np.array_equal(h, z["h"])
        z["xyz"] = xyz
    return z


def list_digest(nn, jn, vn):
    import hashlib
    mask = np.arange(jn.shape[1])[None, :] < nn[:, None]
    m = hashlib.sha256()
    for a in (nn.astype(np.int32), jn[mask].astype(np.int32), vn[mask].astype(np.int32)):
        m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import COracle
    return COracle()
