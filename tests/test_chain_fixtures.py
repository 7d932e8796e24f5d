"""CPU: the golden chains bench.py holds the Monte Carlo driver to (tests/golden/chain_*.npz) ARE the oracle's: regenerated here with
tests/golden/make_chain_fixtures.py's own recipe, they come out move for move -- same molecule, same outcome, same energies."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden")


def _maker():
    spec = importlib.util.spec_from_file_location("make_chain_fixtures", os.path.join(GOLD, "make_chain_fixtures.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", ["chain_farm48_nvt", "chain_farm48_npt", "chain_ih4096"])
def test_chain_fixture_is_the_oracles_chain(name):
    m = _maker()
    g = np.load(os.path.join(GOLD, name + ".npz"))
    d = m.ih4096(int(g["nmoves"])) if name == "chain_ih4096" else m.farm48(name.endswith("npt"), int(g["nmoves"]))
    assert np.array_equal(d["log"][:, :2], g["log"][:, :2])                        # molecule, outcome flags
    assert np.allclose(d["log"][:, 2:], g["log"][:, 2:], rtol=1e-13, atol=1e-13)   # (the same C code on the same inputs: rounding of libm at most)
    flags = g["log"][:, 1].astype(int)
    assert 20 < (flags & 1).sum() < len(flags) - 20                                # moves are accepted and rejected
    if name.endswith("npt"):
        assert ((flags >> 2) & 1).sum() > 5                                        # volume moves take part
