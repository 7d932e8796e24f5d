"""GPU: bench.py's self-check of the production driver -- walker 0's chain on the device against the oracle's committed chains
(tests/golden/chain_*.npz), with one move at a time and with the look-ahead a handful of walkers gets -- and the roofline
bookkeeping of its `production_driver` section."""
import importlib.util
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", ["chain_farm48_nvt", "chain_farm48_npt", "chain_ih4096"])
def test_walker0_follows_the_oracles_chain(name):
    r = _bench().chain_self_check(0, name)
    assert r is not None and r["moves_checked"] >= 512 and r["max_rel_err_energies"] <= 1e-10
    assert r["accepted"] > 20


def test_sweep_roofline_needs_counters_of_the_case():
    b = _bench()
    c = b.load_sweep_counters()
    assert "farm48_nvt" in c and c["farm48_nvt"]["SQ_INSTS_VALU_per_move"] > 500
    r = b.sweep_roofline("farm48_nvt", 1000000, 5.0, c)
    assert r["bound"] == "valu" and 0.0 < r["frac"] < 1.0
    r = b.sweep_roofline("no such case", 1000000, 5.0, c)
    assert r["frac"] is None and r["bound"].startswith("unknown")
