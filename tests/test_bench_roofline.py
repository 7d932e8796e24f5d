"""CPU: bench.py's roofline block (the contract of the measurement section): the HBM-convention fraction next to what the
committed counter summary (profiles/counters.json, written by tools/summarize_profile.py) says binds the kernel."""
import importlib.util
import json
import os
import sys

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_counter_summary_is_the_one_the_docs_quote():
    c = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert c["tag"] == t["tag"] and c["walkers"] == 512 and c["moves"] == 2048        # the default bench workload
    for f in ("kernel_stats.csv", "pmc_counters.txt", "summary.md"):
        assert os.path.exists(os.path.join(ROOT, "profiles", f"{c['tag']}_{f}"))
    assert 1500.0 < c["cycles_per_us"] < 2500.0                                       # a shader clock, not a unit mix-up


@pytest.mark.parametrize("kernel,alg_bytes,ms,units", [("k_model_energy", 387006480, 0.092, 512 * 4096),
                                                       ("k_move_energy", 10867775200, 1.26, 512 * 2048)])
def test_roofline_block(bench, kernel, alg_bytes, ms, units):
    c = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    r = bench.kernel_roofline(kernel, alg_bytes, ms, units, c, 5500.0)
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] == pytest.approx(alg_bytes / (ms * 1e-3) / 1e9) and r["frac"] == pytest.approx(r["achieved"] / 8000.0)
    assert r["bound"] == "valu" and 0.5 < r["valu"]["busy"] < 1.0 and 0.3 < r["lds_busy"] < r["valu"]["busy"]
    assert 0.0 < r["hbm_traffic_frac"] < 0.5                       # neither kernel moves anything like its algorithmic bytes
    assert r["traffic"] < alg_bytes and r["counters_tag"] == c["tag"]
    assert r["valu"]["wave_insts_per_s"] < r["valu"]["peak_wave_insts_per_s_f64"]


def test_roofline_without_counters_does_not_claim_hbm(bench):
    r = bench.kernel_roofline("k_move_energy", 10867775200, 1.26, 512 * 2048, {}, 5500.0)
    assert r["traffic"] is None and r["bound"].startswith("not hbm")        # 8.6 TB/s algorithmic > the box's copy ceiling
    r = bench.kernel_roofline("k_model_energy", 387006480, 0.092, 512 * 4096, {}, 5500.0)
    assert r["bound"].startswith("unknown")
