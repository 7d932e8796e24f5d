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
    """`frac` is measured against the ceiling `bound` names (FP64 VALU issue for both kernels) and cannot exceed 1; the
    SURVEY.md 8(d) convention -- which does exceed 1 for the move kernel -- is carried beside it under its own name."""
    c = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    r = bench.kernel_roofline(kernel, alg_bytes, ms, units, c, 5500.0)
    assert r["bound"] == "valu" and r["unit"].startswith("G wave64-instructions/s")
    assert r["peak"] == pytest.approx(1024 * 2.4e9 / 4 / 1e9)
    assert r["achieved"] == pytest.approx(c[kernel]["SQ_INSTS_VALU"] / (ms * 1e-3) / 1e9)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.4 < r["frac"] <= 1.0
    assert r["algorithmic_GBps"] == pytest.approx(alg_bytes / (ms * 1e-3) / 1e9)
    assert r["convention_frac"] == pytest.approx(r["algorithmic_GBps"] / 8000.0)
    v, l = r["valu"], r["lds"]
    assert 0.5 < v["busy_duration"] < 1.0 and 0.3 < v["busy_grbm"] <= v["busy_duration"] * 1.05      # both cycle bases are printed
    assert 0.2 < l["busy_duration"] < v["busy_duration"] and 0.0 <= l["bank_conflict_share"] < 1.0
    assert 0.0 < r["hbm_traffic_frac"] < 0.6                       # neither kernel moves anything like its algorithmic bytes
    assert r["traffic"] < alg_bytes and r["counters_tag"] == c["tag"]


def test_roofline_without_counters_claims_no_ceiling(bench):
    r = bench.kernel_roofline("k_move_energy", 10867775200, 1.26, 512 * 2048, {}, 5500.0)
    assert r["traffic"] is None and r["frac"] is None and r["bound"].startswith("unknown") and "not hbm" in r["bound"]
    assert r["convention_frac"] > 1.0                              # 8.6 TB/s algorithmic: the convention is not a roofline
    r = bench.kernel_roofline("k_model_energy", 387006480, 0.092, 512 * 4096, {}, 5500.0)
    assert r["bound"].startswith("unknown") and "not hbm" not in r["bound"] and r["frac"] is None


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """`--gpus N` with a launcher environment of another size is an error, never a silent 1-GPU run (and with no launcher
    environment `--gpus N > 1` starts its own ranks: tests/test_gpu_multiprocess.py)."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and p.stdout.strip() == ""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_self_launched_ranks_that_fail_are_reported_at_once():
    """`python bench.py --gpus 2` on a box without an MI355X: both children refuse to run (no CPU fallback), and the parent says
    which ranks failed and returns their status instead of waiting for a line that never comes."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box WITHOUT a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and p.stdout.strip() == ""
    # (both refuse; whichever exits first has the parent end the other one: [1, 1], [1, -15] or [-15, 1])
    import re
    m = re.search(r"rank exit codes \[(-?\d+), (-?\d+)\]", p.stderr)
    assert m and 1 in (int(m.group(1)), int(m.group(2))) and 0 not in (int(m.group(1)), int(m.group(2)))
    assert p.stderr.count("no CPU fallback") >= 1
