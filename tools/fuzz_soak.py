#!/usr/bin/env python3
"""More seeds for tests/test_gpu_fuzz.py than the suite runs (static paths and the Monte Carlo driver against the oracle).
Run on the GPU box: python tools/fuzz_soak.py [first_seed] [count]; prints the seeds that failed (none: "all clean")."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa: E402
import torch  # noqa: E402,F401

import test_gpu_fuzz as fz  # noqa: E402
from oracle import COracle  # noqa: E402

import test_sweep as ts  # noqa: E402
from oracle import SweepOracle  # noqa: E402

class _Env:                      # what the tests use of pytest's monkeypatch
    @staticmethod
    def setenv(k, v):
        os.environ[k] = v


first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
c = COracle()
so = SweepOracle()
bad, skipped = [], 0
for seed in range(first, first + count):
    for name in ("test_random_systems_follow_the_oracle", "test_random_systems_through_the_monte_carlo_driver",
                 "test_npt_driver_on_random_lattice_pairs", "test_driver_run_options_on_random_pairs",
                 "test_lookahead_with_random_run_options", "test_driver_window_decomposition_on_random_pairs"):
        try:
            if name.startswith("test_npt"):
                ts.test_npt_driver_on_random_lattice_pairs(so, c, seed)          # (volume moves on random Ic / Ih pairs)
            elif name.startswith("test_lookahead"):
                ts.test_lookahead_with_random_run_options(seed, _Env)            # (look-ahead 2 / 4 = the sequential chain)
                os.environ.pop("MW_SWEEP_AHEAD", None)
            elif name.startswith("test_driver_window"):
                ts.test_driver_window_decomposition_on_random_pairs(so, c, seed)  # ('dd' windows, equilibration period, its flag)
            elif name.startswith("test_driver"):
                ts.test_driver_run_options_on_random_pairs(so, c, seed)          # (the run options of mc_cycle in random combination)
            else:
                getattr(fz, name)(seed, c)
        except pytest.skip.Exception:
            skipped += 1
        except Exception:                                        # noqa: BLE001
            bad.append((seed, name))
            traceback.print_exc(limit=3)
    if (seed - first) % 25 == 24:
        print(f"... {seed - first + 1} seeds, {len(bad)} failures, {skipped} skipped", flush=True)
print("all clean" if not bad else f"FAILED: {bad}")
sys.exit(1 if bad else 0)
