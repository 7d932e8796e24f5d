#!/usr/bin/env python3
"""Latency of the drop-in single calls through the C ABI on a 4096-molecule box (host wall time per synchronous
call, ctypes inclusive): compute_local_real_energy behind the resident server (default) and behind a kernel launch
(MW_LOCAL_SERVER=0, measured in a child process), compute_model_energy, compute_neighbours.
(The other configurations of BASELINE.json -- 1536 pairs, 32768 boxes -- are bench.py's `secondary` list.)
Prints one JSON object.  Run on the GPU box:  python tools/extra_measurements.py
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure():
    import torch  # noqa: F401  (one HIP runtime per process: torch first)
    from mc_water_ls_mw_amd import lattice as lat
    from mc_water_ls_mw_amd.energy import load_boxes
    h, x = lat.ice_box("ih", (8, 8, 8), 0.15, seed=20250228)
    em = load_boxes([h], [x])
    rng = np.random.default_rng(1)
    mols = [int(m) for m in rng.integers(1, 4097, 8000)]
    for m in mols[:3000]:                       # (also brings the GPU out of its idle clock)
        em.compute_local_real_energy(m, 1)
    t0 = time.perf_counter()
    for m in mols[3000:]:
        em.compute_local_real_energy(m, 1)
    t_loc = (time.perf_counter() - t0) / 5000
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        em.compute_model_energy(1)
    t_full = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(50):
        em.compute_neighbours(1)
    t_nb = (time.perf_counter() - t0) / 50
    em.energy_deinit()
    return {"compute_local_real_energy_us": t_loc * 1e6, "compute_model_energy_us (incl. 96 KiB H2D)": t_full * 1e6,
            "compute_neighbours_us (incl. H2D)": t_nb * 1e6}


if __name__ == "__main__":
    if "--child" in sys.argv:
        print(json.dumps(measure()))
        sys.exit(0)
    out = {"single_call_latency_4096": dict(measure(), note="host wall time per synchronous call through ctypes -> C ABI; "
                                            "local energy: request posted to the resident server kernel's mail slot")}
    child = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True,
                           env=dict(os.environ, MW_LOCAL_SERVER="0"), timeout=600)
    if child.returncode == 0:
        out["single_call_latency_4096_launch_path"] = dict(json.loads(child.stdout.strip().splitlines()[-1]),
                                                            note="MW_LOCAL_SERVER=0: one kernel launch + completion word per local-energy call")
    else:
        out["single_call_latency_4096_launch_path"] = {"error": child.stderr[-500:]}
    print(json.dumps(out, indent=1))
