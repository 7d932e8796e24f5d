#!/usr/bin/env python3
"""Wall time of the WHOLE unmodified reference program with its own energy module (oracle/_ref/mc_water_ref) and
with the engine linked in its place (oracle/_ref/mc_water_hip), on the inputs of examples/single_box and
examples/ice1_sample (48-molecule cells; namelists as in tests/test_gpu_full_program.py with more cycles).
Run on the GPU box:  python tools/program_walltime.py [cycles]   -> one JSON object."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_full_program as fp  # noqa: E402  (namelists + input preparation)


def timed(binary, d):
    t0 = time.perf_counter()
    out = subprocess.run([binary, "ice.input"], cwd=d, capture_output=True, text=True, timeout=3000)
    dt = time.perf_counter() - t0
    if out.returncode != 0:
        raise SystemExit(out.stderr[-2000:])
    return dt


def main():
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    res = {"cycles": cycles, "molecules": 48,
           "note": "wall seconds of the whole program incl. start-up (10^6-sample RNG self test, input parsing, "
                   "device initialisation for the engine build); one translation move = 4 compute_local_real_energy calls"}
    for name, nl, two in (("single_box (NPT, 1 lattice)", fp.SINGLE_BOX, False), ("ice1_sample (lattice switch, 2 lattices)", fp.LATTICE_SWITCH, True)):
        nl = nl.replace("max_mc_cycles    = 600", f"max_mc_cycles    = {cycles}").replace("max_mc_cycles    = 400", f"max_mc_cycles    = {cycles}")
        row = {}
        for tag, binary in (("reference", fp.RAW), ("reference_with_stack_scrub", fp.REF), ("engine", fp.HIP)):
            if not os.path.exists(binary):
                continue
            with tempfile.TemporaryDirectory() as td:
                d = os.path.join(td, "run")
                fp._prepare(d, nl, two)
                timed(binary, d)                      # warm (page cache, first HIP initialisation)
            with tempfile.TemporaryDirectory() as td:
                d = os.path.join(td, "run")
                fp._prepare(d, nl, two)
                row[tag + "_s"] = timed(binary, d)
        if "engine_s" in row and "reference_s" in row:
            moves = cycles * 48
            row["engine_over_reference"] = row["engine_s"] / row["reference_s"]
            row["engine_us_per_move"] = row["engine_s"] / moves * 1e6
            row["reference_us_per_move"] = row["reference_s"] / moves * 1e6
        res[name] = row
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
