#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (tools/profile.sh) into profiles/<tag>_*:
the rocprofv3 --kernel-trace --stats table, per-kernel FETCH_SIZE / WRITE_SIZE
(separate --pmc passes) and the bench line of the traced run.

HBM bytes per launch follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads
by exactly 2x, so  traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (an upper-side
estimate for narrower accesses, which are uncalibrated)."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("fetch", "write", "sq1", "sq2"):
    path = os.path.join(src, f"pmc_{name}", f"{name}_counter_collection.csv")
    if not os.path.exists(path):
        continue
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if name in ("sq1", "sq2") and r["Dispatch_Id"] not in seen:     # the launch's duration IN THIS PASS (counters slow it)
            seen.add(r["Dispatch_Id"])
            pmc[k][f"dur_us_{name}"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
W, M = bench["config"]["walkers_per_gpu"], bench["config"]["moves_per_walker"]
traffic = {"walkers": W, "moves": M, "tag": tag}
lines = [f"# rocprofv3 summary `{tag}`", "",
         f"Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps {bench['steps']} --warmup {bench['warmup']}`"
         f" (walkers {W}, moves/walker {M}); PMC in separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs.", "",
         "| kernel | calls | avg us (rocprof) | avg us (bench HIP events) | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes/launch (2*F+W)*1024 |",
         "|---|---|---|---|---|---|---|"]
stats = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv")))}
for k, r in stats.items():
    if not k.startswith("mw::"):
        continue
    f = pmc.get(k, {}).get("FETCH_SIZE", [])
    w = pmc.get(k, {}).get("WRITE_SIZE", [])
    fa = sum(f) / len(f) if f else None
    wa = sum(w) / len(w) if w else None
    hb = (2 * fa + wa) * 1024 if fa is not None and wa is not None else None
    short = k.replace("mw::", "").split("<")[0]
    if short == "k_model_energy" and k.rstrip().endswith("true>"):
        short = "k_model_energy_moments"          # the step's full-box launch: the evaluation + every molecule's moments (MOMOUT build)
    ev = bench["kernels"].get(short, {}).get("avg_ms")
    if short == "k_model_energy":
        ev = bench["kernels"]["k_model_energy"].get("plain_avg_ms", ev)
    elif short == "k_model_energy_moments":
        ev = bench["kernels"]["k_model_energy"].get("avg_ms")
    if hb is not None and short in ("k_model_energy", "k_move_energy", "k_model_energy_moments"):
        if short != "k_move_energy" or hb > traffic.get(short, 0.0):      # (several builds of the move kernel: the one that carries the launch)
            traffic[short] = hb
    lines.append(f"| {k} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {'' if ev is None else f'{ev*1e3:.1f}'} | "
                 f"{'' if fa is None else f'{fa:.0f}'} | {'' if wa is None else f'{wa:.0f}'} | {'' if hb is None else f'{hb:.4g}'} |")
lines += ["", "Bench line of the traced run:", "", "```json", json.dumps(bench), "```", ""]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines))
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"))

# raw counters per kernel (means over the launches) for bench.py's roofline block: profiles/counters.json
counters = {"walkers": W, "moves": M, "tag": tag,
            "source": "rocprofv3 --pmc, separate passes (tools/profile.sh); FETCH_SIZE / WRITE_SIZE in KiB, SQ_* as reported"}
sq_lines = [f"# SQ / GRBM / TCC counters per kernel, means over the launches of `{tag}` (tools/profile.sh)"]
for k, cs in sorted(pmc.items()):
    if not k.startswith("mw::"):
        continue
    short = k.replace("mw::", "").split("<")[0]
    if short == "k_model_energy" and k.rstrip().endswith("true>"):
        short = "k_model_energy_moments"
    rec = {c: sum(v) / len(v) for c, v in cs.items()}
    if k in stats:
        rec["avg_us"] = float(stats[k]["AverageNs"]) / 1e3
    rec["instantiation"] = k
    if short in ("k_model_energy", "k_move_energy", "k_model_energy_moments"):
        if short not in counters or rec.get("SQ_INSTS_VALU", 0.0) > counters[short].get("SQ_INSTS_VALU", 0.0):
            counters[short] = rec
    for c, v in sorted(rec.items()):
        if c != "instantiation":
            sq_lines.append(f"{k:48s} {c:24s} {v:.6g}")
# shader cycles per microsecond, from the long kernel (GRBM_GUI_ACTIVE sums the 8 XCDs; both in the sq2 pass).  Cycle counts
# of a kernel in a pass = its duration in that pass x this clock: GRBM_GUI_ACTIVE itself over-counts short kernels (a
# persistent k_model_energy launch reads 25 % more "cycles" than its duration allows at the 2.4 GHz ceiling).
mv = counters.get("k_move_energy", {})


def gpu_agent(src_dir):
    """XCDs, CUs and SIMDs of the profiled GPU from rocprofv3's agent table (8 / 256 / 1024 on an MI355X)."""
    for root_, _, files in os.walk(src_dir):
        for f in files:
            if f.endswith("agent_info.csv"):
                for r in csv.DictReader(open(os.path.join(root_, f))):
                    if r.get("Agent_Type", r.get("Type", "")).upper().find("GPU") >= 0 and int(r.get("Cu_Count", 0) or 0) > 0:
                        return int(r.get("Num_Xcc", 8) or 8), int(r["Cu_Count"]), int(r.get("Simd_Count", 0) or 0)
    return 8, 256, 1024


counters["xcds"], counters["compute_units"], counters["simds"] = gpu_agent(src)
if "GRBM_GUI_ACTIVE" in mv and "dur_us_sq2" in mv:
    counters["cycles_per_us"] = mv["GRBM_GUI_ACTIVE"] / counters["xcds"] / mv["dur_us_sq2"]
json.dump(counters, open(os.path.join(dst, "counters.json"), "w"), indent=1)
open(os.path.join(dst, f"{tag}_pmc_counters.txt"), "w").write("\n".join(sq_lines) + "\n")
print("\n".join(lines[:12]))
