#!/usr/bin/env python3
"""Condense gpurun_out/sweepprof_<tag>/ (tools/profile_sweep.sh) into profiles/<tag>_sweep_pmc_counters.txt and
profiles/<tag>_sweep_counters.json: per case the rocprofv3 average duration of the timed k_sweep_translation launch,
the raw SQ / TCC counters of that launch (separate --pmc passes) and what follows from them --
VALU instructions per move, FP64-VALU issue busy (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x launch cycles)), LDS pipe busy,
bank-conflict share, issue-stall share, HBM bytes per move ((2 FETCH_SIZE + WRITE_SIZE) x 1024, MI355X_MICROARCH.md "HBM").

The timed launch is the LONGEST k_sweep_translation dispatch of the run (the 20-move warm-up launch is the other one)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"sweepprof_{tag}")
dst = os.path.join(root, "profiles")


def gpu_agent(src_dir):
    """XCDs, CUs and SIMDs of the profiled GPU from rocprofv3's agent table (8 / 256 / 1024 on an MI355X)."""
    for root_, _, files in os.walk(src_dir):
        for f in files:
            if f.endswith("agent_info.csv"):
                for r in csv.DictReader(open(os.path.join(root_, f))):
                    if r.get("Agent_Type", r.get("Type", "")).upper().find("GPU") >= 0 and int(r.get("Cu_Count", 0) or 0) > 0:
                        return int(r.get("Num_Xcc", 8) or 8), int(r["Cu_Count"]), int(r.get("Simd_Count", 0) or 1024)
    return 8, 256, 1024


NXCD, NCU, NSIMD = gpu_agent(src)


def longest_sweep_dispatch(path):
    """{counter: value} and duration (us) of the longest k_sweep dispatch in one counter_collection.csv."""
    per = collections.defaultdict(dict)
    dur, name = {}, {}
    for r in csv.DictReader(open(path)):
        if "k_sweep" not in r["Kernel_Name"]:
            continue
        d = r["Dispatch_Id"]
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name[d] = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[d]["VGPR_Count"] = float(r.get("VGPR_Count") or r.get("Arch_VGPR_Count") or 0)
        per[d]["LDS_Block_Size"] = float(r.get("LDS_Block_Size") or 0)
    if not dur:
        return None, None, None
    d = max(dur, key=dur.get)
    return per[d], dur[d], name[d]


out, lines = {}, [f"# k_sweep_translation counters `{tag}` (tools/profile_sweep.sh: rocprofv3, one --pmc pass per counter group)", ""]
for case in sorted(os.listdir(src)):
    cdir = os.path.join(src, case)
    if not os.path.isdir(cdir):
        continue
    try:
        run = json.loads(open(os.path.join(cdir, "run_trace.json")).read())
    except Exception as e:                                   # noqa: BLE001
        lines.append(f"## {case}: run failed ({e})")
        continue
    res = list(run.values())[0]
    moves = res["walkers"] * res["moves_per_walker"]
    rec = {"case": list(run.keys())[0], **{k: res[k] for k in ("walkers", "lattices", "molecules", "moves_per_walker", "ms", "moves_per_s",
                                                               "local_energy_evaluations_per_s")}}
    st = glob.glob(os.path.join(cdir, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        for r in csv.DictReader(open(st[0])):
            if "k_sweep" in r["Name"]:
                rec["kernel"] = r["Name"].split("(")[0].replace("void ", "")
                rec["rocprof_calls"] = int(r["Calls"])
                rec["rocprof_max_us"] = float(r["MaxNs"]) / 1e3
    ctr = {}
    for p in ("sq1", "sq2", "fetch", "write"):
        f = glob.glob(os.path.join(cdir, p, "**", "*counter_collection.csv"), recursive=True)
        if not f:
            continue
        c, d, nm = longest_sweep_dispatch(f[0])
        if c is None:
            continue
        ctr.update({k: v for k, v in c.items()})
        ctr[f"dur_us_{p}"] = d
        rec.setdefault("kernel", nm)
    rec["counters"] = ctr
    der = {}
    if "SQ_INSTS_VALU" in ctr:
        der["valu_insts_per_move"] = ctr["SQ_INSTS_VALU"] / moves
        der["salu_insts_per_move"] = ctr.get("SQ_INSTS_SALU", 0) / moves
        der["lds_insts_per_move"] = ctr.get("SQ_INSTS_LDS", 0) / moves
    if "GRBM_GUI_ACTIVE" in ctr and "dur_us_sq2" in ctr:
        clk = ctr["GRBM_GUI_ACTIVE"] / NXCD / ctr["dur_us_sq2"]          # shader cycles per us (long launch: reads true)
        der["clock_MHz"] = clk
        if "SQ_ACTIVE_INST_VALU" in ctr:
            der["valu_busy"] = ctr["SQ_ACTIVE_INST_VALU"] * 4.0 / (NSIMD * ctr["dur_us_sq1"] * clk)
        if "SQ_LDS_IDX_ACTIVE" in ctr:
            der["lds_busy"] = ctr["SQ_LDS_IDX_ACTIVE"] / (NCU * ctr["dur_us_sq2"] * clk)
            der["lds_bank_conflict_share"] = ctr["SQ_LDS_BANK_CONFLICT"] / max(ctr["SQ_LDS_IDX_ACTIVE"], 1.0)
        if "SQ_WAVE_CYCLES" in ctr:
            der["waves_per_simd_avg"] = ctr["SQ_WAVE_CYCLES"] * 4.0 / (NSIMD * ctr["dur_us_sq1"] * clk)
    if "SQ_WAVE_CYCLES" in ctr:
        der["wait_inst_any_share"] = ctr["SQ_WAIT_INST_ANY"] / ctr["SQ_WAVE_CYCLES"]
        der["wait_any_share"] = ctr["SQ_WAIT_ANY"] / ctr["SQ_WAVE_CYCLES"]
        der["active_valu_share_of_wave_cycles"] = ctr["SQ_ACTIVE_INST_VALU"] / ctr["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
        hb = (2 * ctr["FETCH_SIZE"] + ctr["WRITE_SIZE"]) * 1024
        der["hbm_bytes_per_move"] = hb / moves
        der["hbm_GBps"] = hb / (ctr["dur_us_fetch"] * 1e-6) / 1e9
    rec["derived"] = der
    out[case] = rec
    lines.append(f"## {case}: {rec['case']}  ({rec.get('kernel', '?')})")
    lines.append(f"walkers {rec['walkers']} x {rec['lattices']} lattices x {rec['molecules']} molecules, {rec['moves_per_walker']} moves per walker; "
                 f"un-profiled-pass rate {rec['moves_per_s']:.4g} moves/s = {rec['local_energy_evaluations_per_s']:.4g} local-energy evaluations/s ({rec['ms']:.3f} ms)")
    for k, v in sorted(ctr.items()):
        lines.append(f"  {k:28s} {v:.6g}")
    for k, v in der.items():
        lines.append(f"  -> {k:34s} {v:.4g}")
    lines.append("")
os.makedirs(dst, exist_ok=True)
for d in (dst, src):          # profiles/ (tracked) and the scratch directory gpurun merges back
    open(os.path.join(d, f"{tag}_sweep_pmc_counters.txt"), "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join(d, f"{tag}_sweep_counters.json"), "w"), indent=1)
print("\n".join(lines))
