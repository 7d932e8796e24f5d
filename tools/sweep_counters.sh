#!/bin/bash
# profiles/sweep_counters.json for bench.py's production_driver section: SQ_INSTS_VALU per move of k_sweep (exact for a given
# chain) and HBM bytes per move, per workload, from separate rocprofv3 --pmc passes over tools/sweep_measurements.py.
# usage: tools/sweep_counters.sh <tag>      (on the GPU box; copy gpurun_out/sweep_counters.json to profiles/ afterwards)
TAG=${1:-r04}
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$REPO
OUT=$REPO/gpurun_out/sweepctr_$TAG
mkdir -p $OUT
cd /tmp
for pair in farm48_nvt:farm48 farm48_npt:farm48npt chain48_nvt_1:one48 chain48_npt_1:one48npt chain48_nvt_8:eight48 chain48_npt_8:eight48npt ih4096_2048:ih4096; do
  name=${pair%%:*}; export MW_SWEEP_CASE=${pair##*:}; mkdir -p $OUT/$name
  for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    p=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$name/$p -o c -- python3 $REPO/tools/sweep_measurements.py > $OUT/$name/$p.json 2> $OUT/$name/$p.err
  done
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, sys, collections, glob, json, os
src, tag = sys.argv[1], sys.argv[2]
res = {"tag": tag, "note": "per move of the timed k_sweep launch of tools/sweep_measurements.py <case> (the longest dispatch); hbm bytes = (2 FETCH_SIZE + WRITE_SIZE) x 1024"}
for name in sorted(os.listdir(src)):
    d = os.path.join(src, name)
    if not os.path.isdir(d):
        continue
    vals = {}
    moves = None
    for p in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(os.path.join(d, p, "**", "*counter_collection.csv"), recursive=True)
        if not f:
            continue
        per, dur = collections.defaultdict(dict), {}
        for r in csv.DictReader(open(f[0])):
            if "k_sweep" not in r["Kernel_Name"]:
                continue
            k = r["Dispatch_Id"]
            per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            dur[k] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = max(dur, key=dur.get)
        vals.update(per[k])
        if p == "SQ_INSTS_VALU":
            vals["kernel_us"] = dur[k] / 1e3
        o = list(json.load(open(os.path.join(d, p + ".json"))).values())[0]
        moves = o["walkers"] * o["moves_per_walker"]
    if moves and "SQ_INSTS_VALU" in vals:
        e = {"moves": moves, "SQ_INSTS_VALU_per_move": vals["SQ_INSTS_VALU"] / moves, "kernel_us_in_counter_pass": vals["kernel_us"]}
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY"):
            if c in vals:
                e[c + "_per_move"] = vals[c] / moves
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            e["hbm_bytes_per_move"] = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0 / moves
        res[name] = e
json.dump(res, open(os.path.join(os.path.dirname(src), "sweep_counters.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name "*.csv" -delete
