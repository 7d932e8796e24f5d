#!/bin/bash
# One SQ counter pass over bench.py's secondary measurements (configs[4]: 32768-molecule boxes; configs[2]): per-kernel means.
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_secondary
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -o s -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0])))
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict); grid = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("mw::k_model_energy<false"): continue
    key = (k, r["Grid_Size"])
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[key][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for key, cs in agg.items():
    d = sum(dur[key].values()) / len(dur[key])
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(key, "avg_us %.1f" % d, "valu_busy(grbm) %.3f" % (m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)), "waves/simd %.2f" % (m["SQ_WAVE_CYCLES"] * 4 / (1024 * cyc)),
          "wait_any %.2f" % (m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]), "insts %.3g" % m["SQ_INSTS_VALU"])
PY
find $OUT -name "*.csv" -size +1M -delete
