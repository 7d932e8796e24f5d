#!/bin/bash
# Instruction counts of ONE chain per move: rocprofv3 --pmc over tools/sweep_measurements.py <case> for the production library and
# the builds without decisions / without evaluations.  usage: tools/sweep_chain_pmc.sh <case> [ahead]
C=${1:-one48}; A=${2:-}
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$REPO MW_SWEEP_CASE=$C
[ -n "$A" ] && export MW_SWEEP_AHEAD=$A
CTR="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
cd /tmp
for v in ${VARIANTS:-prod nodecide noeval}; do
  lib=$REPO/mc_water_ls_mw_amd/libmw_hip.so; [ $v != prod ] && lib=$REPO/tools/variants/libmw_hip_$v.so
  export MW_HIP_LIB=$lib
  OUT=$REPO/gpurun_out/chainpmc_${C}_${A:-auto}_$v
  mkdir -p $OUT
  rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT -o pmc -- python3 $REPO/tools/sweep_measurements.py > $OUT/out.json 2> $OUT/err.log
  python3 - "$OUT" "$v" <<'PY'
import csv, sys, collections, glob, json
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*counter_collection.csv")[0])))
per = collections.defaultdict(dict); dur = {}
for r in rows:
    if "k_sweep" not in r["Kernel_Name"]: continue
    d = r["Dispatch_Id"]
    per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
d = max(dur, key=dur.get)
res = list(json.load(open(sys.argv[1] + "/out.json")).values())[0]
n = res["moves_per_walker"] * res["walkers"]
print(sys.argv[2], "us/move=%.3f" % (dur[d] / n), " ".join("%s=%.1f" % (k.replace("SQ_", ""), v / n) for k, v in sorted(per[d].items())))
PY
  find $OUT -name "*.csv" -delete
done
