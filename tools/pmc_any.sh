#!/bin/bash
# usage: tools/pmc_any.sh <tag> "<counters>" <script.py> [args...]  -> per-kernel means of the counters
set -e
TAG=$1; CTR=$2; shift; shift
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
SCRIPT=$REPO/$1; shift
cd /tmp
rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT -o pmc -- python3 $SCRIPT "$@" > $OUT/out.json 2> $OUT/err.log
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*counter_collection.csv")[0])))
agg = collections.defaultdict(list)
dur = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("mw::"):
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, c), v in sorted(agg.items()):
    print(f"{k:60s} {c:24s} n={len(v):4d} sum={sum(v):.6g} max={max(v):.6g}")
for k, v in dur.items():
    print(f"{k:60s} launches={len(v)} total_us={sum(v):.1f} max_us={max(v):.1f}")
PY
