#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" [bench args...]   -> gpurun_out/pmc_<tag>/ (one --pmc pass, no tracing)
set -e
TAG=$1; CTR=$2; shift; shift
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc $CTR --output-format csv -d $OUT -o pmc -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.log
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*counter_collection.csv")[0])))
agg = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith("mw::"):
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:45s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
