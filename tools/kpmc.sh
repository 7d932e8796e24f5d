#!/bin/bash
# usage: tools/kpmc.sh <tag> "<counters>" <program> [args...]   (program = a binary, e.g. tools/kbench; never a wrapper)
# One rocprofv3 --pmc pass (with --kernel-trace only), per-kernel means -> gpurun_out/pmc_<tag>/summary.txt
set -e
TAG=$1; CTR=$2; shift; shift
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
PROG=$REPO/$1; shift
cd /tmp
rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT -o pmc -- $PROG "$@" > $OUT/out.txt 2> $OUT/err.log
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, sys, collections, glob
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*counter_collection.csv")[0])))
agg = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if k.startswith(("mw::", "kb::")):
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:52s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
