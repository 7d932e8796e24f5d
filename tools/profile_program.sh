#!/bin/bash
# HIP API + kernel statistics of the whole reference program running on the engine (oracle/_ref/mc_water_hip), ice1_sample input:
# where the drop-in host's wall time goes.  usage: tools/profile_program.sh <tag> [cycles]  -> gpurun_out/progprof_<tag>/
set -e
TAG=$1; CYC=${2:-500}
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/progprof_$TAG
mkdir -p $OUT
RUN=$(python3 - "$REPO" "$CYC" <<'PY'
import os, sys, tempfile
root, cyc = sys.argv[1], sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_full_program as fp
d = os.path.join(tempfile.mkdtemp(), "run")
fp._prepare(d, fp.LATTICE_SWITCH.replace("max_mc_cycles    = 400", f"max_mc_cycles    = {cyc}"), True)
print(d)
PY
)
cd $RUN
rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d $OUT -o p -- $REPO/oracle/_ref/mc_water_hip ice.input > $OUT/stdout.txt 2> $OUT/stderr.txt
cd $OUT
find . -name "*_trace.csv" -size +1M -delete
for f in $(find . -name "*stats.csv"); do echo "== $f"; head -12 $f | cut -c 1-200; done
