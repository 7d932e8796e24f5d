#!/bin/bash
# rocprofv3 kernel stats of a farm run: tools/profile_farm.sh <tag> [farm args...]  -> prints the top kernels
set -e
TAG=$1; shift
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/farmprof_$TAG
mkdir -p $OUT
export PYTHONPATH=$REPO
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o f -- python3 -m mc_water_ls_mw_amd.farm "$@" > $OUT/farm.json 2> $OUT/err.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/f_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(f'{r["Name"][:64]:64s} calls={r["Calls"]:>6s} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
cut -c 1-160 $OUT/farm.json
find $OUT -name "*.csv" -size +1M -delete
