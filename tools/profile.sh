#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then the two PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools_profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -e
TAG=$1; shift
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
