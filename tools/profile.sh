#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass, and SQ has 8 slots: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/ ; then tools/summarize_profile.py <tag>
set -e
TAG=$1; shift
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
B="python3 $REPO/bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $B "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- $B "$@" > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq1 -o sq1 -- $B "$@" --steps 4 --warmup 1 > $OUT/bench_sq1.json 2> $OUT/sq1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o sq2 -- $B "$@" --steps 4 --warmup 1 > $OUT/bench_sq2.json 2> $OUT/sq2.err
find $OUT -name "*.csv" | head -20
