#!/bin/bash
# Counters for the device-resident Monte Carlo driver (k_sweep_translation), one case of tools/sweep_measurements.py at a
# time: kernel trace + stats, then separate --pmc passes (SQ has 8 slots; FETCH_SIZE and WRITE_SIZE cannot share a pass).
# usage: tools/profile_sweep.sh <tag> [case ...]   (cases: pair48 pair1536 ih4096 npt48; default all)
#        -> gpurun_out/sweepprof_<tag>/<case>/{trace,sq1,sq2,fetch,write}; then tools/summarize_sweep_profile.py <tag>
set -e
TAG=$1; shift
CASES=${@:-pair48 pair1536 ih4096 npt48}
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$REPO
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE"
cd /tmp
for C in $CASES; do
  OUT=$REPO/gpurun_out/sweepprof_$TAG/$C
  mkdir -p $OUT
  export MW_SWEEP_CASE=$C
  P="python3 $REPO/tools/sweep_measurements.py"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $P > $OUT/run_trace.json 2> $OUT/trace.err
  rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/sq1 -o c -- $P > $OUT/run_sq1.json 2> $OUT/sq1.err
  rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/sq2 -o c -- $P > $OUT/run_sq2.json 2> $OUT/sq2.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o c -- $P > $OUT/run_fetch.json 2> $OUT/fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o c -- $P > $OUT/run_write.json 2> $OUT/write.err
  find $OUT -name "*kernel_trace.csv" -size +4M -delete
  echo "$C done"
done
python3 $REPO/tools/summarize_sweep_profile.py $TAG
# the raw per-dispatch tables are tens of MiB per pass: only the kernel-stats tables and the summaries travel back
find $REPO/gpurun_out/sweepprof_$TAG -name "*.csv" ! -name "*kernel_stats.csv" -delete
