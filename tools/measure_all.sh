#!/bin/bash
# Everything under profiles/<tag>_* that is not the rocprof passes: the default bench line (CPU baseline included),
# single-call latencies (Python and C), the sweep driver, the farm (NVT / NPT / 'dd'), whole-program wall times.
# usage (on the GPU box): tools/measure_all.sh <tag>   -> gpurun_out/<tag>_*.json|txt
TAG=$1
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
python3 bench.py > $O/${TAG}_bench_with_cpu_baseline.json 2> $O/${TAG}_bench.err
python3 tools/extra_measurements.py > $O/${TAG}_single_call_latency.json 2> $O/${TAG}_extra.err
python3 tools/dump_workload.py /tmp/wl.bin --walkers 64 --moves 256 > /dev/null 2>&1 && ./tools/kbench /tmp/wl.bin 3 latency > $O/${TAG}_kbench_latency.txt 2>&1
MW_LOCAL_SERVER=0 ./tools/kbench /tmp/wl.bin 3 latency >> $O/${TAG}_kbench_latency.txt 2>&1
python3 tools/sweep_measurements.py > $O/${TAG}_sweep_measurements.json 2> $O/${TAG}_sweep.err
python3 -m mc_water_ls_mw_amd.farm --walkers 8192 --cycles 100 --sync 25 > $O/${TAG}_farm_nvt.json 2> $O/${TAG}_farm.err
python3 -m mc_water_ls_mw_amd.farm --walkers 8192 --cycles 100 --sync 25 --npt > $O/${TAG}_farm_npt.json 2>> $O/${TAG}_farm.err
python3 tools/program_walltime.py 2000 > $O/${TAG}_program_walltime.json 2> $O/${TAG}_walltime.err
ls -la $O/${TAG}_*
