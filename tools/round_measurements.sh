#!/bin/bash
# A round's final measurement set in one GPU call: every case of the Monte Carlo driver, its cycle stamps, then the bench under
# rocprofv3 (tools/profile.sh) and its summary.  usage (on the GPU box): tools/round_measurements.sh <tag>   -> gpurun_out/<tag>_*
# (afterwards, here: python tools/summarize_profile.py <tag>, and copy what is to be kept into profiles/)
TAG=${1:-r05}
O=gpurun_out
mkdir -p $O
rm -f $O/${TAG}_sweep_measurements.txt
for c in pair48 pair48wl npt48 farm48 farm48npt eight48 one48 one48plain one48npt eight48npt pair1536 ih4096 one4096 one1536; do
  MW_SWEEP_CASE=$c timeout -k 10 200 python3 tools/sweep_measurements.py 2>/dev/null >> $O/${TAG}_sweep_measurements.txt || echo "case $c failed" >> $O/${TAG}_sweep_measurements.txt
done
python3 tools/variants.py stamps > $O/${TAG}_variants.txt 2>&1 && MW_HIP_LIB=tools/variants/libmw_hip_stamps.so timeout -k 10 200 python3 tools/sweep_stamps.py one48 one48npt > $O/${TAG}_sweep_stamps.json 2> $O/${TAG}_stamps.err
bash tools/profile.sh ${TAG} > $O/${TAG}_profile.log 2>&1
python3 tools/summarize_profile.py ${TAG} > $O/${TAG}_summarize.log 2>&1
tail -3 $O/${TAG}_summarize.log
