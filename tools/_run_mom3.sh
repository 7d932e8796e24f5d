mkdir -p gpurun_out/r05a
timeout -k 10 400 python -m pytest tests/test_sweep.py tests/test_gpu_options.py tests/test_gpu_minu.py tests/test_gpu_schedule.py tests/test_gpu_bench_chain.py -m gpu -q -x > gpurun_out/r05a/tests.txt 2>&1
rc=$?; tail -3 gpurun_out/r05a/tests.txt
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
rm -f gpurun_out/r05a/meas.txt
for c in farm48npt one48npt eight48npt npt48; do
  echo "== $c" >> gpurun_out/r05a/meas.txt; MW_SWEEP_CASE=$c timeout -k 10 120 python tools/sweep_measurements.py 2>/dev/null | grep -E 'moves_per_s|us_per_move' >> gpurun_out/r05a/meas.txt || exit 9
done
cat gpurun_out/r05a/meas.txt
python tools/variants.py stamps > gpurun_out/r05a/build.txt 2>&1 && MW_HIP_LIB=tools/variants/libmw_hip_stamps.so timeout -k 10 120 python tools/sweep_stamps.py one48npt > gpurun_out/r05a/stamps.json 2> gpurun_out/r05a/stamps.err
