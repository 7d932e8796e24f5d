mkdir -p gpurun_out/r04w
rm -f gpurun_out/r04w/*.txt
timeout -k 10 400 python -m pytest tests/test_sweep.py tests/test_gpu_options.py tests/test_gpu_minu.py tests/test_gpu_schedule.py tests/test_gpu_bench_chain.py -m gpu -v -x > gpurun_out/r04w/tests.txt 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r04w/tests.txt
tail -5 gpurun_out/r04w/tests.txt
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
for c in pair48wl farm48 farm48npt one48 one48npt eight48npt; do
  for m in 1 0; do echo "== $c moments=$m" >> gpurun_out/r04w/meas.txt; MW_SWEEP_MOMENTS=$m MW_SWEEP_CASE=$c timeout -k 10 120 python tools/sweep_measurements.py 2>/dev/null | grep -E 'moves_per_s|us_per_move|acceptance' >> gpurun_out/r04w/meas.txt || exit 9; done
done
python tools/variants.py stamps > gpurun_out/r04w/build.txt 2>&1 && MW_HIP_LIB=tools/variants/libmw_hip_stamps.so timeout -k 10 120 python tools/sweep_stamps.py one48 > gpurun_out/r04w/stamps.json 2> gpurun_out/r04w/stamps.err
