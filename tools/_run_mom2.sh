mkdir -p gpurun_out/r04w
python tools/variants.py stamps > gpurun_out/r04w/build.txt 2>&1 && MW_HIP_LIB=tools/variants/libmw_hip_stamps.so timeout -k 10 120 python tools/sweep_stamps.py one48 > gpurun_out/r04w/stamps.json 2> gpurun_out/r04w/stamps.err
