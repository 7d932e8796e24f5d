#!/bin/bash
# Single-call latency with the request lines in host-mapped memory (default) and in device memory (MW_SERVER_REQ=device).
cd "$(dirname "$0")/.."
for mode in host device host device; do
  echo "== MW_SERVER_REQ=$mode"
  MW_SERVER_REQ=$mode timeout -k 10 300 python tools/extra_measurements.py --child || exit 1
done
echo "== parity (device request lines)"
MW_SERVER_REQ=device timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fortran_dropin.py -x -q 2>&1 | tail -3
