#!/bin/bash
# Single-call latency in a C loop: request lines in device memory (default) and in host memory, one launch per call.
cd "$(dirname "$0")/.."
python tools/dump_workload.py /tmp/w.bin --walkers 64 --moves 256 > /dev/null || exit 1
for v in "MW_SERVER_REQ=device" "MW_SERVER_REQ=host" "MW_LOCAL_SERVER=0"; do
echo "== $v"
env $v MW_SERVER_STAMPS=1 timeout -k 10 300 tools/kbench /tmp/w.bin 3 latency || exit 1
done
