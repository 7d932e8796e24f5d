#!/bin/bash
# A/B of the sweep driver inside ONE gpurun call (box-to-box variation is +-4 %): tools/sweep_ab.sh <env var> <a> <b> [cases...]
V=$1; A=$2; B=$3; shift; shift; shift
CASES=${@:-pair48 pair48wl}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for val in $A $B; do for c in $CASES; do
  env $V=$val MW_SWEEP_CASE=$c python3 $REPO/tools/sweep_measurements.py > /tmp/sweep_ab.json 2> /tmp/sweep_ab.err; rc=$?
  if [ $rc -ne 0 ]; then echo "$V=$val $c FAILED rc=$rc"; tail -5 /tmp/sweep_ab.err; continue; fi
  r=$(python3 -c "import json; d=json.load(open('/tmp/sweep_ab.json')); v=list(d.values())[0]; print('%.4g' % v['moves_per_s'])")
  echo "$V=$val $c $r"
done; done; done
