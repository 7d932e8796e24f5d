#!/bin/bash
# E / D split of one chain without stamps: the production library against the builds without decisions / without evaluations
# (tools/variants.py nodecide, noeval), look-ahead 1 and as chosen.  usage: tools/sweep_ablate.sh [case ...]
CASES=${@:-one48 one48npt one4096}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for c in $CASES; do for v in ${VARIANTS:-prod nodecide noeval}; do for a in 1 0; do
  lib=$REPO/mc_water_ls_mw_amd/libmw_hip.so; [ $v != prod ] && lib=$REPO/tools/variants/libmw_hip_$v.so
  if [ $a = 1 ]; then export MW_SWEEP_AHEAD=1; else unset MW_SWEEP_AHEAD; fi
  r=$(MW_HIP_LIB=$lib MW_SWEEP_CASE=$c timeout -k 10 120 python3 $REPO/tools/sweep_measurements.py 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); v=list(d.values())[0]; print('%.3f us/move acc=%.3f' % (v['us_per_move_per_walker'], v['acceptance']))")
  echo "$c $v ahead=$([ $a = 1 ] && echo 1 || echo auto): $r"
done; done; done
