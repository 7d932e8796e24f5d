#!/bin/bash
# k_model_energy A/B on one box (same binary): persistent workgroups (default) against one workgroup per box (MW_MODEL_PERSIST=0).
cd "$(dirname "$0")/.."
for r in 1 2 3; do
for v in 1 0; do
  MW_MODEL_PERSIST=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('persist=$v', 'model_energy_ms', round(b['kernels']['k_model_energy']['avg_ms'],5), 'move_ms', round(b['kernels']['k_move_energy']['avg_ms'],4), 'value', '%.4g' % b['value'], 'err', b['walker0_rel_err_vs_golden'])" || exit 1
done
done
