#!/bin/bash
# A/B of the single-move kernel on the bench workload under environment switches (one bench run each).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  ms=$(env $envs python3 bench.py --no-cpu-baseline --no-secondary --no-exchange --steps 20 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['kernels']['k_move_energy']['avg_ms'])")
  echo "move kernel [$cfg]: $ms ms"
done
