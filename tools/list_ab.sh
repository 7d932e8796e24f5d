#!/bin/bash
# A/B of the neighbour-list build on the bench workload under different environment switches (one bench run each).
# usage: tools/list_ab.sh "VAR=val VAR2=val" "..." ...   ("-" = defaults)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  for rep in 1 2; do
    ms=$(env $envs python3 bench.py --no-cpu-baseline --no-secondary --no-exchange --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['kernels']['k_build_neighbours']['ms_for_all_walkers'])")
    echo "list build [$cfg] rep $rep: $ms ms"
  done
done
