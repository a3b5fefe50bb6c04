#!/bin/bash
# same-box A/B of two builds on bench workloads: ab_build/libmcgra_prev.so (MCGRA_LIB_PATH) against the tree's library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
WL=${WL:-synthetic-10k-hsic}; STEPS=${STEPS:-100}
for rep in 1 2 3; do for lib in prev tree; do
  if [ $lib = prev ]; then export MCGRA_LIB_PATH=$PWD/ab_build/libmcgra_prev.so; else unset MCGRA_LIB_PATH; fi
  python3 bench.py --workload $WL --steps $STEPS --warmup 20 --no-cpu-baseline --no-split-probe 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL $lib rep $rep', round(d['value'],2), round(d['ms_per_step'],4), 'product', round(d['roofline'].get('avg_launch_ms', 0), 4) if 'roofline' in d else '')"
done; done
