#!/bin/bash
# round 6: the fused MSELoss / KL step's decode on the caller's stream (default) against the fourth stream (MCGRA_MSE_DECODE_SIDE=1) at Cora size
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused" 2>&1 | tail -3
for wl in cora-shape-mse cora-shape-kl; do for v in 0 1 0 1; do
  MCGRA_AB=1 MCGRA_MSE_DECODE_SIDE=$v python bench.py --workload $wl --steps 400 --warmup 40 --no-cpu-baseline --no-split-probe 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl decode_side=$v ms/step', round(d['ms_per_step'],4), d['auc'], d['config'].get('fused_steps'))"
done; done
