#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_readme.py tests/test_gpu_multiproc.py -q -m gpu -x -k "fused or elementwise or mse or kl or MSE or KL" 2>&1 | tail -3
for wl in cora-shape-mse cora-shape-kl synthetic-10k-mse; do
  python bench.py --workload $wl --steps 300 --warmup 40 --no-cpu-baseline --no-split-probe 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl ms/step', round(d['ms_per_step'],4), d['auc'], d['config'].get('fused_steps'))"
done
