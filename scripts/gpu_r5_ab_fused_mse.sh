#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for wl in synthetic-10k-mse cora-shape-mse; do for f in 0 1; do
MCGRA_AB=1 MCGRA_NO_FUSED_LR=$f python3 bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline --no-split-probe 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read()); print('NO_FUSED=$f', d['config']['workload'], round(d['value'],1), round(d['ms_per_step'],4), d['auc'], d['config']['fused_steps'], d['config']['general_steps'])
"; done; done
python3 scripts/shard_emulate.py --echo --workload synthetic-10k-mse --worlds 1,2,4,8 --steps 20 2>&1 | grep '^{"world"' | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('echo 10k-mse world', d['world'], round(d['per_rank_compute_ms'],3), 'fused', d['fused_steps'], d['general_steps'], 'collectives', d['collectives_per_step'])
"
(time python -m pytest tests/ -q -m gpu 2>&1 | grep -v "^\[W\|Gloo\|amdgpu.ids\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -25) 2>&1
