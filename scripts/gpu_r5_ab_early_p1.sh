#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "forks_the_next" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -25
for e in 0 1; do
  echo "MCGRA_EARLY_P1=$e"
  MCGRA_AB=1 MCGRA_EARLY_P1=$e python3 scripts/shard_emulate.py --echo --worlds 1,2,4,8 --steps 40 2>&1 | grep '^{"world"' | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('  world', d['world'], round(d['per_rank_compute_ms'],3), 'product', round(d['product_ms_per_step'],3), 'fused', d['fused_steps'], d['general_steps'])
"
  MCGRA_AB=1 MCGRA_EARLY_P1=$e python3 scripts/shard_emulate.py --echo --workload synthetic-30k-hsic-3layer --worlds 4,8 --steps 4 2>&1 | grep '^{"world"' | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('  30k world', d['world'], round(d['per_rank_compute_ms'],3), 'product', round(d['product_ms_per_step'],3), 'fused', d['fused_steps'], d['general_steps'])
"
done
