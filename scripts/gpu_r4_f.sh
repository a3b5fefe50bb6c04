#!/bin/bash
# round 4, call f: early pack A/B + parity
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4g; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "early or fused or adopt or monitor or shard or abandoned" > $O/parity.txt 2>&1; tail -4 $O/parity.txt
for rep in 1 2 3; do
  for ep in 1 0; do
    MCGRA_SWAP_STREAMS=$ep python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-split-probe 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('swap $ep rep $rep', round(l['value'],2), round(l['ms_per_step'],4), 'product', round(l['roofline']['avg_launch_ms'],4), 'alone', round(l['roofline']['alone']['avg_launch_ms'],4), 'auc', l['auc'])" | tee -a $O/ab.txt
  done
done
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "10k" > $O/fullsize.txt 2>&1; tail -3 $O/fullsize.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-split-probe > $O/prof.log 2>&1
