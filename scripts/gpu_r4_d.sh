#!/bin/bash
# round 4, call d: thin edge tiles in the product, ld alignment: parity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4d; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "split or product or edge or ori or shard" > $O/parity_split.txt 2>&1; tail -5 $O/parity_split.txt
for rep in 1 2; do
  for cfg in "1 4" "0 4" "1 32" "0 32"; do
    set -- $cfg
    MCGRA_SPLIT_EDGE=$1 MCGRA_LD_ALIGN=$2 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-split-probe 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('edge $1 ld $2 rep $rep', round(l['value'],2), round(l['ms_per_step'],4), 'product', round(l['roofline']['avg_launch_ms'],4), 'alone', round(l['roofline']['alone']['avg_launch_ms'],4), 'auc', l['auc'])" | tee -a $O/ab.txt
  done
done
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multiproc.py -q -m gpu > $O/fullsize.txt 2>&1; tail -5 $O/fullsize.txt
MCGRA_LD_ALIGN=32 timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "10k" > $O/fullsize_ld32.txt 2>&1; tail -3 $O/fullsize_ld32.txt
timeout 600 python scripts/shard_emulate.py --worlds 1,2,4,8 --steps 6 > $O/emulate.log 2>&1; grep '^{"world"' $O/emulate.log
