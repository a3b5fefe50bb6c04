#!/bin/bash
# round 4, call b: sharding tests after the exchange rework (8 collectives per step), per-rank emulation, power-trace dry run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4b; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_parity.py -q -m gpu -k "shard or row_block or bench or abandoned" > $O/tests.txt 2>&1
cat $O/tests.txt | tail -15
timeout 600 python scripts/shard_emulate.py --worlds 1,2,4,8 --steps 6 > $O/emulate.log 2>&1; grep '^{"world"' $O/emulate.log
timeout 300 python scripts/power_trace.py --out $O/power_trace.json > $O/power_trace.log 2>&1; tail -60 $O/power_trace.log
