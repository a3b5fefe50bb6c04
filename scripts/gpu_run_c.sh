#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2 3; do
for v in 0 1; do
MCGRA_NO_ST3=$v python bench.py --no-cpu-baseline --no-split-probe --no-shard-probe --steps 40 > gpurun_out/${TAG}_x.json 2>/dev/null
python -c "import json;j=json.load(open('gpurun_out/${TAG}_x.json'));print('no_st3=$v', round(j['value'],2), round(j['ms_per_step'],3), round(j['roofline']['avg_launch_ms'],3))"
done
done
