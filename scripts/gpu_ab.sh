#!/bin/bash
# A/B runs of the headline bench on ONE box: VAR=<env var> VALS="a b c" REPS=2 bash scripts/gpu_ab.sh
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in $(seq 1 ${REPS:-2}); do
for v in $VALS; do
env $VAR=$v python bench.py --no-cpu-baseline --no-split-probe --no-shard-probe --steps 40 ${EXTRA} > gpurun_out/ab_x.json 2>/dev/null
python -c "import json;j=json.load(open('gpurun_out/ab_x.json'));print('$VAR=$v', round(j['value'],2), round(j['ms_per_step'],3), round(j['roofline']['avg_launch_ms'],3) if j.get('roofline') else '')"
done
done
