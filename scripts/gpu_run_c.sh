#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export MCGRA_REPORT_DIR="$GRAFT_REPO_ROOT/gpurun_out"
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|^tests" | tail -8
for i in 1 2; do
python bench.py --workload cora-shape-hsic --no-cpu-baseline --no-split-probe --steps 200 --warmup 10 > gpurun_out/${TAG}_cora.json 2> gpurun_out/${TAG}_cora.err
python -c "import json;j=json.load(open('gpurun_out/${TAG}_cora.json'));print('cora', j['value'], j['ms_per_step'])"
python bench.py --no-cpu-baseline --no-split-probe --steps 30 > gpurun_out/${TAG}_10k.json 2>/dev/null
python -c "import json;j=json.load(open('gpurun_out/${TAG}_10k.json'));print('10k', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"
done
