cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|^tests" | tail -4
for i in 1 2; do
python bench.py --workload cora-shape-hsic --no-cpu-baseline --no-split-probe --steps 200 --warmup 10 > gpurun_out/x_cora.json 2>/dev/null
python -c "import json;j=json.load(open('gpurun_out/x_cora.json'));print('cora', round(j['value'],1), round(j['ms_per_step'],4))"
python bench.py --no-cpu-baseline --no-split-probe --no-shard-probe --steps 40 > gpurun_out/x_10k.json 2>/dev/null
python -c "import json;j=json.load(open('gpurun_out/x_10k.json'));print('10k', round(j['value'],2), round(j['ms_per_step'],3), round(j['roofline']['avg_launch_ms'],3))"
done
