#!/bin/bash
# round 6: whole GPU suite on the current build, KL kernel tables, configs[2]-shaped bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out/r6
timeout 3300 python -m pytest tests/ -q -m gpu > gpurun_out/r6/suite.log 2>&1
echo "suite rc=$?" | tee -a gpurun_out/r6/suite.log
tail -30 gpurun_out/r6/suite.log | cut -c1-220
for i in 1 2; do
timeout 300 python bench.py --workload citeseer-shape-gat-hsic --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r6/bench_cit_$i.json 2> gpurun_out/r6/bench_cit_$i.err
python -c "import json; d=json.load(open('gpurun_out/r6/bench_cit_$i.json')); print('citeseer-shape-gat-hsic', d['value'], d['ms_per_step'])"
done
bash scripts/gpu_r6_kl_prof.sh
