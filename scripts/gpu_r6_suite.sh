#!/bin/bash
# round 6: the whole GPU suite + the configs[2]-shaped and KL bench workloads
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6
timeout 3000 python -m pytest tests/ -q -m gpu -x > gpurun_out/r6/suite.log 2>&1
echo "suite rc=$?" | tee -a gpurun_out/r6/suite.log
tail -25 gpurun_out/r6/suite.log
for wl in citeseer-shape-gat-hsic synthetic-10k-kl; do
  timeout 300 python bench.py --workload $wl --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r6/bench_$wl.json 2> gpurun_out/r6/bench_$wl.err
  echo "$wl rc=$?"; python -c "import json,sys; d=json.load(open('gpurun_out/r6/bench_$wl.json')); print(d['value'], d['ms_per_step'], d['config'].get('fused_steps'), d['config'].get('general_steps'))"
done
