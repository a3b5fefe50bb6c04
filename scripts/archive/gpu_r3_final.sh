#!/bin/bash
# end-of-round visit: full GPU suite, the bench line (with the CPU baseline), the other workloads, Citeseer step time, shard emulation
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03f
export MCGRA_REPORT_DIR="$GRAFT_REPO_ROOT/gpurun_out/r03f"
python -m pytest tests -m gpu -q --tb=short --maxfail=30 -p no:cacheprovider > gpurun_out/r03f/pytest.log 2>&1
tail -4 gpurun_out/r03f/pytest.log
python bench.py > gpurun_out/r03f/bench.json 2> gpurun_out/r03f/bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r03f/bench.json'))
r=j['roofline']
print('10k', round(j['value'],2), round(j['ms_per_step'],3), 'insitu', round(r['avg_launch_ms'],3), r['frac'], r['issued_frac'], 'alone', r.get('alone'), 'auc', j['auc'], 'cpu', j.get('cpu_baseline',{}).get('value'), j.get('other_workloads'))
PY
TAG=r03f/o bash scripts/gpu_other_benches.sh
python scripts/shard_emulate.py > gpurun_out/r03f/shard_emulate.json.log 2>&1; tail -3 gpurun_out/r03f/shard_emulate.json.log
