#!/bin/bash
# the bench lines of the round (default bench with the CPU baseline, the other workloads), after gpu_r4_final.sh
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04g; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r04g/bench.json').read().strip().splitlines()[-1])
r=j['roofline']
print('10k', round(j['value'],2), round(j['ms_per_step'],3), 'insitu', round(r['avg_launch_ms'],3), r['frac'], r['issued_frac'], 'alone', r.get('alone'), 'auc', j['auc'], 'cpu', j.get('cpu_baseline',{}).get('value'), j.get('other_workloads'), j.get('gram_path_probe'), j.get('step_outside_product'))
PY
TAG=r04g/o bash scripts/gpu_other_benches.sh
python bench.py --workload synthetic-10k-hsic-masked --no-cpu-baseline --no-split-probe > $O/o_bench_synthetic-10k-hsic-masked.json 2>/dev/null
python -c "
import json; l=json.loads(open('$O/o_bench_synthetic-10k-hsic-masked.json').read().strip().splitlines()[-1]); print('masked', l['value'], l['ms_per_step'], l['config']['fused_steps'], l['config']['masked_fused_steps'])"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-split-probe > $O/bench_20_3.json 2>/dev/null
python -c "
import json; l=json.loads(open('$O/bench_20_3.json').read().strip().splitlines()[-1]); print('20/3', l['value'], l['ms_per_step'])"
