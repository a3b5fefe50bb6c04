#!/bin/bash
# end-of-round visit: full GPU suite, the bench line (with the CPU baseline), the other workloads, diagnostics, shard emulation,
# power trace.  PMC passes: TAG=r04 bash scripts/gpu_pmc.sh (own call).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04f; mkdir -p $O
export MCGRA_REPORT_DIR="$GRAFT_REPO_ROOT/$O"
python -m pytest tests -m gpu -q --tb=short --maxfail=30 -p no:cacheprovider > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r04f/bench.json').read().strip().splitlines()[-1])
r=j['roofline']
print('10k', round(j['value'],2), round(j['ms_per_step'],3), 'insitu', round(r['avg_launch_ms'],3), r['frac'], r['issued_frac'], 'alone', r.get('alone'), 'auc', j['auc'], 'cpu', j.get('cpu_baseline',{}).get('value'), j.get('other_workloads'), j.get('gram_path_probe'))
PY
TAG=r04f/o bash scripts/gpu_other_benches.sh
python bench.py --workload synthetic-10k-hsic-masked --no-cpu-baseline --no-split-probe --steps 20 > $O/o_bench_synthetic-10k-hsic-masked.json 2>/dev/null
python scripts/shard_emulate.py > $O/shard_emulate.json.log 2>&1; grep '^{"world"' $O/shard_emulate.json.log | cut -c1-150
python scripts/diag_10k.py > $O/diag_10k.txt 2>&1; tail -12 $O/diag_10k.txt | cut -c1-330
python scripts/diag_readme.py > $O/readme_lines.txt 2>&1; tail -5 $O/readme_lines.txt | cut -c1-200
python scripts/power_trace.py --out $O/power_trace.json > $O/power_trace.log 2>&1; tail -3 $O/power_trace.log
# the driver's own window, and the early tail pass at the largest single-GPU shape
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-probe > $O/bench_20_5.json 2>/dev/null
python -c "
import json; l=json.loads(open('$O/bench_20_5.json').read().strip().splitlines()[-1]); print('20/5', l['value'], l['ms_per_step'])"
for i in 1 2 3; do for v in 1 0; do
MCGRA_EARLY_TAIL=$v python bench.py --steps 10 --warmup 3 --workload synthetic-30k-hsic-3layer --no-cpu-baseline --no-split-probe | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('30k early_tail $v', round(d['value'],3), round(d['ms_per_step'],2))"
done; done 2>/dev/null | tee $O/early_tail_30k.txt
