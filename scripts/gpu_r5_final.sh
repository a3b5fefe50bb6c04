#!/bin/bash
# end-of-round visit (round 5): the whole -m gpu suite, the bench line (CPU baseline + live traffic), the driver's window, the other
# workloads (incl. the fused MSELoss step and the Gram evaluation), the README-line diagnostics (42 lines), the 10k diagnostics,
# the echo emulation of the sharded HSIC and MSELoss steps.  PMC passes: TAG=r05 PMC_STEPS=11 bash scripts/gpu_pmc.sh (own call).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05f; mkdir -p $O
python -m pytest tests -m gpu -q --tb=short --maxfail=30 -p no:cacheprovider > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -1
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r05f/bench.json').read().strip().splitlines()[-1])
r=j['roofline']
print('10k', round(j['value'],2), round(j['ms_per_step'],3), 'insitu', round(r['avg_launch_ms'],3), r['frac'], r['issued_frac'], 'alone', r.get('alone'), 'auc', j['auc'], 'cpu', j.get('cpu_baseline',{}).get('value'), j.get('other_workloads'), j.get('gram_path_probe'))
PY
python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2>/dev/null
python -c "
import json; l=json.loads(open('$O/bench_20_5.json').read().strip().splitlines()[-1]); print('20/5', l['value'], l['ms_per_step'])"
TAG=r05f/o bash scripts/gpu_other_benches.sh
python bench.py --workload synthetic-10k-hsic-masked --no-cpu-baseline --no-split-probe --steps 20 > $O/o_bench_synthetic-10k-hsic-masked.json 2>/dev/null
python bench.py --workload synthetic-10k-mse --no-cpu-baseline --no-split-probe --steps 100 --warmup 20 > $O/o_bench_synthetic-10k-mse.json 2>/dev/null
MCGRA_AB=1 MCGRA_NO_FUSED_LR=1 python bench.py --workload synthetic-10k-mse --no-cpu-baseline --no-split-probe --steps 40 --warmup 10 > $O/o_bench_synthetic-10k-mse_general.json 2>/dev/null
python scripts/shard_emulate.py --echo --worlds 1,2,4,8 --steps 20 > $O/shard_echo_10k_hsic.log 2>&1; grep '^{"world"' $O/shard_echo_10k_hsic.log | cut -c1-200
python scripts/shard_emulate.py --echo --workload synthetic-10k-mse --worlds 1,2,4,8 --steps 20 > $O/shard_echo_10k_mse.log 2>&1; grep '^{"world"' $O/shard_echo_10k_mse.log | cut -c1-200
python scripts/shard_emulate.py --echo --workload synthetic-30k-hsic-3layer --worlds 1,2,4,8 --steps 4 > $O/shard_echo_30k_hsic.log 2>&1; grep '^{"world"' $O/shard_echo_30k_hsic.log | cut -c1-200
python scripts/diag_10k.py > $O/diag_10k.txt 2>&1; tail -12 $O/diag_10k.txt | cut -c1-330
python scripts/diag_readme.py > $O/readme_lines.txt 2>&1; tail -5 $O/readme_lines.txt | cut -c1-200
for ep in 20 100; do [ -f tests/golden/horizon${ep}_readme.npz ] && python scripts/diag_readme_horizon.py $ep 2>/dev/null > $O/readme_horizon$ep.txt && tail -2 $O/readme_horizon$ep.txt; done
for w in hsic kl; do python scripts/citeseer_gat_steps.py $w 40 2>&1 | grep "ms/step"; done > $O/citeseer_gat_steps.txt; cat $O/citeseer_gat_steps.txt
(cd /tmp; rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r05f_gat" -- python3 "$GRAFT_REPO_ROOT/scripts/citeseer_gat_steps.py" hsic 12 > /dev/null 2>&1)
python scripts/general_step_timeline.py "$(find gpurun_out/r05f_gat -name '*kernel_trace.csv' | head -1)" > $O/citeseer_gat_step_timeline.txt 2>&1; rm -rf gpurun_out/r05f_gat; head -1 $O/citeseer_gat_step_timeline.txt
python scripts/gemm_mid_bench.py 2>&1 | grep "^n " > $O/gemm_mid_bench.txt; cat $O/gemm_mid_bench.txt
for e in 0 1; do echo "MCGRA_EARLY_P1=$e"; MCGRA_AB=1 MCGRA_EARLY_P1=$e python3 scripts/shard_emulate.py --echo --worlds 2,4,8 --steps 40 2>&1 | grep '^{"world"'; done > $O/ab_early_p1.txt; cut -c1-140 $O/ab_early_p1.txt
