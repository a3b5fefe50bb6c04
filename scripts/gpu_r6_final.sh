#!/bin/bash
# end-of-round visit (round 6): the whole -m gpu suite, smoke, the bench line (CPU baseline + live traffic + other workloads incl. the
# fused KL step and the single-plane mode), the driver's window, the other workloads on their own, the KL kernel tables, the
# single-plane accuracy table, the configs[2]-shaped step launch by launch, the Gram-evaluation step launch by launch, the 10k and
# README-line diagnostics.  PMC passes: TAG=r06 PMC_STEPS=11 bash scripts/gpu_pmc.sh (own call).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; O=gpurun_out/r06f; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --tb=short --maxfail=30 -p no:cacheprovider > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' 2>&1 | tail -1
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06f/bench.json').read().strip().splitlines()[-1])
r=j['roofline']
print('10k', round(j['value'],2), round(j['ms_per_step'],3), 'insitu', round(r['avg_launch_ms'],3), r['frac'], r['issued_frac'], 'alone', r.get('alone'), 'traffic', r.get('traffic'), 'auc', j['auc'], 'cpu', j.get('cpu_baseline',{}).get('value'))
for k, v in (j.get('other_workloads') or {}).items():
    print('   ', k, v if not isinstance(v, dict) else {a: v[a] for a in ('value', 'ms_per_step', 'fused_steps', 'general_steps', 'auc', 'product_avg_launch_ms') if a in v})
print('    gram', j.get('gram_path_probe'))
PY
python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2>/dev/null
python -c "
import json; l=json.loads(open('$O/bench_20_5.json').read().strip().splitlines()[-1]); print('20/5', l['value'], l['ms_per_step'])"
for wl in synthetic-10k-kl cora-shape-kl synthetic-10k-mse cora-shape-mse cora-shape-hsic synthetic-4k-hsic citeseer-shape-gat-hsic synthetic-10k-hsic-masked; do
  LT=""; case $wl in *-kl|*-mse) LT="--live-traffic";; esac      # (the fused MSELoss / KL steps: roofline.traffic per step from two --pmc child passes)
  python bench.py --workload $wl --no-cpu-baseline --no-split-probe $LT --steps 100 --warmup 20 > $O/o_bench_$wl.json 2>/dev/null
  python -c "
import json; l=json.loads(open('$O/o_bench_$wl.json').read().strip().splitlines()[-1]); print('$wl', round(l['value'],1), round(l['ms_per_step'],4), l['config'].get('fused_steps'), l['config'].get('general_steps'), (l.get('roofline') or {}).get('frac'))"
done
python bench.py --workload synthetic-30k-hsic-3layer --no-cpu-baseline --no-split-probe --steps 6 --warmup 2 > $O/o_bench_synthetic-30k-hsic-3layer.json 2>/dev/null
python -c "
import json; l=json.loads(open('$O/o_bench_synthetic-30k-hsic-3layer.json').read().strip().splitlines()[-1]); print('30k', l['value'], l['ms_per_step'])"
MCGRA_AB=1 MCGRA_NO_FUSED_LR=1 python bench.py --workload synthetic-10k-kl --no-cpu-baseline --no-split-probe --steps 40 --warmup 10 > $O/o_bench_synthetic-10k-kl_general.json 2>/dev/null
python -c "
import json; l=json.loads(open('$O/o_bench_synthetic-10k-kl_general.json').read().strip().splitlines()[-1]); print('10k-kl general step', l['value'], l['ms_per_step'])"
python scripts/single_plane_table.py > $O/single_plane_table.txt 2>/dev/null; cat $O/single_plane_table.txt
python scripts/shard_emulate.py --echo --workload synthetic-10k-kl --worlds 1,2,4,8 --steps 20 > $O/shard_echo_10k_kl.log 2>&1; grep '^{"world"' $O/shard_echo_10k_kl.log | cut -c1-200
python scripts/shard_emulate.py --echo --worlds 1,2,4,8 --steps 20 > $O/shard_echo_10k_hsic.log 2>&1; grep '^{"world"' $O/shard_echo_10k_hsic.log | cut -c1-200
python scripts/diag_10k.py > $O/diag_10k.txt 2>&1; tail -12 $O/diag_10k.txt | cut -c1-330
python scripts/diag_readme.py > $O/readme_lines.txt 2>&1; tail -5 $O/readme_lines.txt | cut -c1-200
for ep in 20 100; do [ -f tests/golden/horizon${ep}_readme.npz ] && python scripts/diag_readme_horizon.py $ep 2>/dev/null > $O/readme_horizon$ep.txt && tail -2 $O/readme_horizon$ep.txt; done
cd /tmp
for wl in synthetic-10k-kl cora-shape-kl; do
  rm -rf "$R/$O/stats_$wl"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/stats_$wl" -- python3 "$R/bench.py" --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-split-probe > "$R/$O/stats_$wl.log" 2>&1
  (cd "$R"; echo "== $wl (50 steps incl. warm-up)"; python3 scripts/kstats.py $O/stats_$wl 50 22) > "$R/$O/kstats_$wl.txt"; head -14 "$R/$O/kstats_$wl.txt"
  find "$R/$O/stats_$wl" -name "*kernel_trace*" -size +8M -delete 2>/dev/null
done
rm -rf "$R/$O/cit_trace" "$R/$O/gram_trace"
rocprofv3 --kernel-trace --output-format csv -d "$R/$O/cit_trace" -- python3 "$R/bench.py" --workload citeseer-shape-gat-hsic --steps 12 --warmup 4 --no-cpu-baseline --no-split-probe > /dev/null 2>&1
MCGRA_AB=1 MCGRA_NO_LOWRANK=1 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/gram_trace" -- python3 "$R/bench.py" --steps 8 --warmup 3 --no-cpu-baseline --no-split-probe > /dev/null 2>&1
cd "$R"
python3 scripts/general_step_timeline.py "$(find $O/cit_trace -name '*kernel_trace.csv' | head -1)" > $O/citeseer_gat_step_timeline.txt 2>&1; head -1 $O/citeseer_gat_step_timeline.txt
python3 scripts/gram_timeline.py "$(find $O/gram_trace -name '*kernel_trace.csv' | head -1)" > $O/gram_path_timeline.txt 2>&1; head -1 $O/gram_path_timeline.txt
rm -rf $O/cit_trace $O/gram_trace
