cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3d
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "split or fused_lowrank or gram or sharded_ranks_match" 2>&1 | tail -4
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r3d/bench10k.json 2> gpurun_out/r3d/bench10k.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r3d/bench10k.json'))
print('10k', round(j['value'],2), round(j['ms_per_step'],3), j['config'].get('fused_steps'), j.get('roofline',{}).get('avg_launch_ms'), j.get('roofline',{}).get('alone',{}).get('avg_launch_ms'), j.get('roofline',{}).get('alone',{}).get('issued_frac'), 'auc', j['auc'])
PY
done
