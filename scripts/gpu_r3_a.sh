cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize.py::test_bench_workload_matches_reference_at_10k -p no:cacheprovider > gpurun_out/r3a/pytest.log 2>&1
tail -40 gpurun_out/r3a/pytest.log
python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r3a/bench10k.json 2> gpurun_out/r3a/bench10k.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r3a/bench10k.json'))
print('10k', round(j['value'],2), round(j['ms_per_step'],3), j['config'].get('fused_steps'), j['config'].get('general_steps'), j.get('roofline',{}).get('avg_launch_ms'), j.get('roofline',{}).get('alone'), 'auc', j['auc'])
PY
