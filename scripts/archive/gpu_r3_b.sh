# tests (minus the 10k reference test while its float64 fixture is regenerated) + bench + serial kernel trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3b
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fullsize.py::test_bench_workload_matches_reference_at_10k --deselect tests/test_gpu_fullsize.py::test_mutations_turn_the_10k_reference_test_red -p no:cacheprovider 2>&1 | tail -15
python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r3b/bench10k.json 2> gpurun_out/r3b/bench10k.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r3b/bench10k.json'))
print('10k', round(j['value'],2), round(j['ms_per_step'],3), j['config'].get('fused_steps'), j['config'].get('general_steps'), j.get('roofline',{}).get('avg_launch_ms'), j.get('roofline',{}).get('alone'), 'auc', j['auc'], 'cora', j.get('other_workloads'))
PY
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
MCGRA_AB=1 MCGRA_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r3b/prof_serial" -o ks -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/r3b/prof_serial.log" 2>&1
cd "$R"; find gpurun_out/r3b -name "*kernel_trace*" -size +8M -delete 2>/dev/null
python scripts/kstats.py gpurun_out/r3b/prof_serial 8 16 | grep -v "rocprim\|at::native"
