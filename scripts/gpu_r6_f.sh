#!/bin/bash
# round 6: the KL step with the hardware exp -- parity tests, kernel table, bench; the masked / sharded test fixes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out/r6
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fused_mse or sharded_mse or fused_elementwise or refuses_kde or gram_kx or small_operand" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_multiproc.py -q -m gpu -k "production or kl" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_readme.py -q -m gpu -k "kl" 2>&1 | tail -4
for wl in synthetic-10k-kl cora-shape-kl; do
  timeout 300 python bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r6/bench_$wl.json 2> gpurun_out/r6/bench_$wl.err
  python -c "import json; d=json.load(open('gpurun_out/r6/bench_$wl.json')); print('$wl', d['value'], d['ms_per_step'], d['config'].get('fused_steps'), (d.get('roofline') or {}).get('frac'))"
done
cd /tmp
rm -rf "$R/gpurun_out/r6/stats_synthetic-10k-kl"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r6/stats_synthetic-10k-kl" -- \
  python3 "$R/bench.py" --workload synthetic-10k-kl --steps 40 --warmup 10 --no-cpu-baseline --no-split-probe > /dev/null 2>&1
cd "$R"; python3 scripts/kstats.py gpurun_out/r6/stats_synthetic-10k-kl 50 8
find gpurun_out/r6 -name "*kernel_trace*" -size +8M -delete 2>/dev/null
