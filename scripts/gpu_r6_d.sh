#!/bin/bash
# round 6: bit-identity of the new overlaps (early Kx, small terms beside the general step), the sharded KL case, configs[2] bench + timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out/r6
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded_mse or small_operand or gram_kx or gram_products or fused_tail or step_gradients" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_citeseer.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --workload citeseer-shape-gat-hsic --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r6/bench_cit_$i.json 2> gpurun_out/r6/bench_cit_$i.err
python -c "import json; d=json.load(open('gpurun_out/r6/bench_cit_$i.json')); print('citeseer-shape-gat-hsic', d['value'], d['ms_per_step'])"
MCGRA_AB=1 MCGRA_SMALL_SIDE=0 timeout 300 python bench.py --workload citeseer-shape-gat-hsic --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r6/bench_cit_noside_$i.json 2> /dev/null
python -c "import json; d=json.load(open('gpurun_out/r6/bench_cit_noside_$i.json')); print('  small terms on the caller stream:', d['value'], d['ms_per_step'])"
done
cd /tmp
rm -rf "$R/gpurun_out/r6/cit_trace"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r6/cit_trace" -- \
  python3 "$R/bench.py" --workload citeseer-shape-gat-hsic --steps 12 --warmup 4 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/r6/cit_trace.log" 2>&1
cd "$R"
T=$(find gpurun_out/r6/cit_trace -name "*kernel_trace.csv" | head -1)
python3 scripts/general_step_timeline.py "$T" > gpurun_out/r6/citeseer_gat_step_timeline.txt 2>&1
find gpurun_out/r6 -name "*kernel_trace*" -size +8M -delete 2>/dev/null
head -100 gpurun_out/r6/citeseer_gat_step_timeline.txt
