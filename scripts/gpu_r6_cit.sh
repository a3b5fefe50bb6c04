#!/bin/bash
# round 6: configs[2]'s general step after a change -- the general-step / Gram / README parity tests, the bench line, one step's launch list
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out/r6
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_readme.py -x -q -m gpu -k "${1:-gat or gram or general or citeseer or readme or sage or cka or small_operand or kx_forked}" 2>&1 | tail -5
for k in 1 2; do
timeout 300 python bench.py --workload citeseer-shape-gat-hsic --steps 200 --warmup 20 --no-cpu-baseline --no-split-probe 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('citeseer-shape-gat-hsic ms/step', d['ms_per_step'])"
done
MCGRA_AB=1 MCGRA_NO_LOWRANK=1 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-split-probe 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gram evaluation 10k ms/step', d['ms_per_step'])"
cd /tmp
rm -rf "$R/gpurun_out/r6/cit_trace"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r6/cit_trace" -- \
  python3 "$R/bench.py" --workload citeseer-shape-gat-hsic --steps 12 --warmup 4 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/r6/cit_trace.log" 2>&1
cd "$R"
T=$(find gpurun_out/r6/cit_trace -name "*kernel_trace.csv" | head -1)
python3 scripts/general_step_timeline.py "$T" > gpurun_out/r6/citeseer_gat_step_timeline_b.txt 2>&1
find gpurun_out/r6 -name "*kernel_trace*" -size +8M -delete 2>/dev/null
head -100 gpurun_out/r6/citeseer_gat_step_timeline_b.txt | cut -c1-110
