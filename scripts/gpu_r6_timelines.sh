#!/bin/bash
# round 6: launch lists of one general-path step (configs[2]'s shape: Citeseer-sized GAT, HSIC) and of one Gram-evaluation step at
# N = 10 000 from rocprofv3 kernel traces (no counters) -> gpurun_out/r6/*_timeline.txt; the sharded-KL test; the raster A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out/r6
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded_mse" 2>&1 | tail -3
cd /tmp
rm -rf "$R/gpurun_out/r6/cit_trace" "$R/gpurun_out/r6/gram_trace"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r6/cit_trace" -- \
  python3 "$R/bench.py" --workload citeseer-shape-gat-hsic --steps 12 --warmup 4 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/r6/cit_trace.log" 2>&1
MCGRA_AB=1 MCGRA_NO_LOWRANK=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r6/gram_trace" -- \
  python3 "$R/bench.py" --steps 8 --warmup 3 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/r6/gram_trace.log" 2>&1
cd "$R"
T=$(find gpurun_out/r6/cit_trace -name "*kernel_trace.csv" | head -1)
python3 scripts/general_step_timeline.py "$T" > gpurun_out/r6/citeseer_gat_step_timeline.txt 2>&1
T=$(find gpurun_out/r6/gram_trace -name "*kernel_trace.csv" | head -1)
python3 scripts/gram_timeline.py "$T" > gpurun_out/r6/gram_path_timeline.txt 2>&1
find gpurun_out/r6 -name "*kernel_trace*" -size +8M -delete 2>/dev/null
head -100 gpurun_out/r6/citeseer_gat_step_timeline.txt
head -60 gpurun_out/r6/gram_path_timeline.txt
bash scripts/gpu_r6_ab_raster.sh
