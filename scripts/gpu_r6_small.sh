#!/bin/bash
# round 6: launch lists of the fused steps at Cora size (configs[0] / [1] shapes)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out/r6
export TMPDIR=/tmp
for wl in cora-shape-hsic cora-shape-mse; do
cd /tmp; rm -rf "$R/gpurun_out/r6/tr_$wl"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/r6/tr_$wl" -- \
  python3 "$R/bench.py" --workload $wl --steps 30 --warmup 10 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/r6/tr_$wl.log" 2>&1
cd "$R"
T=$(find gpurun_out/r6/tr_$wl -name "*kernel_trace.csv" | head -1)
python3 scripts/general_step_timeline.py "$T" 3 k_tail_adam > gpurun_out/r6/${wl}_step_timeline.txt 2>&1
rm -rf gpurun_out/r6/tr_$wl
cat gpurun_out/r6/${wl}_step_timeline.txt | cut -c1-120
done
