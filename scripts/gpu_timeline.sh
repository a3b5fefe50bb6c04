#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp; cd $R
rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/tl" -- python3 bench.py --steps 60 --warmup 40 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/tl.log" 2>&1
T=$(find "$R/gpurun_out/tl" -name "*kernel_trace.csv" | head -1)
python scripts/step_timeline.py "$T" > $R/gpurun_out/step_timeline.txt
cp "$T" $R/gpurun_out/tl_kernel_trace.csv
rm -rf "$R/gpurun_out/tl"
cat $R/gpurun_out/step_timeline.txt
