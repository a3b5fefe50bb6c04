#!/bin/bash
# the kernel-stats pass of gpu_pmc.sh alone (TAG names the round)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
B="$R/bench.py --steps ${STATS_STEPS:-60} --warmup ${STATS_WARMUP:-20} --no-cpu-baseline --no-split-probe"
rm -rf "$R/gpurun_out/${TAG}_stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_stats" -- python3 $B > "$R/gpurun_out/${TAG}_stats.log" 2>&1
cd "$R"
find gpurun_out/${TAG}_stats -name "*kernel_trace*" -size +8M -delete 2>/dev/null
tail -2 gpurun_out/${TAG}_stats.log | cut -c1-400
