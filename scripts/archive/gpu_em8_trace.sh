#!/bin/bash
# kernel trace of the world-8 lockstep emulation (per-rank compute of the sharded step): where a rank's 1.3 ms go
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/em8" -- python3 scripts/shard_emulate.py --worlds 8 --steps 6 > "$R/gpurun_out/em8.log" 2>&1
find "$R/gpurun_out/em8" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$R/gpurun_out/em8_kernel_stats.csv"
find "$R/gpurun_out/em8" -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} "$R/gpurun_out/em8_kernel_trace.csv"
rm -rf "$R/gpurun_out/em8"
tail -3 "$R/gpurun_out/em8.log"
