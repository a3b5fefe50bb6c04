#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=r05j
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_mse" -- python3 "$R/bench.py" --workload synthetic-10k-mse --steps 40 --warmup 10 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/${TAG}_mse.log" 2>&1
cd "$R"
python3 scripts/kstats.py gpurun_out/${TAG}_mse 51 30 > gpurun_out/${TAG}_mse_kstats.txt 2>&1
T=$(find gpurun_out/${TAG}_mse -name "*kernel_trace.csv" | head -1)
python3 scripts/gram_timeline.py "$T" > gpurun_out/${TAG}_mse_timeline.txt 2>&1
find gpurun_out/${TAG}_mse -name "*kernel_trace*" -size +8M -delete 2>/dev/null
tail -1 gpurun_out/${TAG}_mse.log | cut -c1-300
head -32 gpurun_out/${TAG}_mse_kstats.txt
head -60 gpurun_out/${TAG}_mse_timeline.txt
