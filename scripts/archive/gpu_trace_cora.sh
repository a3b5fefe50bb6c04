#!/bin/bash
# kernel trace of the Cora-shape step (timeline analysis: scripts/trace_timeline.py)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_cora" -o kt -- python3 "$R/bench.py" --workload cora-shape-hsic --steps 20 --warmup 5 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/${TAG}_prof_cora.log" 2>&1
