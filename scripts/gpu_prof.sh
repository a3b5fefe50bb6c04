#!/bin/bash
# kernel-trace profiles of the bench (side stream on / off); TAG names the output
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof" -o ks -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/${TAG}_prof.log" 2>&1
export MCGRA_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_serial" -o ks -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/${TAG}_prof_serial.log" 2>&1
unset MCGRA_OVERLAP
ls "$R/gpurun_out/${TAG}_prof" "$R/gpurun_out/${TAG}_prof_serial"
