#!/bin/bash
# kernel-trace of the bench, product serialised (MCGRA_OVERLAP=0) and default; TAG names the output
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
export MCGRA_AB=1 MCGRA_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_serial" -o ks -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/${TAG}_prof_serial.log" 2>&1
unset MCGRA_OVERLAP
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof" -o ks -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/${TAG}_prof.log" 2>&1
cd "$R"
find gpurun_out/${TAG}_prof gpurun_out/${TAG}_prof_serial -name "*kernel_trace*" -size +8M -delete 2>/dev/null
for d in ${TAG}_prof_serial ${TAG}_prof; do echo "== $d"; f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); head -22 "$f" | cut -d, -f1-5 | cut -c1-150; done
