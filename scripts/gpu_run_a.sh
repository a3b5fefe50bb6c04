#!/bin/bash
# one GPU-box visit: full GPU suite, bench line, kernel-trace profiles (side stream on / off)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --tb=short --maxfail=30 > gpurun_out/${TAG}_pytest.log 2>&1
tail -5 gpurun_out/${TAG}_pytest.log
python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/${TAG}_prof" -o ks -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/${TAG}_prof.log" 2>&1
export MCGRA_OVERLAP=0
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/${TAG}_prof_serial" -o ks -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/${TAG}_prof_serial.log" 2>&1
unset MCGRA_OVERLAP
cd "$R"
find gpurun_out/${TAG}_prof gpurun_out/${TAG}_prof_serial -name "*kernel_trace*" -size +20M -delete 2>/dev/null
ls -la gpurun_out/${TAG}_prof* | head
