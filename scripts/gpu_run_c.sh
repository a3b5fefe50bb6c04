#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hsic_py" 2>&1 | grep -E "passed|failed|FAILED|^tests|^E " | tail -8
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_cora" -o kt -- python3 "$R/bench.py" --workload cora-shape-hsic --steps 20 --warmup 5 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/${TAG}_prof_cora.log" 2>&1
ls $R/gpurun_out/${TAG}_prof_cora | head
