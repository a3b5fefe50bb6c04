#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x -k "fused or tail or symmetric or determin or lowrank or sharded" 2>&1 | grep -E "passed|failed|FAILED|^tests" | tail -4
TAG=$TAG bash scripts/gpu_prof.sh > /dev/null 2>&1
