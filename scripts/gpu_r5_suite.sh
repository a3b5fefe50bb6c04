#!/bin/bash
# GPU box: the whole -m gpu suite as the driver runs it, then the bench line in the driver's window
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
(time python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^\[W\|Gloo\|amdgpu.ids\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15) 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench20.json 2> gpurun_out/r05_bench20.err
cut -c1-300 gpurun_out/r05_bench20.json
