#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "mse" 2>&1 | grep -v "^\[W\|Gloo\|amdgpu.ids\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40
