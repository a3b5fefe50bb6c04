#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
(time python -m pytest tests/ -q -m gpu 2>&1 | grep -v "^\[W\|Gloo\|amdgpu.ids\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -25) 2>&1
