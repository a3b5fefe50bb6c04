#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu --deselect tests/test_gpu_readme.py::test_readme_line_at_the_20_epoch_horizon 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -12
