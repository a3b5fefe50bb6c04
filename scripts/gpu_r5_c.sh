#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "kde or mutual_information" 2>&1 | tail -30
timeout 900 python -m pytest tests/test_gpu_readme.py -x -q -k "kde" 2>&1 | tail -5
