#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multiproc.py -x -q -k "bench_world2 or plain_bench" 2>&1 | tail -30
