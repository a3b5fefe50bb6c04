#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4h; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "masked or fused or early or falls_back" > $O/parity.txt 2>&1; tail -30 $O/parity.txt
