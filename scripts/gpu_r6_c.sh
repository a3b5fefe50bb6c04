#!/bin/bash
# round 6: fused KL parity, the single-plane product mode (test + accuracy table), the multi-process suite
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_mse or sharded_mse or kde_columns or single_plane or split_bf16" > gpurun_out/r6/kl_tests.log 2>&1
echo "parity rc=$?" | tee -a gpurun_out/r6/kl_tests.log
tail -5 gpurun_out/r6/kl_tests.log
timeout 900 python scripts/single_plane_table.py > gpurun_out/r6/single_plane_table.txt 2> gpurun_out/r6/single_plane_table.err
echo "table rc=$?"; cat gpurun_out/r6/single_plane_table.txt; tail -5 gpurun_out/r6/single_plane_table.err
timeout 2400 python -m pytest tests/test_gpu_multiproc.py -q -m gpu > gpurun_out/r6/mp_tests.log 2>&1
echo "multiproc rc=$?" | tee -a gpurun_out/r6/mp_tests.log
tail -30 gpurun_out/r6/mp_tests.log
