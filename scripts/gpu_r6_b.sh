#!/bin/bash
# round 6: fused KL parity (single process + multi-process), KDE guard, the extended multi-process suite
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_mse or sharded_mse or kde_columns" > gpurun_out/r6/kl_tests.log 2>&1
echo "parity rc=$?" | tee -a gpurun_out/r6/kl_tests.log
tail -5 gpurun_out/r6/kl_tests.log
timeout 2400 python -m pytest tests/test_gpu_multiproc.py -x -q -m gpu > gpurun_out/r6/mp_tests.log 2>&1
echo "multiproc rc=$?" | tee -a gpurun_out/r6/mp_tests.log
tail -30 gpurun_out/r6/mp_tests.log
