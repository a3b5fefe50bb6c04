#!/bin/bash
# round 6: symmetry diagnosis of the fused KL step + the KDE guard test
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6
python scripts/diag_kl_sym.py 1100 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kde_columns or kde_steps or ab_switches" 2>&1 | tail -15
