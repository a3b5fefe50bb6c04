#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python3 scripts/diag_readme_horizon.py 2>/dev/null > gpurun_out/r05_readme_horizon20.txt; cat gpurun_out/r05_readme_horizon20.txt
