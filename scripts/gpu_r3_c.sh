cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3c
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider 2>&1 | tail -40
python scripts/diag_10k.py > gpurun_out/r3c/diag_10k.txt 2>&1; tail -20 gpurun_out/r3c/diag_10k.txt
