#!/bin/bash
# GPU box (round 5): the tests this round added or touched
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multiproc.py -x -q -k "class_shards or main_entry_under" 2>&1 | tail -30
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "parity_builds_bits or defect_injector or mutations_turn" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gram_evaluation_on_the_split or split or cka_steps" 2>&1 | tail -15
