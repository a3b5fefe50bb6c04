#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -12
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-probe 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 20/5', round(d['value'],2), round(d['ms_per_step'],3), d['auc'])"
