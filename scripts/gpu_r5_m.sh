#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
python3 scripts/shard_emulate.py --echo --worlds 4,8 --steps 40 2>&1 | grep '^{"world"' | cut -c1-330
python3 scripts/shard_emulate.py --echo --workload synthetic-10k-mse --worlds 8 --steps 40 2>&1 | grep '^{"world"' | cut -c1-330
