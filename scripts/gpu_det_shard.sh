#!/bin/bash
# determinism screen of the row-block sharded step: one hash per (workload, world)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/det
{
for i in $(seq 1 6); do python scripts/shard_state_hash.py synthetic-4k-hsic 4 12; done
for i in $(seq 1 4); do python scripts/shard_state_hash.py synthetic-4k-hsic 8 12; done
for i in $(seq 1 4); do python scripts/shard_state_hash.py synthetic-10k-hsic 2 6; done
for i in $(seq 1 4); do python scripts/shard_state_hash.py cora-shape-mse 3 20; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/det/shard_hashes.txt | awk '{print $1, $2, $3, $4}' | sort | uniq -c
