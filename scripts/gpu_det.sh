#!/bin/bash
# determinism screen of the overlapped step (early tail pass, cut product, side streams): the same state hash from every run
mkdir -p gpurun_out/det
{
for i in $(seq 1 12); do python scripts/state_hash.py synthetic-10k-hsic 12; done
MCGRA_AB=1 MCGRA_EARLY_TAIL=0 python scripts/state_hash.py synthetic-10k-hsic 12
MCGRA_AB=1 MCGRA_OVERLAP=0 python scripts/state_hash.py synthetic-10k-hsic 12
for i in $(seq 1 6); do python scripts/state_hash.py synthetic-4k-hsic 30; done
for i in $(seq 1 6); do python scripts/state_hash.py cora-shape-hsic 50; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/det/hashes.txt | awk '{print $1, $2, $3}' | sort | uniq -c
