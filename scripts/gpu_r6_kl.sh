#!/bin/bash
# round 6: the fused KL step -- parity tests (fused vs general vs oracle, free run, row-block ranks in lockstep) + its bench workloads
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_mse or sharded_mse" > gpurun_out/r6/kl_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6/kl_tests.log
tail -15 gpurun_out/r6/kl_tests.log
for wl in synthetic-10k-kl cora-shape-kl synthetic-10k-mse; do
  timeout 300 python bench.py --workload $wl --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r6/bench_$wl.json 2> gpurun_out/r6/bench_$wl.err
  echo "$wl rc=$?"; head -c 600 gpurun_out/r6/bench_$wl.json; echo
done
