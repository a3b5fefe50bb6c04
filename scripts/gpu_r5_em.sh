#!/bin/bash
# GPU box (round 5): per-rank compute of the configurations BASELINE.json puts on 4 and 8 GPUs.
#   10k: lockstep emulation at world 1/2/4/8 (all ranks on one GPU) and the echo mode at 4/8 (one rank alone) -- the two must agree
#   30k 3-layer: echo at world 1/2/4/8 (58 GB per engine: the ranks do not fit side by side)
# + kernel tables of one rank's step (echo) at 10k/4, 10k/8, 30k/4, 30k/8
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
python3 scripts/shard_emulate.py --worlds 1,2,4,8 --steps 6 > gpurun_out/r05_em_10k_lockstep.log 2>&1; tail -1 gpurun_out/r05_em_10k_lockstep.log | cut -c1-300
python3 scripts/shard_emulate.py --echo --worlds 1,2,4,8 --steps 10 > gpurun_out/r05_em_10k_echo.log 2>&1; grep '^{"world"' gpurun_out/r05_em_10k_echo.log | cut -c1-400
python3 scripts/shard_emulate.py --echo --workload synthetic-30k-hsic-3layer --worlds 1,2,4,8 --steps 4 > gpurun_out/r05_em_30k_echo.log 2>&1; grep '^{"world"' gpurun_out/r05_em_30k_echo.log | cut -c1-400
for cfg in "synthetic-10k-hsic 4" "synthetic-10k-hsic 8" "synthetic-30k-hsic-3layer 4" "synthetic-30k-hsic-3layer 8"; do
  set -- $cfg
  tag="r05_em_${1}_w${2}"
  (cd /tmp; rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/$tag" -- python3 "$R/scripts/shard_emulate.py" --echo --workload $1 --worlds $2 --steps 4 > "$R/gpurun_out/$tag.log" 2>&1)
  T=$(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1)
  python3 scripts/echo_trace_summary.py "$T" 4 > gpurun_out/${tag}_kernels.txt 2>&1
  rm -rf gpurun_out/$tag
  head -30 gpurun_out/${tag}_kernels.txt
done
