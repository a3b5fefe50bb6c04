#!/bin/bash
# round 6: one rank of world 8 at N = 10 000 (echo emulation) launch by launch -- where its ~1.09 ms go
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
python3 scripts/shard_emulate.py --echo --worlds 8 --steps 40 2>&1 | grep '^{"world"' | cut -c1-260
tag=r6/em_w8
rm -rf gpurun_out/$tag
(cd /tmp; rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/$tag" -- python3 "$R/scripts/shard_emulate.py" --echo --workload synthetic-10k-hsic --worlds 8 --steps 4 > "$R/gpurun_out/$tag.log" 2>&1)
T=$(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1)
python3 scripts/echo_trace_summary.py "$T" 4 --timeline > gpurun_out/r6/em_w8_kernels.txt 2>&1
rm -rf gpurun_out/$tag
cat gpurun_out/r6/em_w8_kernels.txt | cut -c1-150
