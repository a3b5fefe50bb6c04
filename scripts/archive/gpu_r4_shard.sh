#!/bin/bash
# row-block ranks, round 4 (second half): tests of the cut product, the emulation profile and a kernel table of a world-8 rank
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/shard
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multiproc.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 > $R/gpurun_out/shard/tests.txt
python scripts/shard_emulate.py --worlds 1,2,4,8 --steps 10 2>/dev/null | grep '^{' > $R/gpurun_out/shard/emulate.json.log
cd /tmp; export TMPDIR=/tmp; cd $R
rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/shard/em8" -- python3 scripts/shard_emulate.py --worlds 8 --steps 6 > "$R/gpurun_out/shard/em8.log" 2>&1
T=$(find "$R/gpurun_out/shard/em8" -name "*kernel_trace.csv" | head -1)
python scripts/em_trace_summary.py "$T" 8 6 > $R/gpurun_out/shard/em8_kernels.txt
rm -rf "$R/gpurun_out/shard/em8"
cat $R/gpurun_out/shard/tests.txt; cut -c1-200 $R/gpurun_out/shard/emulate.json.log | head -4; head -8 $R/gpurun_out/shard/em8_kernels.txt
