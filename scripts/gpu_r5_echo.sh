#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
python3 scripts/shard_emulate.py --echo --worlds 1,2,4,8 --steps 20 2>&1 | grep '^{"world"' | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('echo 10k world', d['world'], round(d['per_rank_compute_ms'],3), 'product', round(d['product_ms_per_step'],3), 'fused', d['fused_steps'], d['general_steps'])
"
python3 scripts/shard_emulate.py --echo --workload synthetic-30k-hsic-3layer --worlds 8 --steps 4 2>&1 | grep '^{"world"' | cut -c1-200
for wl in cora-shape-hsic synthetic-4k-hsic; do python3 bench.py --workload $wl --steps 200 --warmup 30 --no-cpu-baseline --no-split-probe 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'], d['ms_per_step'], d['auc'])
"; done
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-split-probe 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'], d['ms_per_step'], d['auc'])
"
tag=r05g_em_w8
(cd /tmp; rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/$tag" -- python3 "$R/scripts/shard_emulate.py" --echo --workload synthetic-10k-hsic --worlds 8 --steps 4 > "$R/gpurun_out/$tag.log" 2>&1)
T=$(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1)
python3 scripts/echo_trace_summary.py "$T" 4 --timeline > gpurun_out/${tag}_kernels.txt 2>&1
rm -rf gpurun_out/$tag
head -3 gpurun_out/${tag}_kernels.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "shard or fused or cut_product or early" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_multiproc.py -x -q -k "row_block" 2>&1 | tail -4
