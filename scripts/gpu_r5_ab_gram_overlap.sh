#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "gram or cka" 2>&1 | tail -4
for ovl in 1 0 1 0; do
  MCGRA_AB=1 MCGRA_GRAM_OVERLAP=$ovl MCGRA_NO_LOWRANK=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-split-probe > gpurun_out/r05f_gram_ovl$ovl.json 2> gpurun_out/r05f_gram_ovl$ovl.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r05f_gram_ovl$ovl.json')); print('overlap $ovl', d['value'], d['ms_per_step'], d['auc'], d['config']['gram_split_steps'])"
done
tag=r05f_em_w8
(cd /tmp; rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/$tag" -- python3 "$R/scripts/shard_emulate.py" --echo --workload synthetic-10k-hsic --worlds 8 --steps 4 > "$R/gpurun_out/$tag.log" 2>&1)
T=$(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1)
python3 scripts/echo_trace_summary.py "$T" 4 --timeline > gpurun_out/${tag}_kernels.txt 2>&1
rm -rf gpurun_out/$tag
grep -A130 "last step" gpurun_out/${tag}_kernels.txt
