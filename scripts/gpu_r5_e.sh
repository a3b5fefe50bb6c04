#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "gram or cka or masked or falls_back or lowrank" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_citeseer.py -x -q 2>&1 | tail -5
for ovl in 1 0; do
  MCGRA_GRAM_OVERLAP=$ovl MCGRA_NO_LOWRANK=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-split-probe > gpurun_out/r05e_gram_ovl$ovl.json 2> gpurun_out/r05e_gram_ovl$ovl.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r05e_gram_ovl$ovl.json')); print('overlap $ovl', d['value'], d['ms_per_step'], d['auc'], d['config']['gram_split_steps'])"
done
TAG=r05e
cd /tmp
MCGRA_NO_LOWRANK=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_gram" -- python3 "$R/bench.py" --steps 8 --warmup 3 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/${TAG}_gram.log" 2>&1
cd "$R"
python3 scripts/kstats.py gpurun_out/${TAG}_gram 11 30 > gpurun_out/${TAG}_gram_kstats.txt 2>&1
T=$(find gpurun_out/${TAG}_gram -name "*kernel_trace.csv" | head -1)
python3 scripts/gram_timeline.py "$T" > gpurun_out/${TAG}_gram_timeline.txt 2>&1
find gpurun_out/${TAG}_gram -name "*kernel_trace*" -size +8M -delete 2>/dev/null
head -70 gpurun_out/${TAG}_gram_timeline.txt
