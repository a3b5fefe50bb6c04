#!/bin/bash
# round 4, call c: README eps lines on the GPU, ld alignment A/B, kernel trace of the world-8 emulation
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_readme.py -q -m gpu -k "eps" > $O/readme_eps.txt 2>&1; tail -5 $O/readme_eps.txt
for rep in 1 2; do
  for al in 4 32; do
    MCGRA_LD_ALIGN=$al python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-split-probe 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ld_align $al rep $rep', round(l['value'],2), round(l['ms_per_step'],4), 'product', round(l['roofline']['avg_launch_ms'],4), 'alone', round(l['roofline']['alone']['avg_launch_ms'],4))" | tee -a $O/ld_align.txt
  done
done
MCGRA_LD_ALIGN=32 timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "matches_reference or invariants" > $O/ld32_tests.txt 2>&1; tail -3 $O/ld32_tests.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/emu8 -o emu8 -- python3 $R/scripts/shard_emulate.py --worlds 8 --steps 4 > $O/emu8.log 2>&1
MCGRA_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o ks -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-split-probe > $O/serial.log 2>&1
ls $O/emu8 $O/serial
