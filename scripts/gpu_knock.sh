#!/bin/bash
mkdir -p gpurun_out/knock
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multiproc.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/knock/tests.txt
{
for i in 1 2; do
for k in 0 1 2 4 8 15; do
MCGRA_KNOCK=$k python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-split-probe | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('knock $k', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3), d['config']['fused_steps'])"
done; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/knock/ab.txt
cat gpurun_out/knock/tests.txt
