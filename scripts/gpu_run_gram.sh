#!/bin/bash
# Gram-evaluation steps on the split kernel: parity tests that walk the Gram path, then the bench line of that path
TAG=${TAG:-gram}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -k "gram or masked or general or citeseer or gat or lowrank or 30k or config4" > gpurun_out/${TAG}_pytest.log 2>&1
tail -5 gpurun_out/${TAG}_pytest.log
MCGRA_NO_LOWRANK=1 timeout 600 python bench.py --no-cpu-baseline --no-split-probe --steps 10 > gpurun_out/${TAG}_bench_gram.json 2> gpurun_out/${TAG}_bench_gram.err
cat gpurun_out/${TAG}_bench_gram.json
MCGRA_NO_LOWRANK=1 MCGRA_GRAM_SPLIT=0 timeout 600 python bench.py --no-cpu-baseline --no-split-probe --steps 5 > gpurun_out/${TAG}_bench_gram_f32.json 2> gpurun_out/${TAG}_bench_gram_f32.err
cat gpurun_out/${TAG}_bench_gram_f32.json
