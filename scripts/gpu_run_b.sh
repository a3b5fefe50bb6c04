#!/bin/bash
# one GPU-box visit: full GPU suite, Gram-path bench line, kernel trace of the Cora-shape step
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export MCGRA_REPORT_DIR="$GRAFT_REPO_ROOT/gpurun_out"
python -m pytest tests -m gpu -q --tb=short --maxfail=30 > gpurun_out/${TAG}_pytest.log 2>&1
tail -5 gpurun_out/${TAG}_pytest.log
MCGRA_NO_LOWRANK=1 python bench.py --no-cpu-baseline --no-split-probe --steps 10 > gpurun_out/${TAG}_bench_gram.json 2> gpurun_out/${TAG}_bench_gram.err
python bench.py --workload cora-shape-hsic --no-cpu-baseline --no-split-probe --steps 200 --warmup 10 > gpurun_out/${TAG}_bench_cora.json 2> gpurun_out/${TAG}_bench_cora.err
tail -c 400 gpurun_out/${TAG}_bench_cora.json
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_prof_cora" -o kt -- python3 "$R/bench.py" --workload cora-shape-hsic --steps 20 --warmup 5 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/${TAG}_prof_cora.log" 2>&1
cd "$R"
ls -la gpurun_out/${TAG}_prof_cora* | head
