#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for w in cora-shape-hsic cora-shape-mse citeseer-shape-gat-hsic synthetic-10k-mse synthetic-4k-hsic synthetic-30k-hsic-3layer; do
  st=5; case $w in cora-*|citeseer-*|synthetic-4k-*) st=100;; synthetic-10k-*) st=20;; esac      # (short steps: a 5-step run times the clock ramp)
  python bench.py --workload $w --no-cpu-baseline --no-split-probe --steps $st --warmup 5 > gpurun_out/${TAG}_bench_$w.json 2>/dev/null
  tail -c 200 gpurun_out/${TAG}_bench_$w.json | head -c 10 >/dev/null
done
MCGRA_AB=1 MCGRA_NO_LOWRANK=1 python bench.py --no-cpu-baseline --no-split-probe --steps 10 > gpurun_out/${TAG}_bench_gram_path.json 2>/dev/null
MCGRA_SPLIT_BF16=2 python bench.py --no-cpu-baseline --no-split-probe > gpurun_out/${TAG}_bench_split_bf16x3.json 2>/dev/null
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % os.environ["TAG"])):
    try:
        b = json.load(open(f)); print(os.path.basename(f), round(b["value"], 2), round(b["ms_per_step"], 3))
    except Exception as e:
        print(f, "ERR", e)
PY
