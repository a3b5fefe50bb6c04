"""Per-phase time of ONE rank of a row-block sharded step, measured on a single GPU (no exchange; rows owned
by other ranks hold stale data, which does not change the work done).  Usage: shard_emulate.py [world] [rank ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mcgra_loader
pkg = mcgra_loader.load()
import bench
from mc_gra_amd.sharded import RowBlockPlan, HipShardBackend

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ranks = [int(x) for x in sys.argv[2:]] or [0, world - 1]
dev = torch.device("cuda:0")
wl = "synthetic-10k-hsic"
n = bench.WORKLOADS[wl][0]
res = {}
for r in ranks:
    plan = RowBlockPlan(n, world, r)
    eng, _, _ = bench.build_engine(pkg, torch, dev, wl, 0, row_begin=plan.row_begin, row_end=plan.row_end)
    b = HipShardBackend(eng, plan)
    for name in ("KX", "KY", "G_adjn", "G_A1"):
        b.exchanged[name].zero_()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    acc = [0.0] * 4
    for it in range(6):
        for k in range(4):
            ev[k].record()
            b.phase(k)
        ev[4].record()
        torch.cuda.synchronize()
        if it >= 2:
            for k in range(4):
                acc[k] += ev[k].elapsed_time(ev[k + 1]) / 4
    res[f"rank{r}"] = {"rows": [plan.row_begin, min(plan.row_end, n)], "phase_ms": [round(x, 3) for x in acc],
                       "total_ms": round(sum(acc), 3)}
    del eng, b
    torch.cuda.empty_cache()
blk = plan.rows_per_rank * eng_ld if (eng_ld := ((n + 3) // 4) * 4) else 0
res["exchange_bytes_per_rank_per_step"] = 4 * 4 * plan.rows_per_rank * eng_ld * (world - 1)
print(json.dumps({"world": world, **res}))
