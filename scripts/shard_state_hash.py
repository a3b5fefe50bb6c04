"""GPU box: hash of the adjacency after `steps` row-block sharded steps (+ monitoring forwards) of a bench workload, `world` ranks in
lockstep on one GPU -- the determinism screen of the sharded step (product forked by the forward, cut product beside the all-to-all,
decode / small terms on side streams): every run must print the same hash.
    python scripts/shard_state_hash.py <workload> <world> [steps]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mcgra_loader
pkg = mcgra_loader.load()
import bench
from mc_gra_amd import sharded as S

wl, world, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda:0")
n = bench.WORKLOADS[wl][0]
plans = [S.RowBlockPlan(n, world, r) for r in range(world)]
bks = S.lockstep_backends([bench.build_engine(pkg, torch, dev, wl, 0, plan=p)[0] for p in plans], plans)
for i in range(steps):
    S.run_lockstep(bks, S.SHARD_STEP)
    S.run_lockstep(bks, S.SHARD_MONITOR_LAST if i == steps - 1 else S.SHARD_MONITOR)
torch.cuda.synchronize()
rows = torch.cat([b.eng.get_rows() for b in bks if b.plan.has_rows], 0).cpu().numpy()
print(wl, f"world{world}", steps, hashlib.sha256(rows.tobytes()).hexdigest()[:16], "fused", [b.eng.fused_steps() for b in bks])
