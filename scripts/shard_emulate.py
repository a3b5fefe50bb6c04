"""Per-rank compute time of the row-block sharded step, measured on ONE GPU: `world` row-block engines of the bench
workload advance in lockstep (sharded.run_lockstep: the collectives are device copies between their arenas), so the
wall time of a step is the SUM over ranks of their compute (+ the copies); divided by `world` it is what one rank of
a real job spends computing per step -- the part of a multi-GPU step that is not RCCL time.  Also splits the time of
rank 0 into sharded N x N passes and replicated node-level work by running the same step with world = 1.

    python scripts/shard_emulate.py [--workload synthetic-10k-hsic] [--worlds 1,2,4,8] [--steps 6]
    python scripts/shard_emulate.py --echo --workload synthetic-30k-hsic-3layer --worlds 4,8
--echo: ONE rank of each world by itself with its collectives answered by its own data (sharded.run_echo) -- the launches
of a real rank's step, for configurations whose `world` engines do not fit one GPU side by side (N = 30 000: 58 GB each).
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mcgra_loader
pkg = mcgra_loader.load()
import bench
from mc_gra_amd import sharded as S

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="synthetic-10k-hsic")
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--echo", action="store_true")
ap.add_argument("--rank", type=int, default=-1, help="--echo: which rank (default: the middle one)")
a = ap.parse_args()
dev = torch.device("cuda:0")
n = bench.WORKLOADS[a.workload][0]
out = {"workload": a.workload, "what": __doc__.split("\n\n")[0], "rows": []}
for world in [int(x) for x in a.worlds.split(",")]:
    if a.echo:
        rk = a.rank if a.rank >= 0 else world // 2
        plan = S.RowBlockPlan(n, world, rk)
        eng = bench.build_engine(pkg, torch, dev, a.workload, 0, plan=plan)[0]
        bk = S.HipShardBackend(eng, plan)
        # (a rank's monitoring forward forks the NEXT step's product: the last warm-up and the last timed iteration say that no step
        # follows -- as PGDAttack.attack and bench.py do -- so the timed region holds exactly a.steps products)
        for i in range(3):
            S.run_echo(bk, S.SHARD_STEP); S.run_echo(bk, S.SHARD_MONITOR_LAST if i == 2 else S.SHARD_MONITOR)
        torch.cuda.synchronize()
        eng.profile(True); eng.gemm_stats(reset=True)
        t0 = time.perf_counter()
        nex = 0
        for i in range(a.steps):
            nex += S.run_echo(bk, S.SHARD_STEP); nex += S.run_echo(bk, S.SHARD_MONITOR_LAST if i == a.steps - 1 else S.SHARD_MONITOR)
        host = (time.perf_counter() - t0) / a.steps          # the host is done enqueueing: if this is the step time, the host binds
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        st = eng.gemm_stats(reset=True)
        row = {"world": world, "rank": rk, "rows_per_rank": plan.rows_per_rank, "mode": "echo (one rank alone, collectives answered by its own data)",
               "per_rank_compute_ms": 1e3 * dt, "host_enqueue_ms": 1e3 * host, "product_ms_per_step": st["ms"] / max(1, a.steps), "collectives_per_step": nex / a.steps,
               "fused_steps": eng.fused_steps(), "general_steps": eng.path_stats()["general_steps"], "steps": a.steps + 3,
               "alltoall_bytes_per_rank": 4 * world * plan.rows_per_rank ** 2 if world > 1 else 0,
               "allgather_node_bytes_per_rank": plan.n_pad * 64 * 4}
        out["rows"].append(row)
        print(json.dumps(row), flush=True)
        del bk, eng
        torch.cuda.empty_cache()
        continue
    plans = [S.RowBlockPlan(n, world, r) for r in range(world)]
    bks = S.lockstep_backends([bench.build_engine(pkg, torch, dev, a.workload, 0, plan=p)[0] for p in plans], plans)
    for i in range(2):
        S.run_lockstep(bks, S.SHARD_STEP); S.run_lockstep(bks, S.SHARD_MONITOR_LAST if i == 1 else S.SHARD_MONITOR)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nex = 0
    for i in range(a.steps):
        nex += S.run_lockstep(bks, S.SHARD_STEP); nex += S.run_lockstep(bks, S.SHARD_MONITOR_LAST if i == a.steps - 1 else S.SHARD_MONITOR)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    # the emulation's own copies (what a real run replaces by RCCL collectives): the same exchanges replayed on the joint
    # arena without any engine work, timed alone
    joint = bks[0].joint
    recorded = []
    orig_next = [b.next for b in bks]
    bks[0].next = lambda: (recorded.append(orig_next[0]()) or recorded[-1])
    S.run_lockstep(bks, S.SHARD_STEP); S.run_lockstep(bks, S.SHARD_MONITOR)
    bks[0].next = orig_next[0]
    exs = [e for e in recorded if e[0] in (S.XCHG_ALLGATHER, S.XCHG_ALLTOALL)]
    def replay():
        w = world
        for kind, count, off, off2, chunk in exs:
            if kind == S.XCHG_ALLGATHER:
                full = joint[:, off:off + w * chunk].view(w, w, chunk)
                own = torch.diagonal(full, dim1=0, dim2=1).t().clone()
                full.copy_(own.unsqueeze(0).expand(w, w, chunk))
            else:
                send = joint[:, off:off + w * chunk].view(w, w, chunk)
                joint[:, off2:off2 + w * chunk].view(w, w, chunk).copy_(send.transpose(0, 1))
    replay(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        replay()
    torch.cuda.synchronize()
    copies_ms = 1e3 * (time.perf_counter() - t1) / 20
    row = {"world": world, "rows_per_rank": plans[0].rows_per_rank, "sum_over_ranks_ms": 1e3 * dt,
           "per_rank_compute_ms": 1e3 * dt / world, "collectives_per_step": nex / a.steps,
           "emulation_copies_ms_per_step": copies_ms,
           "per_rank_compute_ms_without_emulation_copies": (1e3 * dt - copies_ms) / world,
           "alltoall_bytes_per_rank": 4 * world * plans[0].rows_per_rank ** 2 if world > 1 else 0,
           "allgather_node_bytes_per_rank": plans[0].n_pad * 64 * 4}
    out["rows"].append(row)
    print(json.dumps(row), flush=True)
    del bks
    torch.cuda.empty_cache()
print(json.dumps(out))
