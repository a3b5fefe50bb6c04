"""One row-block rank of ONE attack as its own process, on a GPU it shares with the other ranks.

Each rank builds an AttackEngine(plan=RowBlockPlan(n, world, rank)) + HipShardBackend and drives it with the product's
ShardedStepper over a gloo process group with host-staged arena slices (sharded.run_exchange(host_staged=True): RCCL
refuses two ranks on one device, the protocol, the arena offsets and the engine code are the ones an RCCL run uses).
Started by tests/test_gpu_multiproc.py from a fork server that never touched the GPU.  Writes <out>.rank<k>.npz."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def case_of(spec):
    """The synthetic problem of a spec dict (same generator as the single-process sharding tests)."""
    from tests import helpers as H
    kw = {"weight_param": tuple(spec["weight_param"])} if spec.get("weight_param") else {}
    return H.synthetic_case(spec["n"], 11, tuple(spec["widths"]), 4, seed=spec["seed"], **kw)


def masked_weights(z):
    """Weights whose second-layer bias kills about half of em: the decode masks pairs, the fused step hands over."""
    from tests import helpers as H
    return H.masked_weights(z)


def run_rank(rank, world, port, spec, out):
    try:
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), OMP_NUM_THREADS="2")
        for k, v in spec.get("env", {}).items():
            os.environ[k] = v
        import numpy as np
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        import mcgra_loader
        pkg = mcgra_loader.load()
        from mc_gra_amd import sharded as S
        from tests import helpers as H
        z = case_of(spec)
        plan = S.RowBlockPlan(spec["n"], world, rank)
        eng = H.engine_from(pkg, z, plan=plan)
        masked = spec.get("masked_steps", 0)
        if masked:
            w = masked_weights(z)
            eng.set_model(w.W, w.b, w.Wlin, w.blin, w.Ws)
        st = S.ShardedStepper(S.HipShardBackend(eng, plan), plan, dist=dist, host_staged=True)
        res = {}
        for t in range(spec["steps"]):
            if masked and t == masked:          # back to the original weights: the decode stops masking
                w0 = H.weights_from(z)
                eng.set_model(w0.W, w0.b, w0.Wlin, w0.blin, w0.Ws)
            sc = st.step(want_scalars=True)
            st.monitor()
            res[f"rows{t}"] = eng.get_rows().cpu().numpy()
            res[f"scal{t}"] = np.array([sc[k] for k in ("loss", "c1", "c2", "c6", "c7", "c9", "c10", "nll", "clamp_sum")])
        res["fused_steps"] = eng.fused_steps()
        res["general_steps"] = eng.path_stats()["general_steps"]
        res["exchanges"] = st.exchanges
        res["row_range"] = np.array([plan.row_begin, plan.row_end])
        np.savez(f"{out}.rank{rank}.npz", **res)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        with open(f"{out}.rank{rank}.err", "w") as fh:
            fh.write(traceback.format_exc())
        raise


def run_bench_rank(rank, world, port, argv, out):
    """bench.py's own `world > 1` branch (one sharded attack over the ranks) on a shared GPU: gloo + host staging."""
    try:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), OMP_NUM_THREADS="2", MCGRA_BENCH_SHARED_GPU="1")
        import json
        import bench
        line = bench.main(argv)
        if rank == 0:
            with open(f"{out}.json", "w") as fh:
                json.dump(line, fh)
    except Exception:
        with open(f"{out}.rank{rank}.err", "w") as fh:
            fh.write(traceback.format_exc())
        raise


def run_bench_plain(argv, out, env):
    """Plain `python bench.py --gpus N ...` as a user (or the driver at N = 1's form) types it: NO launcher environment.
    bench.main sees --gpus N > 1 without WORLD_SIZE and starts the ranks itself (bench.launch_ranks: torch.distributed.run
    from this process, which never touched the GPU).  Everything it prints on stdout goes to <out>.stdout."""
    try:
        if ROOT not in sys.path:
            sys.path.insert(0, ROOT)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        fd = os.open(f"{out}.stdout", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        sys.stdout.flush()
        os.dup2(fd, 1)
        import bench
        bench.main(argv)
        sys.stdout.flush()
    except BaseException:
        with open(f"{out}.rank0.err", "w") as fh:
            fh.write(traceback.format_exc())
        raise


def run_workload_rank(rank, world, port, spec, out):
    """One row-block rank of a bench.py WORKLOAD (e.g. the headline synthetic-10k-hsic: slab split-K tail of the product,
    planes_mm, 20-panel row blocks) as its own process: per step the rank's rows of the learnable adjacency (raw float32
    file), the mirrored packed gradient on its rows at the fixture's sampled positions, and the loss terms."""
    try:
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), OMP_NUM_THREADS="2", MCGRA_KEEP_GSYM="1")
        import numpy as np
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        import mcgra_loader
        pkg = mcgra_loader.load()
        import bench
        from mc_gra_amd import sharded as S
        wl = spec["workload"]
        n = bench.WORKLOADS[wl][0]
        plan = S.RowBlockPlan(n, world, rank)
        eng, inp, adj_dev = bench.build_engine(pkg, torch, dev, wl, spec["seed"], plan=plan)
        st = S.ShardedStepper(S.HipShardBackend(eng, plan), plan, dist=dist, host_staged=True)
        pi, pj = np.asarray(spec["pos_i"]), np.asarray(spec["pos_j"])
        own = (pi >= plan.row_begin) & (pi < plan.row_end)
        ti = torch.as_tensor(pi[own], device=dev); tj = torch.as_tensor(pj[own], device=dev)
        res = {"own": own, "row_range": np.array([plan.row_begin, plan.row_end])}
        for t in range(spec["steps"]):
            sc = st.step(want_scalars=True)
            st.monitor()
            res[f"g{t}"] = eng.buffer("G_sym")[ti, tj].cpu().numpy()
            res[f"scal{t}"] = np.array([sc[k] for k in ("loss", "c1", "c2", "c6", "c7", "c9", "c10", "nll", "clamp_sum")])
            eng.get_rows().cpu().numpy().tofile(f"{out}.rank{rank}.rows{t}.f32")
        res["fused_steps"] = eng.fused_steps()
        res["general_steps"] = eng.path_stats()["general_steps"]
        res["exchanges"] = st.exchanges
        np.savez(f"{out}.rank{rank}.npz", **res)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        with open(f"{out}.rank{rank}.err", "w") as fh:
            fh.write(traceback.format_exc())
        raise
