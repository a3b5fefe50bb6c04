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
    if spec.get("measure"):
        kw["measure"] = spec["measure"]
    return H.synthetic_case(spec["n"], spec.get("nfeat", 11), tuple(spec["widths"]), 4, seed=spec["seed"], **kw)


def join_group(rank, world, rccl):
    """The rank's process group and device: gloo with every rank on cuda:0 (host-staged exchanges: the one-GPU test mode), or --
    rccl, on a box with >= world devices -- backend `nccl` (= RCCL) with rank k on cuda:k, the exchanges on device memory."""
    import torch
    import torch.distributed as dist
    if rccl:
        dev = torch.device(f"cuda:{rank}")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dev = torch.device("cuda:0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(dev)
    return dist, dev


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
        rccl = bool(spec.get("rccl"))
        dist, dev = join_group(rank, world, rccl)
        import mcgra_loader
        pkg = mcgra_loader.load()
        from mc_gra_amd import sharded as S
        from tests import helpers as H
        z = case_of(spec)
        plan = S.RowBlockPlan(spec["n"], world, rank)
        eng = H.engine_from(pkg, z, device=str(dev), plan=plan)
        masked = spec.get("masked_steps", 0)
        if masked:
            w = masked_weights(z)
            eng.set_model(w.W, w.b, w.Wlin, w.blin, w.Ws)
        st = S.ShardedStepper(S.HipShardBackend(eng, plan), plan, dist=dist, host_staged=not rccl)
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


def run_cora_class(name, epochs=None, device="cuda:0", measure=None):
    """PGDAttack.attack on a Cora fixture with the reference-trained weights it carries (tests/test_gpu_parity.py:_run_cora):
    returns (fixture, modified_adj, AUC, model).  Under an initialised process group the class shards the attack itself."""
    import argparse
    import numpy as np
    import torch
    import mcgra_loader
    pkg = mcgra_loader.load()
    from mc_gra_amd import engine as E
    from oracle import mcgra_oracle as O
    from tests import helpers as H
    z = H.load_cora(name)
    if epochs is not None:
        z["epochs"] = np.array(epochs)
    w = O.GCNWeights([z["W0"], z["W1"]], [z["b0"], z["b1"]], z["Wlin"], z["blin"])
    victim, emb = H.FakeGCN(w), H.FakeGCN(w)
    X, adj, lab = z["features"], z["adj"], z["labels"]
    fadj = H.cora_feature_adj(X)
    d = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=device)
    if measure is not None:      # (the fixture's graph, weights and start under another `calc`: no reference AUC to hold it to)
        z["measure"] = np.array(measure)
    Y_A, H_A2 = E.gcn_forward(d(X), d(adj), [d(x) for x in w.W], [d(x) for x in w.b], d(w.Wlin), d(w.blin), emb_nlayer=2)
    model = pkg.PGDAttack(model=victim, embedding=emb, H_A=H_A2, Y_A=Y_A, nnodes=adj.shape[0], loss_type="CE", device=device)
    if H.a0_of(z) is not None:
        model.adj_changes = H.a0_of(z)
    args = argparse.Namespace(max_eval=100, lr=0, dataset="cora", eps=0, measure=str(z["measure"]), useH_A=True, useY_A=True,
                              useY=True, w1=0, w2=0, w6=0, w7=0, w8=0, w9=0, w10=0)
    model.attack(args, None, float(z["lr"]), 0, float(z["weight_sup"]), tuple(z["weight_param"]), fadj, 0, 0, 0, None, None,
                 z["idx_test"], adj, X, np.zeros_like(adj), lab, z["idx_attack"], float(z["num_edges"]), 0,
                 epochs=int(z["epochs"]), label_adj=(lab[:, None] == lab[None, :]).astype(np.float32))
    final = model.modified_adj.cpu().numpy()
    return z, final, O.metric_pool(adj, final, z["idx_attack"]), model


def run_class_rank(rank, world, port, spec, out):
    """One rank of PGDAttack.attack under a process group (gloo: the ranks share the GPU): the class finds the group, shards
    the attack over the ranks (mc-gra_amd/topology_attack.py) and returns the same modified_adj on every rank."""
    try:
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), OMP_NUM_THREADS="2")
        os.environ.pop("MCGRA_KEEP_GSYM", None)
        import numpy as np
        dist, dev = join_group(rank, world, bool(spec.get("rccl")))
        z, final, auc, model = run_cora_class(spec["name"], spec.get("epochs"), device=str(dev), measure=spec.get("measure"))
        sp = z["sample_pos"]
        np.savez(f"{out}.rank{rank}.npz", auc=auc, final_sample=final[sp[:, 0], sp[:, 1]], final_sum=final.astype(np.float64).sum(),
                 acc_test=np.array(model.history.get("acc_test", [])), sharded_world=model.history["path"]["sharded_world"],
                 fused_steps=model.history["path"]["fused_steps"], general_steps=model.history["path"]["general_steps"],
                 collectives=model.history["path"]["collectives"],
                 adj_changes_sum=float(model.adj_changes.double().sum()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        with open(f"{out}.rank{rank}.err", "w") as fh:
            fh.write(traceback.format_exc())
        raise


def main_args(root, extra=()):
    """Command line of the main.py runs below: cora, HSIC with the README's weights, 6 epochs, from a seeded sparse start."""
    return ["--dataset", "cora", "--dataset_root", str(root), "--w1", "0.01", "--w2", "0.01", "--w6", "10", "--w7", "10",
            "--w9", "10", "--w10", "1000", "--lr", "-4.5", "--useH_A", "--useY_A", "--useY", "--measure", "HSIC",
            "--epochs", "6"] + list(extra)


def run_main(argv, n):
    """mc_gra_amd.main.run from a seeded sparse start (adj_changes = U[0,1) / n: the origin is a fixed point of the HSIC loss)."""
    import numpy as np
    import mcgra_loader
    mcgra_loader.load()
    from mc_gra_amd import main as M
    args = M.build_parser().parse_args(argv)
    args.adj_changes_init = (np.random.RandomState(123).rand(n * (n - 1) // 2) / n).astype(np.float32)
    return M.run(args)


def run_main_rank(rank, world, port, argv, n, cwd, out):
    """One rank of `torchrun ... main.py` on a shared GPU (MCGRA_SHARED_GPU=1: gloo, host-staged): main.py joins the group
    itself, before it touches the GPU."""
    try:
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), OMP_NUM_THREADS="2", MCGRA_SHARED_GPU="1")
        os.environ.pop("MCGRA_KEEP_GSYM", None)
        os.chdir(cwd)
        import json
        res = run_main(argv, n)
        with open(f"{out}.rank{rank}.json", "w") as fh:
            json.dump(res, fh)
        import torch.distributed as dist
        if dist.is_initialized():      # (main.run destroys the group it created itself)
            dist.barrier()
            dist.destroy_process_group()
    except Exception:
        with open(f"{out}.rank{rank}.err", "w") as fh:
            fh.write(traceback.format_exc())
        raise


def run_production_line(name, out):
    """One README-line fixture through PGDAttack.attack in the PRODUCTION configuration of the engine: neither MCGRA_AB (the parity
    suite's A/B switches are ignored) nor MCGRA_KEEP_GSYM (no mirrored store of the gradient) nor MCGRA_TESTING in the environment --
    what a user's process looks like.  Runs the class-level checks of tests/test_gpu_readme.py (ensemble sample, sum, AUC within
    north_star's 1e-4 of the reference's, every step of a MSELoss / KL line fused) and writes <out>.ok."""
    try:
        for p in (ROOT, os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        from tests import helpers as H                      # (sets the suite's defaults ...)
        for k in ("MCGRA_AB", "MCGRA_KEEP_GSYM", "MCGRA_TESTING", "MCGRA_SPLIT_BF16"):
            os.environ.pop(k, None)                         # (... which a production process does not have)
        import mcgra_loader
        pkg = mcgra_loader.load()
        from tests import test_gpu_readme as R
        z = H.load_readme(name)
        final = R._class_run(pkg, z, int(z["epochs"]))
        R._final_checks(z, final, name)
        with open(f"{out}.ok", "w") as fh:
            fh.write("ok")
    except BaseException:
        with open(f"{out}.rank0.err", "w") as fh:
            fh.write(traceback.format_exc())
        raise
