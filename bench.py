#!/usr/bin/env python3
"""attack-steps/sec of the MC-GRA adjacency-optimisation hot path on MI355X.

One "step" = one iteration of PGDAttack.attack's loop (topology_attack.py:161-298):
forward over the learnable N x N adjacency, HSIC/entropy/CE losses against the
H_A / Y_A / Y priors, backward into the adjacency, Adam + projection, and the
per-step monitoring forward (:290-296).  Inputs are resident in HBM before the
timed region.  Prints ONE JSON line (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload synthetic-10k-hsic|cora-shape-hsic|...]

N > 1 runs ONE attack, row-block sharded over N ranks, one per GPU (every N x N pass split by
rows, collectives over RCCL inside the timed region); `value` is the steps of that attack per second ("scaling":
"strong").  Under torch.distributed.run (RANK / WORLD_SIZE set) this process is one of the ranks and --gpus must equal
WORLD_SIZE; a plain `python bench.py --gpus N` starts the N ranks itself (launch_ranks) and prints rank 0's line.
Independent replicas, one per GPU, are the side figure `replica_probe`.

At N = 1 two things happen BEFORE this process touches the GPU (child processes): the CPU baseline (whole oracle steps of the
same workload on the host cores, `cpu_baseline`) and two runs of this script under `rocprofv3 --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` that measure the build's memory-side traffic on this box (`live_traffic` -> `roofline.traffic`,
`step_outside_product.bytes_per_step`; without rocprofv3 the committed passes of profiles/ stand in, and the line says which).
Defaults: 200 timed steps behind 20 untimed ones (the clock settles over the first ~30 steps of a run).

Workloads carry their own start and learning rate (WORKLOADS below): adj_changes starts at start_kappa / N x U[0, 1)
-- row sums O(1), the scale of a sparse graph, near the balance of the loss's N x N terms (which shrink the adjacency)
and its small-operand terms (which grow it) -- and lr is a fraction of that scale, so that the timed steps stay in the
regime where every term family carries gradient (scripts/nxn_share.py, profiles/r03_nxn_share_*.json).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 matrix peak

WORKLOADS = {
    # name: (N, nfeat, nclass, hidden, nlayer, measure, weight_param)
    "synthetic-10k-hsic": (10000, 128, 7, 16, 2, "HSIC", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)),
    "synthetic-10k-mse": (10000, 128, 7, 16, 2, "MSELoss", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)),
    # calc = calc_kl (KLDivLoss over rows: 13 of the reference README's 42 command lines) through the fused KL step (round 6)
    "synthetic-10k-kl": (10000, 128, 7, 16, 2, "KL", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)),
    "cora-shape-kl": (2708, 1433, 7, 16, 2, "KL", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)),
    "cora-shape-hsic": (2708, 1433, 7, 16, 2, "HSIC", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)),
    "cora-shape-mse": (2708, 1433, 7, 16, 2, "MSELoss", (0.01, 0, 0, 0, 0, 10, 10, 0, 10, 1000)),
    "synthetic-4k-hsic": (4096, 128, 7, 16, 2, "HSIC", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)),
    # BASELINE.json configs[2]'s shape: Citeseer-sized (N = 3312, 3703 attributes, 6 classes), a GAT victim as the engine sees it
    # (5 heads x 16 = 80 wide, ELU: ENGINE_KW) -- no low-rank forms: the general step with its Gram evaluation on the split kernel
    "citeseer-shape-gat-hsic": (3312, 3703, 6, 80, 2, "HSIC", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)),
    # the headline workload with ~100 relu-masked decode pairs (10 x 10 nodes whose embeddings have disjoint supports: see
    # masked_variant): rounds 1 - 3 sent every such step to the Gram evaluation (3x slower), now it stays fused
    "synthetic-10k-hsic-masked": (10000, 128, 7, 16, 2, "HSIC", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)),
    # BASELINE.json configs[4] shape on ONE GPU (54 GB of N x N buffers; ~0.5 s per step)
    "synthetic-20k-hsic-3layer": (20000, 256, 7, 16, 3, "HSIC", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)),
    "synthetic-30k-hsic-3layer": (30000, 256, 7, 16, 3, "HSIC", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)),
}


ENGINE_KW = {"citeseer-shape-gat-hsic": dict(act="elu", head_act="elu")}      # AttackEngine keywords a workload needs


def make_inputs(n, f, c, hid, nlayer, seed):
    """Synthetic homophilous graph of the workload's shape + random-init victim GCN
    (GraphConvolution.reset_parameters, models/gcn.py:28-33; nn.Linear default)."""
    rng = np.random.RandomState(seed)
    labels = rng.randint(0, c, size=n)
    centers = rng.randn(c, f).astype(np.float32)
    feats = ((centers[labels] * 0.6 + rng.randn(n, f).astype(np.float32) * 0.8) > 0.5).astype(np.float32)
    # sparse symmetric adjacency, ~8 expected neighbours, 3:1 intra-class preference
    deg = 8.0
    p_in, p_out = 3.0 * deg / n / (3.0 / c + (1 - 1.0 / c)), deg / n / (3.0 / c + (1 - 1.0 / c))
    adj = np.zeros((n, n), np.float32)
    blk = 2048
    for i0 in range(0, n, blk):
        same = labels[i0:i0 + blk, None] == labels[None, :]
        u = rng.rand(min(blk, n - i0), n).astype(np.float32)
        adj[i0:i0 + blk] = (u < np.where(same, p_in, p_out))
    adj = np.triu(adj, 1)
    adj = adj + adj.T
    dims = [f] + [hid] * nlayer
    W, b = [], []
    for l in range(nlayer):
        stdv = 1.0 / np.sqrt(dims[l + 1])
        W.append(rng.uniform(-stdv, stdv, size=(dims[l], dims[l + 1])).astype(np.float32))
        b.append(rng.uniform(-stdv, stdv, size=(dims[l + 1],)).astype(np.float32))
    k = 1.0 / np.sqrt(hid)
    Wlin = rng.uniform(-k, k, size=(c, hid)).astype(np.float32)
    blin = rng.uniform(-k, k, size=(c,)).astype(np.float32)
    idx_attack = rng.permutation(n)
    return dict(adj=adj, features=feats, labels=labels, W=W, b=b, Wlin=Wlin, blin=blin, idx_attack=idx_attack, dims=dims)


# Start and learning rate of a workload: adj_changes_0 = (START_KAPPA / n) * U[0, 1) (row sums of modified_adj ~ kappa / 2,
# a sparse graph's scale) and lr = LR_FRACTION * START_KAPPA / n (main.py's --lr is an exponent: any value is a legal
# run).  Why: with the README's weights the small-operand terms c9 / c10 grow like the 3rd power of the row sums and
# the N x N terms c1 / c2 fall with them; from a dense start (round 2: 0.05 * U, row sums 250 at N = 10 000) c1 / c2
# carry 1e-11 of the gradient and are rounded away in its fp32 sum, from an empty one the N x N terms are alone and
# pull every entry to 0 in one Adam step.  kappa ~ 1 is where both families carry the gradient at N = 10 000, and an
# Adam step of lr = kappa / n / 50 (2 % of the start scale per entry) keeps the timed steps there
# (scripts/nxn_share.py -> profiles/r03_nxn_share_10k.json).
START_KAPPA = 1.0
LR_FRACTION = 1.0 / 50.0


def start_scale(workload, n):
    return START_KAPPA / n


def workload_lr(workload, n=None):
    return LR_FRACTION * START_KAPPA / (n or WORKLOADS[workload][0])


def make_a0(n, seed, scale):
    """Seeded non-zero start for adj_changes, generated on the host so that tests/golden/make_golden.py can hand
    the SAME vector to the reference (tests/golden/bench10k_hsic.npz pins this workload against it)."""
    return (np.random.RandomState(seed + 1000).rand(n * (n - 1) // 2) * scale).astype(np.float32)


MASKED_VARIANTS = {"synthetic-10k-hsic-masked": (10, 10)}      # (nodes alive on the first half of em only, on the second half only)


def masked_variant(workload, inp, a0, seed):
    """Weights and start of a `*-masked` workload (tests/test_gpu_parity.py:_few_masked_pairs_case at the bench's scale).  The
    second GCN layer's input is almost rank one, T_1 = 1 u^T + 1e-3 noise, so em_i = relu(s_i u + b), s_i = rowsum_i of the
    learnable adjacency (~ 0.5 at the bench's start): the first half of the coordinates is alive for s_i < 1 (u = -1,
    b = 1), the second half for s_i > 0.3 (u = +1, b = -0.3).  `lo` rows are scaled down (s ~ 0.1) and `hi` rows up (s ~ 1.5):
    their lo x hi cross pairs have embeddings with disjoint supports, S_ij == 0 exactly -- relu-masked pairs, no dead row."""
    if workload not in MASKED_VARIANTS:
        return inp, a0
    n_lo, n_hi = MASKED_VARIANTS[workload]
    n, hid = inp["adj"].shape[0], inp["dims"][-1]
    rng = np.random.RandomState(seed + 77)
    W = [w.copy() for w in inp["W"]]; b = [x.copy() for x in inp["b"]]
    W[0][:, 0] = 0.0; b[0][0] = 1.0                                   # H_0[:, 0] == 1 on every node
    u = np.concatenate([-np.ones(hid // 2), np.ones(hid - hid // 2)]).astype(np.float32)
    W[1] = (rng.randn(*W[1].shape) * 1e-3).astype(np.float32); W[1][0] += u
    b[1] = np.concatenate([1.0 * np.ones(hid // 2), -0.3 * np.ones(hid - hid // 2)]).astype(np.float32)
    f = np.ones(n, np.float32)
    lo = 5 + 7 * np.arange(n_lo); hi = 9 + 11 * np.arange(n_hi)
    f[lo] = 0.2; f[hi] = 3.0
    # scale rows and columns of the packed start: entry (i, j), i > j, at i (i - 1) / 2 + j
    ii = np.repeat(np.arange(1, n), np.arange(1, n)); jj = np.concatenate([np.arange(k) for k in range(1, n)])
    a0 = (a0 * f[ii] * f[jj]).astype(np.float32)
    return dict(inp, W=W, b=b), a0


def feature_adj_cora(feats_dev, torch):
    """main.dot_product_decode for cora (main.py:44-48), on device (setup, untimed)."""
    from mc_gra_amd import engine as E
    Z = E.sgemm(feats_dev, feats_dev, tb=True)
    Z.sub_(torch.eye(Z.shape[0], device=Z.device)).clamp_(min=0)
    return torch.sigmoid(Z)


def gpu_auc(real, pred, torch):
    """Mann-Whitney AUC with average ranks for ties (== sklearn roc_curve + auc)."""
    pred = pred.reshape(-1).double()
    real = real.reshape(-1) > 0
    order = torch.argsort(pred)
    ps = pred[order]
    n = ps.numel()
    newgrp = torch.ones(n, dtype=torch.bool, device=ps.device)
    newgrp[1:] = ps[1:] != ps[:-1]
    gid = torch.cumsum(newgrp.long(), 0) - 1
    starts = torch.nonzero(newgrp).flatten()
    ends = torch.cat([starts[1:], torch.tensor([n], device=ps.device)])
    avg = (starts + ends - 1).double() * 0.5 + 1.0
    ranks = torch.empty(n, dtype=torch.float64, device=ps.device)
    ranks[order] = avg[gid]
    npos = int(real.sum()); nneg = n - npos
    return float((ranks[real].sum() - npos * (npos + 1) / 2.0) / (npos * nneg))


def _oracle_for(workload, seed, ns):
    from oracle import mcgra_oracle as O
    n0, f, c, hid, nl, measure, wp = WORKLOADS[workload]
    inp = make_inputs(ns, f, c, hid, nl, seed)
    X = inp["features"]
    Z = X @ X.T
    fadj = (1.0 / (1.0 + np.exp(-np.maximum(Z - np.eye(ns, dtype=np.float32), 0)))).astype(np.float32)
    w = O.GCNWeights(inp["W"], inp["b"], inp["Wlin"], inp["blin"])
    cfg = O.AttackConfig(measure=measure, weight_sup=1.0, weight_param=wp, lr=workload_lr(workload, ns), num_edges=float("inf"))
    orc = O.PGDAttackOracle(w, X, inp["adj"], np.zeros((ns, ns), np.float32), fadj, inp["labels"], inp["idx_attack"], cfg)
    orc.set_adj_changes(make_a0(ns, seed, start_scale(workload, ns)))
    return orc          # (the `*-masked` variants differ in weights and start only: the same work per step)


def _cpu_child(workload, seed, ns, budget_s, q):
    """Child process (forked before the parent touches the GPU): oracle steps at N = ns, one result per step."""
    try:
        orc = _oracle_for(workload, seed, ns)
        t_all = time.perf_counter()
        while True:
            t0 = time.perf_counter()
            orc.step()
            dt = time.perf_counter() - t0
            q.put(dt)
            if time.perf_counter() - t_all + dt > budget_s:
                break
    except Exception as e:          # e.g. MemoryError on a small host: the parent falls back to the smaller sample
        q.put(f"{type(e).__name__}: {e}"[:200])


def cpu_baseline(workload, seed, timeout_s=210.0):
    """The oracle (numpy restatement of the reference CPU path: Gram evaluation of linear_HSIC, the reference's own
    algorithm) timed on the host cores ON THE WORKLOAD ITSELF: whole steps at the workload's N, in a child process forked
    before this process initialises the GPU, bounded by `timeout_s`.  Only if not even one step finishes inside the bound
    (or the host cannot hold it) does it fall back to a smaller N and say so; nothing is extrapolated."""
    import multiprocessing as mp
    n0, f, c, hid, nl, measure, wp = WORKLOADS[workload]
    ctx = mp.get_context("fork")
    for ns, bound in ((n0, timeout_s), (min(n0, 3072), 60.0)):
        q = ctx.Queue()
        p = ctx.Process(target=_cpu_child, args=(workload, seed, ns, 20.0 if ns == n0 else 10.0, q), daemon=True)
        t0 = time.perf_counter()
        p.start()
        times, err = [], None
        while time.perf_counter() - t0 < bound:
            try:
                x = q.get(timeout=1.0)
            except Exception:
                if not p.is_alive():
                    break
                continue
            if isinstance(x, str):
                err = x
                break
            times.append(x)
        if p.is_alive():
            p.terminate()
        p.join(5)
        if times:
            best = min(times)
            # the four N x N x N products of the Gram evaluation (two centred Grams, two gradient products): 8 N^3 flop
            gflops = 8.0 * ns ** 3 / best / 1e9 if measure in ("HSIC", "CKA") else None
            out = dict(value=1.0 / best, unit="attack-steps/s", cores=os.cpu_count(), kind="port",
                       sample=f"{len(times)} whole oracle step(s) at N={ns} of this workload (same generator, config and start "
                              f"as the GPU run; numpy + BLAS on {os.cpu_count()} host threads), fastest step {best:.2f} s"
                              + ("" if ns == n0 else f"; N={n0} did not finish one step within {timeout_s:.0f} s"
                                 + (f" ({err})" if err else "") + ": smaller sample, NOT scaled"),
                       sample_nodes=ns, step_seconds=times)
            if gflops:
                out["host_gflops_in_products"] = gflops
            return out
    return dict(value=None, unit="attack-steps/s", cores=os.cpu_count(), kind="port", sample=f"oracle did not finish a step ({err})")


_NOT_IN_STEP = ("fillBufferAligned grid=65536", "k_dd2_accum", "k_axpby2d", "rankk_nt_kernel<16>", "k_decode_post", "k_unpack_sym",
                "k_split_absmax", "k_center_cols", "k_rowsum", "copyBuffer")


def live_traffic(workload, seed, steps=6, timeout_s=120.0):
    """Memory-side bytes of THIS build on THIS box, measured inside the bench invocation: two child runs of this script
    under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, `--kernel-trace` only beside them: the form
    MI355X_MICROARCH.md prescribes), started -- like the CPU baseline -- before this process touches the GPU.  Bytes =
    2 x FETCH_SIZE + WRITE_SIZE (the gfx950 correction; counted at the L2's memory side, Infinity-Cache hits included).
    Returns {"product_bytes_per_launch", "outside_product_bytes_per_step", ...} or None (no rocprofv3, a failed pass): the
    bench line then falls back to the committed passes of profiles/ and says so.  The accounting is the one of
    scripts/profile_summary.py (kernels a step launches, by launch count; the one-off launches of set_graph / finalize
    excluded by name)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return {"error": "rocprofv3 not on PATH"}
    # the profiler's preloaded library initialises the GPU before the program behind `--` starts: that program must be the
    # interpreter ITSELF (an ELF binary), not a shim that would exec it (a pyenv / conda wrapper, a `#!/usr/bin/env` script)
    py = os.path.realpath(sys.executable)
    try:
        with open(py, "rb") as fh:
            if fh.read(4) != b"\x7fELF":
                return {"error": f"{py} is not an ELF binary (a launcher that re-execs is refused under the profiler)"}
    except OSError as e:
        return {"error": f"cannot read {py}: {e}"}
    per = {}
    window = {}      # counter -> (bytes between the first and the last Adam pass of the run, steps in between)
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", py, os.path.abspath(__file__),
                   "--steps", str(steps), "--warmup", "1", "--workload", workload, "--seed", str(seed), "--no-cpu-baseline",
                   "--no-split-probe", "--no-live-traffic"]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                   stderr=subprocess.DEVNULL, timeout=timeout_s)
            except Exception as e:
                return {"error": f"{ctr} pass: {type(e).__name__}: {e}"[:200]}
            fs = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not fs:
                return {"error": f"{ctr} pass: exit code {r.returncode}, {len(fs)} counter file(s)"}
            with open(fs[0]) as fh:
                rows_ = [row for row in csv.DictReader(fh) if row["Counter_Name"] == ctr]
            # every launch between the first and the last Adam pass of the run, in dispatch order: whole iterations (step + monitoring
            # forward) and nothing else -- no allocation fills of engine creation, no finalize
            rows_.sort(key=lambda r_: int(r_["Dispatch_Id"]))
            adam = [i for i, r_ in enumerate(rows_) if any(x in r_["Kernel_Name"] for x in ("k_tail_adam", "k_adam_sym", "k_rankk_apply_adam"))]
            if len(adam) >= 2:
                window[ctr] = (sum(float(r_["Counter_Value"]) for r_ in rows_[adam[0] + 1:adam[-1] + 1]), len(adam) - 1)
            if True:
                for row in rows_:
                    k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0] + f" grid={row['Grid_Size']}"
                    e = per.setdefault(k, {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
                    e[ctr] += float(row["Counter_Value"])
                    if ctr == "FETCH_SIZE":
                        e["n"] += 1
    nsteps = steps + 1
    prod_main, outside = None, 0.0
    for k, v in per.items():
        n = max(v["n"], 1)
        b = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) / n * 1024.0
        if "split2_m16_kernel" in k or "split3_symm_kernel" in k or "k_split3_reduce" in k:
            # ONE product per step, in several launches: the parts of a cut launch (whole rounds of the chip, DESIGN.md section
            # 1c), the split-K launch of the ragged last round and the sum of its slabs -- all of them, per step
            prod_main = (prod_main or 0.0) + b * n / nsteps
            continue
        if ("mcgra::" in k or "rocclr" in k) and n >= nsteps and "gemm_f32_kernel<128, 128" not in k \
                and not any(x in k for x in _NOT_IN_STEP):
            outside += b * (n // nsteps)
    if prod_main is None:
        if WORKLOADS[workload][5] == "HSIC":
            return {"error": "no launch of the product kernel in the counter pass"}
        prod_main = 0.0      # (the fused MSELoss / KL steps have no N x N x N product: everything is "outside")
    step_bytes = None
    if "FETCH_SIZE" in window and "WRITE_SIZE" in window:
        step_bytes = (2.0 * window["FETCH_SIZE"][0] / window["FETCH_SIZE"][1] + window["WRITE_SIZE"][0] / window["WRITE_SIZE"][1]) * 1024.0
    return {"product_bytes_per_launch": prod_main, "outside_product_bytes_per_step": outside, "steps_in_pass": nsteps,
            "iteration_bytes": step_bytes,      # whole iterations between the run's first and last Adam pass, per iteration
            "iteration_how": "2 x FETCH_SIZE + WRITE_SIZE over every launch between the first and the last Adam pass of a "
                             f"{nsteps}-step run under rocprofv3 --pmc (two child passes of this invocation), per iteration "
                             "(128-byte fabric requests confirmed for the tile kernels: profiles/r06_tcc_requests_mse_tail.txt)",
            "how": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace over {nsteps} steps of this workload, run by this invocation "
                   "before its timed region; bytes = 2 x FETCH_SIZE + WRITE_SIZE"}


def timed_region(step_fn, steps, warmup, sync, world, dist=None, device=None, torch=None):
    """The driver's timing contract: W untimed steps, then exactly K steps bracketed by barrier + device sync on
    both sides; returns the MAX over ranks of the elapsed seconds.  `sync()` is torch.cuda.synchronize on GPU
    (a no-op in the gloo CPU test that covers the N > 1 plumbing)."""
    for _ in range(warmup):
        step_fn()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def aggregate_value(world, steps, dt):
    """Whole-job throughput of N independent replicas (weak scaling): every rank ran `steps` attack steps."""
    return world * steps / dt


def build_engine(pkg, torch, dev, workload, seed, weight_param=None, **kw):
    """Engine + inputs of one workload, seeded: the same seed gives bit-identical state on every rank."""
    n, f, c, hid, nl, measure, wp = WORKLOADS[workload]
    wp = wp if weight_param is None else weight_param
    inp = make_inputs(n, f, c, hid, nl, seed)
    a0 = make_a0(n, seed, start_scale(workload, n))
    inp, a0 = masked_variant(workload, inp, a0, seed)
    X = torch.as_tensor(inp["features"], device=dev)
    fadj = feature_adj_cora(X, torch)
    eng = pkg.AttackEngine(n, inp["dims"], c, 2, measure, 1.0, wp, workload_lr(workload, n), 1e30, n, device=dev,
                           **dict(ENGINE_KW.get(workload, {}), **kw))
    eng.set_model(inp["W"], inp["b"], inp["Wlin"], inp["blin"])
    adj_dev = torch.as_tensor(inp["adj"], device=dev)
    eng.set_graph(X, adj_dev, None, fadj, inp["labels"], inp["idx_attack"])
    # seeded non-zero start: with measure=HSIC the origin is a fixed point of the exact dynamics
    # (DESIGN.md section 5, fact 2), so a zero start would time a run that optimises nothing
    eng.set_adj_changes(torch.as_tensor(a0, device=dev))
    return eng, inp, adj_dev


def replica_probe(pkg, torch, dist, dev, rank, world, workload, seed, steps, warmup, monitor, red_dev=None):
    """N independent attacks, one per rank, each on its own graph and with no data-path collective (e.g. the trials of
    main.py's search loop): aggregate throughput of the job used that way.  A side figure at N > 1, never `value`."""
    eng, _, _ = build_engine(pkg, torch, dev, workload, seed + 1 + rank)

    def one_step():
        eng.step()
        if monitor:
            eng.monitor()

    dt = timed_region(one_step, steps, warmup, torch.cuda.synchronize, world, dist, red_dev or dev, torch)
    del eng
    torch.cuda.empty_cache()
    return {"steps_per_s_all_replicas": aggregate_value(world, steps, dt), "ms_per_step_per_replica": 1e3 * dt / steps,
            "replicas": world, "steps": steps, "what": "independent attacks, one per GPU, no collective (weak scaling)"}


def ab_switch(name):
    """An A/B switch of the engine as the engine reads it: honoured only beside MCGRA_AB=1 (attack.hip: ab_env)."""
    return os.environ.get(name) if os.environ.get("MCGRA_AB") == "1" else None


PRODUCT_MODES = {0: "fp32 MFMA SYMM (gemm_f32_kernel, SYM_MM)",
                 1: "SINGLE-plane fp16 product x0 y0 of the 2-plane operands (low planes compiled out of split2_m16_kernel): fp16 accuracy, "
                    "2^-11 per operand -- a named mode (MCGRA_SPLIT_BF16=1), never a default",
                 2: "3-plane bf16 split, hand-written split3_symm_kernel on packed planes (256 x 256 tiles)",
                 3: "2-plane fp16 split (3 plane products, exact power-of-two operand scales), hand-written "
                    "split2_m16_kernel on packed planes (256 x 256 tiles)"}
PLANE_PRODUCTS = {0: 1, 1: 1, 2: 6, 3: 3}


def product_probe(pkg, torch, dev, workload, seed, steps, warmup, monitor, mode, overlap=None, extra_env=None):
    """The same workload with another evaluation of the one N x N x N product (MCGRA_SPLIT_BF16=mode): the pure fp32
    MFMA path beside a split headline, or the other way round; overlap=0: the product alone on the chip instead of
    beside the step's HBM-bound kernels.  Never the reported `value`."""
    env = {"MCGRA_SPLIT_BF16": str(mode), "MCGRA_AB": "1"}      # (A/B switches are honoured only beside MCGRA_AB=1)
    if overlap is not None:
        env["MCGRA_OVERLAP"] = str(overlap)
    env.update(extra_env or {})
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        eng, inp, adj_dev = build_engine(pkg, torch, dev, workload, seed)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v

    def one_step():
        eng.step()
        if monitor:
            eng.monitor()

    for _ in range(warmup):
        one_step()
    eng.profile(True); eng.gemm_stats(reset=True)
    dt = timed_region(one_step, steps, 0, torch.cuda.synchronize, 1, None, dev, torch)
    st = eng.gemm_stats(reset=True)
    eng.profile(False)
    lab = torch.as_tensor(inp["labels"], device=dev)
    final = eng.finalize(0, eng.buffer("HA"), eng.buffer("YA"), (lab[:, None] == lab[None, :]).float())
    n = WORKLOADS[workload][0]
    pm = eng.product_mode()
    out = {"value": steps / dt, "unit": "attack-steps/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
           "auc": gpu_auc(adj_dev, final, torch), "product": PRODUCT_MODES[pm]}
    if st["launches"]:
        ms = st["ms"] / st["launches"]
        out["product_avg_launch_ms"] = ms
        out["product_fp32_equivalent_tflops"] = 2.0 * n ** 3 / (ms * 1e-3) / 1e12
        if pm:
            out["product_16bit_tflops_issued"] = 2.0 * PLANE_PRODUCTS[pm] * n ** 3 / (ms * 1e-3) / 1e12
    return out


def multi_rank_diagnostics(pkg, torch, dist, dev, rank, world, a, eng, stepper, plan, comm, st, auc, dt, monitor, shared_gpu):
    """Fields of the N > 1 line beside `value` (never part of it; everything here runs behind the timed region):
      comm_ms_per_step / allgather_ms / alltoall_ms   HIP events around every collective of the timed steps, on the stream it is
                                issued on: what the compute stream loses to it, per step -- mean and max over the ranks
      product_ms_per_rank       every rank's N x N x N product per step (its own HIP events)
      compute_only_ms_per_step  the same steps with every collective answered by the rank's own data (sharded.run_echo): the
                                launches of a real step, nothing exchanged -- max over the ranks
      exposed_comm_ms_per_step  ms_per_step - compute_only_ms_per_step: what the exchange costs the step after overlap
      state_check               recovered-adjacency AUC of the sharded attack against ONE rank running the same attack
                                (same workload, start, warm-up + timed steps) -- rank 0 replays it when its HBM holds a second
                                engine; |difference| <= 1e-4 is `ok`"""
    from mc_gra_amd import sharded as S
    n = WORKLOADS[a.workload][0]
    steps = a.steps
    prod_ms = st["ms"] / steps if st["launches"] else None
    # (a rank that fails here still answers the collectives below: the line must not hang on one rank's diagnostics)
    try:
        # compute only: from the workload's own start again (the echo leaves a state that is not the attack's)
        a0 = make_a0(n, a.seed, start_scale(a.workload, n))
        if a.workload in MASKED_VARIANTS:
            a0 = masked_variant(a.workload, make_inputs(*WORKLOADS[a.workload][:5], a.seed), a0, a.seed)[1]
        eng.set_adj_changes(torch.as_tensor(a0, device=dev))
        del a0
        k = max(2, min(steps, 20))
        f0 = eng.fused_steps()
        for i in range(2):
            S.run_echo(stepper.b, S.SHARD_STEP)
            if monitor:      # (as in the timed region: k products inside the clock, none forked in front of it or left behind it)
                S.run_echo(stepper.b, S.SHARD_MONITOR_LAST if i == 1 else S.SHARD_MONITOR)
        torch.cuda.synchronize()                 # (rank-local: the echo exchanges nothing, no barrier inside this try)
        t0 = time.perf_counter()
        for i in range(k):
            S.run_echo(stepper.b, S.SHARD_STEP)
            if monitor:
                S.run_echo(stepper.b, S.SHARD_MONITOR_LAST if i == k - 1 else S.SHARD_MONITOR)
        torch.cuda.synchronize()
        echo_ms = 1e3 * (time.perf_counter() - t0) / k
        echo_fused = eng.fused_steps() - f0
        mine = {"rank": rank, "product_ms": prod_ms, "echo_ms": echo_ms, "echo_fused": echo_fused == k + 2,
                "comm": {kk: (v / steps if kk != "count" else v / steps) for kk, v in (comm or {}).items()}}
    except Exception as e:
        mine = {"rank": rank, "product_ms": prod_ms, "echo_ms": None, "echo_fused": False, "comm": {},
                "error": f"{type(e).__name__}: {e}"[:300]}
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    out = None
    if rank == 0:
        cm = lambda key: [r["comm"].get(key, 0.0) for r in allr]
        tot = [r["comm"].get("allgather", 0.0) + r["comm"].get("alltoall", 0.0) + r["comm"].get("allreduce", 0.0) for r in allr]
        echo = [r["echo_ms"] for r in allr]
        compute_only = max(echo) if all(x is not None for x in echo) else None
        out = {"comm_ms_per_step": {"mean": sum(tot) / world, "max": max(tot)},
               "allgather_ms_per_step": {"mean": sum(cm("allgather")) / world, "max": max(cm("allgather"))},
               "alltoall_ms_per_step": {"mean": sum(cm("alltoall")) / world, "max": max(cm("alltoall"))},
               "collectives_timed_per_step": allr[0]["comm"].get("count"),
               "product_ms_per_rank": [r["product_ms"] for r in allr],
               "compute_only_ms_per_step": compute_only,
               "compute_only_steps_all_fused": all(r["echo_fused"] for r in allr),
               "exposed_comm_ms_per_step": (1e3 * dt / steps - compute_only) if compute_only is not None else None,
               "errors": [r["error"] for r in allr if r.get("error")] or None,
               "how": "comm: HIP events around each collective on the issuing stream, timed steps; compute only: the same steps with "
                      "every collective answered by the rank's own data (sharded.run_echo), max over ranks; exposed = ms_per_step - "
                      "compute only" + (" [ranks share ONE GPU over gloo: plumbing test, not a measurement]" if shared_gpu else "")}
    # the same attack on ONE rank (rank 0, when its HBM holds a second engine): the state check
    check = None
    if rank == 0:
        try:
            free, total = torch.cuda.mem_get_info(dev)
            need = 16.5 * 4.0 * n * ((n + 31) // 32 * 32)            # N x N buffers of a monolithic HSIC engine + planes
            if free > 1.15 * need:
                e1, inp1, adj1 = build_engine(pkg, torch, dev, a.workload, a.seed)
                for _ in range(a.warmup + steps):
                    e1.step()
                    if monitor:
                        e1.monitor()
                lab = torch.as_tensor(inp1["labels"], device=dev)
                fin1 = e1.finalize(0, e1.buffer("HA"), e1.buffer("YA"), (lab[:, None] == lab[None, :]).float())
                auc1 = gpu_auc(adj1, fin1, torch)
                check = {"auc_sharded": auc, "auc_one_rank": auc1, "abs_diff": abs(auc - auc1), "ok": abs(auc - auc1) <= 1e-4,
                         "steps": a.warmup + steps, "how": "rank 0 replayed the attack monolithically behind the timed region"}
                del e1, fin1
                torch.cuda.empty_cache()
            else:
                check = {"auc_sharded": auc, "auc_one_rank": None, "ok": None,
                         "how": f"not replayed: {free / 2**30:.0f} GiB free on rank 0, a second engine needs ~{need / 2**30:.0f} GiB"}
        except Exception as e:
            check = {"auc_sharded": auc, "auc_one_rank": None, "ok": None, "how": f"replay failed: {type(e).__name__}: {e}"[:300]}
    dist.barrier()
    if out is not None:
        out["state_check"] = check
    return out


def launch_ranks(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, one per GPU, exactly as the driver's
    own multi-GPU command does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py ...`) from this still-GPU-free process, forward rank 0's JSON line as the ONLY stdout line
    and fail if any rank fails.  (MCGRA_BENCH_SHARED_GPU=1 in the environment puts every rank on cuda:0 over gloo: the
    1-GPU test mode.)"""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_ranks)))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for attempt in range(3):
        # (a free port found by bind(0) can be taken before the launcher binds it -- two bench invocations on one box: the
        # launch is tried again on another one when the rendezvous reports the address in use)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        err = r.stderr or ""
        sys.stderr.write(err)                                                         # the ranks' stderr passes through
        if r.returncode == 0 or not ("EADDRINUSE" in err or "address already in use" in err.lower()):
            break
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            try:
                cand = json.loads(ln)
            except ValueError:
                continue
            if isinstance(cand, dict) and "metric" in cand:
                line = cand
    if r.returncode != 0 or line is None:
        sys.stderr.write(r.stdout)
        raise SystemExit(f"bench.py: the {n_ranks}-rank launch failed (exit code {r.returncode}"
                         + ("" if line is not None else ", no result line from rank 0") + ")")
    if line["n_gpus"] != n_ranks:
        raise SystemExit(f"bench.py: asked for {n_ranks} ranks, the line says {line['n_gpus']}")
    print(json.dumps(line), flush=True)
    return line


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one per GPU) of ONE row-block sharded attack; default: WORLD_SIZE under a launcher, else 1.  "
                         "Without a launcher and N > 1 this process starts the N ranks itself (torch.distributed.run)")
    # (defaults: 200 timed steps behind 20 untimed ones, 1.3 s in all.  The chip's clock settles over the first ~30 steps of
    # a run -- the product takes 4.80 ms per launch over steps 4 - 23 and 4.66 over steps 21 - 220 on the same box -- so a
    # 20-step window behind 3 warm-up steps, rounds 1 - 3's default, times the transient: 172 - 173 against 175 - 178 steps/s)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="synthetic-10k-hsic", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-monitor", action="store_true", help="skip the per-step monitoring forward (:290-296)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-shard-probe", action="store_true",
                    help="N > 1: skip the extra (untimed-for-value) run of independent replicas, one attack per rank")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc child passes that measure the memory-side traffic of this build on this box "
                         "(N = 1; implied by --no-split-probe)")
    ap.add_argument("--live-traffic", action="store_true",
                    help="run the two --pmc child passes even with --no-split-probe (the fused MSELoss / KL workloads: roofline.traffic per step)")
    ap.add_argument("--no-split-probe", action="store_true",
                    help="skip the extra runs at N = 1 (other evaluation of the N x N x N product, Cora-shape workload)")
    a = ap.parse_args(argv)

    if "WORLD_SIZE" not in os.environ:
        if (a.gpus or 1) > 1:
            # plain `python bench.py --gpus N`: this process (which has not touched the GPU) starts the N ranks itself
            return launch_ranks(a.gpus, list(sys.argv[1:] if argv is None else argv))
    elif a.gpus is not None and a.gpus != int(os.environ["WORLD_SIZE"]):
        raise SystemExit(f"bench.py: --gpus {a.gpus} under a launcher that set WORLD_SIZE={os.environ['WORLD_SIZE']}: "
                         "the two must agree (one rank per GPU)")
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # MCGRA_BENCH_SHARED_GPU=1 (tests on a 1-GPU box): every rank on cuda:0, the collectives over gloo with host-staged
    # arena slices (RCCL refuses two ranks on one device); same engine, protocol and timing contract, never a reported number
    shared_gpu = world > 1 and "1" in (os.environ.get("MCGRA_BENCH_SHARED_GPU"), os.environ.get("MCGRA_SHARED_GPU"))
    # CPU baseline first (N = 1 only): its child process is forked before anything here touches the GPU, and it is over
    # before the timed region starts, so the host cores are idle while the GPU is timed
    cpu = cpu_baseline(a.workload, a.seed) if (world == 1 and not a.no_cpu_baseline) else None
    # ... and so are the two PMC passes that measure this build's memory-side traffic on this box (children under rocprofv3)
    live, live_error = None, None
    if world == 1 and not a.no_live_traffic and (not a.no_split_probe or a.live_traffic) and WORKLOADS[a.workload][5] in ("HSIC", "MSELoss", "KL"):
        try:
            live = live_traffic(a.workload, a.seed)
        except Exception as e:
            live = {"error": f"{type(e).__name__}: {e}"[:200]}
        if live is not None and "error" in live:      # the line says why the committed passes of profiles/ stand in
            live, live_error = None, live["error"]
    import torch
    import torch.distributed as dist
    if shared_gpu:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    red_dev = torch.device("cpu") if shared_gpu else dev          # where timed_region's MAX-over-ranks tensor lives
    import mcgra_loader
    pkg = mcgra_loader.load()

    n, f, c, hid, nl, measure, wp = WORKLOADS[a.workload]
    monitor = not a.no_monitor
    stepper = plan = None
    if world > 1:
        # ONE attack, row-block sharded over the ranks: every rank builds the same graph and owns
        # the rows of the learnable adjacency its plan names; collectives over RCCL inside the timed region
        from mc_gra_amd.sharded import RowBlockPlan, ShardedStepper, HipShardBackend
        plan = RowBlockPlan(n, world, rank)
        eng, inp, adj_dev = build_engine(pkg, torch, dev, a.workload, a.seed, plan=plan)
        stepper = ShardedStepper(HipShardBackend(eng, plan), plan, dist=dist, host_staged=shared_gpu)

        def one_step(last=False):
            stepper.step()
            if monitor:
                stepper.monitor(last=last)
    else:
        eng, inp, adj_dev = build_engine(pkg, torch, dev, a.workload, a.seed)

        def one_step(last=False):
            eng.step()
            if monitor:
                eng.monitor()

    # (a row-block rank's monitoring forward forks the NEXT step's product: the last warm-up step and the last timed step say
    # `last`, so the timed region holds exactly K steps' worth of products -- none started in front of it, none left behind it)
    for i in range(a.warmup):
        one_step(last=(i == a.warmup - 1))
    eng.profile(True); eng.gemm_stats(reset=True)
    ex0 = stepper.exchanges if stepper is not None else 0
    if stepper is not None:
        stepper.time_exchanges(True)          # HIP events around every collective of the timed steps (comm_ms_per_step below)
    k_done = [0]

    def timed_step():
        k_done[0] += 1
        one_step(last=(k_done[0] == a.steps))

    dt = timed_region(timed_step, a.steps, 0, torch.cuda.synchronize, world, dist, red_dev, torch)
    comm = stepper.comm_ms() if stepper is not None else None
    if stepper is not None:
        stepper.time_exchanges(False)
    st = eng.gemm_stats(reset=True)
    eng.profile(False)
    eng_path = dict(eng.path_stats(), fused_steps=eng.fused_steps(), gram_split_steps=eng.gram_split_steps(),
                    masked_fused_steps=eng.masked_fused_steps(), cut_product_steps=eng.cut_product_steps())
    stepper_exchanges = ((stepper.exchanges - ex0) / a.steps) if stepper is not None else None      # of the timed steps

    # the same N x N x N product alone on the chip (no side-stream company), for the roofline's "alone" figure
    alone_ms = None
    if world == 1 and st["launches"] and eng_path["lowrank_steps"] > 0 and measure == "HSIC" and eng.product_mode() == 0:
        from mc_gra_amd import engine as E
        KFC, Bop = eng.buffer("KFC"), eng.buffer("adj_norm")
        out_s = torch.empty_like(Bop)
        E.ssymm_lower(KFC, Bop, out=out_s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            E.ssymm_lower(KFC, Bop, out=out_s)
        e1.record()
        torch.cuda.synchronize()
        alone_ms = e0.elapsed_time(e1) / 3
        del out_s

    # recovered-adjacency AUC of the run (post-loop ensemble, topology_attack.py:300-324), untimed
    lab = torch.as_tensor(inp["labels"], device=dev)
    label_adj = (lab[:, None] == lab[None, :]).float()
    H_A = eng.buffer("HA"); Y_A = eng.buffer("YA")
    final = eng.finalize(0, H_A, Y_A, label_adj)
    auc = gpu_auc(adj_dev, final, torch)

    # N > 1: what makes the first multi-GPU line explain itself -- the collectives' time by kind, every rank's product time, the
    # same steps with nothing exchanged (compute only), and the sharded attack's result against the 1-rank attack's
    multi = None
    if world > 1:
        multi = multi_rank_diagnostics(pkg, torch, dist, dev, rank, world, a, eng, stepper, plan, comm, st, auc, dt, monitor, shared_gpu)

    # the other evaluation of the N x N x N product beside the headline: pure fp32 MFMA when the default (bf16 split,
    # fp32-level error) ran, the split when MCGRA_SPLIT_BF16=0 was given
    pmode = eng.product_mode()
    split = None
    if world == 1 and measure == "HSIC" and not a.no_split_probe:
        try:
            split = product_probe(pkg, torch, dev, a.workload, a.seed + rank, min(a.steps, 20), min(a.warmup, 3), monitor, 0 if pmode else 2)
            split["headline_auc"] = auc
            split["headline_over_this"] = aggregate_value(world, a.steps, dt) / split["value"]
        except Exception as e:
            split = {"error": f"{type(e).__name__}: {e}"[:300]}

    # the split product by itself: the timed engine's own last product replayed back to back with nothing beside it
    replay_ms = None
    # (not under --no-split-probe: the rocprofv3 passes of profiles/ then hold the step's own launches only)
    if world == 1 and measure == "HSIC" and pmode in (2, 3) and eng_path["fused_steps"] > 0 and wp[0] != 0 and not a.no_split_probe:
        try:
            eng.product_replay(3)
            replay_ms = eng.product_replay(20)
        except Exception as e:
            replay_ms = None
    # ... and the step without the side stream (same engine configuration with MCGRA_OVERLAP=0): what the fork is worth
    alone = None
    side_stream = (ab_switch("MCGRA_OVERLAP") or ("1" if pmode == 3 else "0")) == "1"
    if world == 1 and measure == "HSIC" and pmode in (2, 3) and side_stream and not a.no_split_probe:
        try:
            alone = product_probe(pkg, torch, dev, a.workload, a.seed + rank, min(a.steps, 40), min(a.warmup, 10), monitor, pmode, overlap=0)
        except Exception as e:
            alone = {"error": f"{type(e).__name__}: {e}"[:300]}

    # the same workload when the fused low-rank step does not apply (a dead embedding row, a GAT / GraphSAGE victim, CKA, a
    # hidden width > 32): the reference's own formulation, four N x N x N products per step on the same split kernel
    gram = None
    if world == 1 and measure == "HSIC" and pmode in (2, 3) and not a.no_split_probe:
        try:
            g = product_probe(pkg, torch, dev, a.workload, a.seed, min(a.steps, 5), 1, monitor, pmode, extra_env={"MCGRA_NO_LOWRANK": "1"})
            gram = {"value": g["value"], "unit": "attack-steps/s", "ms_per_step": g["ms_per_step"], "steps": g["steps"], "auc": g["auc"],
                    "headline_over_this": (a.steps / dt) / g["value"],
                    "what": "Gram evaluation of linear_HSIC (MCGRA_NO_LOWRANK=1): what a step costs when the fused low-rank step "
                            "does not apply"}
        except Exception as e:
            gram = {"error": f"{type(e).__name__}: {e}"[:300]}

    # BASELINE.json configs[1] (Cora-sized dense GCN + HSIC, fp32) beside the 10k headline: same engine, same timing
    extra = None
    if world == 1 and a.workload == "synthetic-10k-hsic" and not a.no_split_probe:
        try:
            e2, _, _ = build_engine(pkg, torch, dev, "cora-shape-hsic", a.seed)

            def cora_step():
                e2.step()
                if monitor:
                    e2.monitor()

            dt2 = timed_region(cora_step, 100, 10, torch.cuda.synchronize, 1, None, dev, torch)
            extra = {"cora-shape-hsic": {"value": 100 / dt2, "unit": "attack-steps/s", "ms_per_step": 10.0 * dt2,
                                         "nodes": WORKLOADS["cora-shape-hsic"][0], "steps": 100}}
            del e2
            # ... and calc = MSELoss (the measure of the reference's README headline run, configs[0]) through the fused MSELoss
            # step: Cora shape and the headline's N
            # ... and configs[2]'s shape (Citeseer-sized, GAT victim): the general step, its four Gram products on the split kernel
            # ... and calc = calc_kl (the README's most common measure) through the fused KL step
            for wl2, k2 in (("cora-shape-mse", 100), ("synthetic-10k-mse", 60), ("citeseer-shape-gat-hsic", 60), ("cora-shape-kl", 100),
                            ("synthetic-10k-kl", 60)):
                torch.cuda.empty_cache()
                e3, _, _ = build_engine(pkg, torch, dev, wl2, a.seed)

                def mse_step():
                    e3.step()
                    if monitor:
                        e3.monitor()

                dt3 = timed_region(mse_step, k2, 10, torch.cuda.synchronize, 1, None, dev, torch)
                extra[wl2] = {"value": k2 / dt3, "unit": "attack-steps/s", "ms_per_step": 1e3 * dt3 / k2, "nodes": WORKLOADS[wl2][0], "steps": k2,
                              "fused_steps": e3.fused_steps(), "general_steps": e3.path_stats()["general_steps"],
                              "gram_split_steps": e3.gram_split_steps()}
                del e3
            # ... and the precision BASELINE.json's configs[2] / [4] name ("bf16 MFMA"), taken literally: the headline workload and the
            # Cora shape with the N x N x N product as ONE fp16 plane product (MCGRA_SPLIT_BF16=1).  A named, non-headline mode: the
            # reference's CPU path is fp32 and `value` stays on the fp32-level split; its accuracy against float64 and its AUC delta
            # are measured by scripts/single_plane_table.py (profiles/r06_single_plane_table.txt)
            for wl2, k2 in (("synthetic-10k-hsic", 60), ("cora-shape-hsic", 100), ("citeseer-shape-gat-hsic", 60)):
                torch.cuda.empty_cache()
                sp1 = product_probe(pkg, torch, dev, wl2, a.seed, k2, 10, monitor, 1)
                extra[wl2 + "-f16-single-plane"] = {
                    "value": sp1["value"], "unit": "attack-steps/s", "ms_per_step": sp1["ms_per_step"], "nodes": WORKLOADS[wl2][0], "steps": k2,
                    "dtype": "f16 single plane (the N x N x N products only -- one per fused step, four per Gram-evaluation step: x0 y0 of the "
                             "power-of-two-scaled operands, fp32 accumulate; everything else as the headline)", "product_avg_launch_ms": sp1.get("product_avg_launch_ms"),
                    "product_16bit_tflops": sp1.get("product_16bit_tflops_issued"), "auc": sp1["auc"],
                    "auc_minus_headline": (sp1["auc"] - auc) if wl2 == a.workload and k2 == a.steps else None,
                    "note": "non-headline: never `value`; accuracy table in profiles/r06_single_plane_table.txt"}
        except Exception as e:
            extra = dict(extra or {}, error=f"{type(e).__name__}: {e}"[:300])

    replicas = None
    if world > 1 and not a.no_shard_probe:
        del eng, final, H_A, Y_A, label_adj, stepper
        torch.cuda.empty_cache()
        try:
            replicas = replica_probe(pkg, torch, dist, dev, rank, world, a.workload, a.seed, min(a.steps, 10), 2, monitor, red_dev)
        except Exception as e:                       # never lose the headline line to the probe
            replicas = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        out = {
            # N > 1: ONE attack sharded over the ranks -- steps of that attack per second, not a sum over replicas
            "metric": "attack-steps/sec", "value": a.steps / dt, "unit": "attack-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            # one attack whatever N: its N x N passes are split over the ranks, total work fixed
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if not pmode else ("f32 (the N x N x N product: operands as 2 fp16 planes, 3 MFMA products, fp32 accumulate)"
                                              if pmode == 3 else
                                              "f32 (the N x N x N product: operands as 3 bf16 planes, 6 MFMA products, fp32 accumulate)"),
            "data": "synthetic",
            "config": {"workload": a.workload, "nodes": n, "features": f, "gcn_layers": nl, "hidden": hid,
                       "classes": c, "measure": measure, "priors": "H_A+Y_A+Y", "weight_param": list(wp),
                       "start": f"adj_changes_0 = {start_scale(a.workload, n):.3g} * U[0,1) (kappa / N, kappa = {START_KAPPA:g}), "
                                f"lr = {workload_lr(a.workload, n):.3g}",
                       "lr": workload_lr(a.workload, n), "start_scale": start_scale(a.workload, n),
                       # which step implementation ran, over warmup + timed steps (the fused low-rank step needs a decode
                       # without relu-masked pairs; a masked step is redone by the Gram evaluation, 2.8x slower)
                       "fused_steps": eng_path["fused_steps"], "lowrank_steps": eng_path["lowrank_steps"],
                       "general_steps": eng_path["general_steps"], "gram_split_steps": eng_path["gram_split_steps"],
                       # fused steps whose decode relu-masked pairs of live embedding rows (they stand; only a dead row falls back)
                       "masked_fused_steps": eng_path["masked_fused_steps"],
                       "monitor_forward": monitor,
                       "forward_reuse": bool(monitor and ab_switch("MCGRA_NO_FWD_REUSE") != "1"),
                       "parallelism": (f"row-block x{world}: one attack, rows of the learnable adjacency and of every N x N pass "
                                       f"split over the ranks ({plan.rows_per_rank} rows each); per step one all-to-all of "
                                       "P1 tile blocks and all-gathers of n x c node arrays with the partial scalars in their lane "
                                       + ("(gloo, host-staged: the ranks share ONE GPU -- a test mode, not a measurement)"
                                          if shared_gpu else "(RCCL)")
                                       if world > 1 else "single")},
            "auc": auc,
        }
        if replicas is not None:
            out["replica_probe"] = replicas
        if multi is not None:
            out["multi_rank"] = multi
        if stepper_exchanges is not None:
            out["collectives_per_step"] = stepper_exchanges
            # steps (warm-up + timed) whose product ran the peers' row panels first and handed them to the all-to-all while the
            # own panels were still running (rank 0's count; default for 2 <= N <= 4)
            out["config"]["all_to_all_beside_product_steps"] = eng_path["cut_product_steps"]
        if split is not None:
            out["fp32_mfma_probe" if pmode else "split_bf16_probe"] = split
        if extra is not None:
            out["other_workloads"] = extra
        if gram is not None:
            out["gram_path_probe"] = gram
        if st["launches"]:
            avg_ms = st["ms"] / st["launches"]
            ach = st["flops"] / st["launches"] / (avg_ms * 1e-3) / 1e12
            # HBM-side bytes per launch come from separate rocprofv3 --pmc passes of this same command
            # (profiles/r04_gemm_traffic.json, FETCH_SIZE doubled per the gfx950 correction); PMC counters
            # cannot be read from inside the timed process, so the committed profile value is reported.
            traffic = None
            ps = eng_path
            lowrank = ps["lowrank_steps"] > 0 and ps["general_steps"] == 0
            tp = os.path.join(ROOT, "profiles", "r04_gemm_traffic.json")
            role = {2: "split", 3: "split_f16"}.get(pmode, "symm") if lowrank else "symm"
            traffic_source = None
            if a.workload == "synthetic-10k-hsic" and os.path.exists(tp):
                ks = [k for k in json.load(open(tp))["kernels"] if k.get("role") == role or not lowrank]
                traffic = sum(k["hbm_bytes_corrected"] * k["launches"] for k in ks) / max(1, sum(k["launches"] for k in ks)) if ks else None
                traffic_source = "committed PMC passes (profiles/r04_gemm_traffic.json)"
            if live is not None and lowrank and pmode:
                traffic, traffic_source = live["product_bytes_per_launch"], live["how"]
            if lowrank and pmode:
                # the 16-bit matrix cores issue 6 (bf16 x 3) or 3 (fp16 x 2) plane products per fp32-equivalent product
                npp = PLANE_PRODUCTS[pmode]
                issued = npp * ach
                arith = ("2-plane fp16 split: 3 plane products per product" if pmode == 3 else
                         "3-plane bf16 split: 6 plane products per product")
                # `achieved` counts the ALGORITHMIC work of the launch (one N x N x N product = 2 n^3 flop) against the peak
                # of the pipe it runs on; the matrix cores issue `npp` plane products for it (`issued_*`)
                out["roofline"] = {"bound": "mfma",
                                   "kernel": "split2_m16_kernel (P1 = (H Kf H) Xc, the one N x N x N product of a low-rank "
                                             "linear_HSIC step, as a " + arith + ", fp32 accumulate, fp32-level error; one "
                                             "product per step -- `launch` below = that product: the kernel's grid cut behind whole "
                                             "rounds of the chip for the tail pass that runs beside its last rounds, + the split-K "
                                             "launch of the ragged round and the sum of its slabs; timed from the first to the last)",
                                   "achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "frac": ach / PEAK_BF16_MFMA_TFLOPS,
                                   "algorithmic_flop_per_launch": st["flops"] / st["launches"],      # 2 n^2 x the rank's rows
                                   "issued_flop_per_launch": npp * st["flops"] / st["launches"], "issued_achieved": issued,
                                   "issued_frac": issued / PEAK_BF16_MFMA_TFLOPS,
                                   "fp32_mfma_peak_multiple": ach / PEAK_F32_MFMA_TFLOPS,
                                   "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_source,
                                   "launches_per_step": st["launches"] / a.steps, "avg_launch_ms": avg_ms,
                                   "gemm_share_of_step": st["ms"] / (1e3 * dt),
                                   "side_stream": side_stream}
                if live_error:
                    out["roofline"]["live_traffic_error"] = live_error
                if replay_ms or (alone is not None and "product_avg_launch_ms" in alone):
                    # the same launch with nothing beside it: 20 back-to-back replays on this engine's planes
                    # (mcgra_attack_product_replay), else the MCGRA_OVERLAP=0 run's launches: faster product, slower step
                    ams = replay_ms or alone["product_avg_launch_ms"]
                    out["roofline"]["alone"] = {"avg_launch_ms": ams, "achieved": 2.0 * n ** 3 / (ams * 1e-3) / 1e12,
                                                "frac": 2.0 * n ** 3 / (ams * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                                                "issued_frac": 2.0 * npp * n ** 3 / (ams * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                                                "how": "20 back-to-back replays" if replay_ms else "MCGRA_OVERLAP=0 run"}
                    if alone is not None and "value" in alone:
                        out["roofline"]["alone"]["steps_per_s_without_side_stream"] = alone["value"]
                        out["roofline"]["alone"]["launch_ms_without_side_stream"] = alone.get("product_avg_launch_ms")
            elif eng.gram_split_steps() > 0:
                # Gram evaluation (MCGRA_NO_LOWRANK=1, masked or GAT / SAGE steps) on the same 2-plane fp16 kernel: four
                # launches per step; each counted as the reference's dense n x n x n product (2 n^3 flop) although the two
                # Gram launches compute only the tiles on or below the diagonal
                out["roofline"] = {"bound": "mfma",
                                   "kernel": "split2_m16_kernel (the four N x N x N products of the Gram evaluation of "
                                             "linear_HSIC as 2-plane fp16 splits: Kx = Xc Xc^T and Ky = Yc Yc^T on lower tiles "
                                             "with a mirrored store, then G_adjn += Ky' Xc and G_A1 += Kx' Yc)",
                                   "achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "frac": ach / PEAK_BF16_MFMA_TFLOPS,
                                   "algorithmic_flop_per_launch": st["flops"] / st["launches"],
                                   "fp32_mfma_peak_multiple": ach / PEAK_F32_MFMA_TFLOPS,
                                   "traffic": None, "traffic_unit": "bytes/launch",
                                   "launches_per_step": st["launches"] / a.steps, "avg_launch_ms": avg_ms,
                                   "gemm_share_of_step": st["ms"] / (1e3 * dt)}
            else:
                what = ("P1 = (H Kf H) Xc, the one N x N x N product of a low-rank linear_HSIC step: SYMM on lower tile "
                        "storage, one launch per step") if lowrank else ("N x N x N products of the Gram evaluation of "
                        "linear_HSIC: one batched SYRK pair + one batched SYMM pair launch per step")
                out["roofline"] = {"bound": "mfma", "kernel": "gemm_f32_kernel<128,128,32> (" + what + ")",
                                   "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch",
                                   "launches_per_step": st["launches"] / a.steps, "avg_launch_ms": avg_ms,
                                   "gemm_share_of_step": st["ms"] / (1e3 * dt)}
                if alone_ms:     # the same SYMM by itself right after the timed region
                    out["roofline"]["alone"] = {"avg_launch_ms": alone_ms,
                                                "achieved": 2.0 * n ** 3 / (alone_ms * 1e-3) / 1e12,
                                                "frac": 2.0 * n ** 3 / (alone_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
        elif measure == "KL" and eng_path["fused_steps"] > 0 and eng_path["general_steps"] == 0 and world == 1:
            # the fused KL step: the fused MSELoss step's passes (below) with softmax(feature_adj) in feature_adj's place, plus the
            # second per-pair pass over M for the row statistics of calc_kl (k_decode_stats): one more p
            p = 4.0 * n * n
            passes = nl + (nl - 1) + 2 + 2 + 4 + (nl if monitor and not out["config"]["forward_reuse"] else 0)
            ach = passes * p / (1e-3 * 1e3 * dt / a.steps) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "the fused KL step as a whole (k_tail_adam, the skinny products on M, k_tail_reduce, "
                               "k_decode_stats, k_decode_fly: no N x N x N product, no N x N intermediate)", "achieved": ach, "peak": 8000.0,
                               "unit": "GB/s", "frac": ach / 8000.0, "algorithmic_bytes_per_step": passes * p,
                               "traffic": live["iteration_bytes"] if live is not None else None,
                               "traffic_unit": "bytes/step", "traffic_source": live["iteration_how"] if live is not None else live_error}
        elif measure == "MSELoss" and eng_path["fused_steps"] > 0 and eng_path["general_steps"] == 0 and world == 1:
            # the fused MSELoss step has no N x N x N product: it is a chain of HBM-bound passes over the learnable adjacency.
            # Algorithmic bytes per step = the passes its formulation cannot do without, p = 4 n^2 bytes each: L forward products
            # + (L - 1) backward products on M, the decode's read of M (adj_norm_ij per pair), the tail's first pass (M and G2 on
            # the lower tile pairs, feature_adj whole: 2 p), the Adam pass (G2, M, both moments read on the lower pairs; M written
            # whole, the moments on the lower pairs: 4 p), + the monitoring forward's L products.
            p = 4.0 * n * n
            passes = nl + (nl - 1) + 1 + 2 + 4 + (nl if monitor and not out["config"]["forward_reuse"] else 0)
            ach = passes * p / (1e-3 * 1e3 * dt / a.steps) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "the fused MSELoss step as a whole (k_tail_adam, the skinny products on M, k_tail_reduce, "
                               "k_decode_fly: no N x N x N product, no N x N intermediate)", "achieved": ach, "peak": 8000.0, "unit": "GB/s",
                               "frac": ach / 8000.0, "algorithmic_bytes_per_step": passes * p,
                               "traffic": live["iteration_bytes"] if live is not None else None, "traffic_unit": "bytes/step",
                               "traffic_source": live["iteration_how"] if live is not None else live_error}
        else:
            out["roofline"] = None
        # the rest of the step against the HBM roofline: PMC bytes per step outside the product launches (committed
        # profile) over the step time that is not the product (step without the side stream - product alone)
        sp = os.path.join(ROOT, "profiles", "r04_step_traffic.json")
        if world == 1 and alone is not None and "value" in alone and (live is not None or (a.workload == "synthetic-10k-hsic" and os.path.exists(sp))):
            bytes_out = live["outside_product_bytes_per_step"] if live is not None else json.load(open(sp))["outside_product_bytes_per_step"]
            ms_out = 1e3 / alone["value"] - alone.get("product_avg_launch_ms", 0.0)
            if ms_out > 0:
                out["step_outside_product"] = {"bound": "hbm", "bytes_per_step": bytes_out, "ms_per_step": ms_out,
                                               "achieved": bytes_out / (ms_out * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                               "frac": bytes_out / (ms_out * 1e-3) / 1e9 / 8000.0,
                                               "bytes_source": live["how"] if live is not None else "committed PMC passes (profiles/r04_step_traffic.json)",
                                               "note": "launch- and VALU-bound kernels included: ~75 node-level launches per step"}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
