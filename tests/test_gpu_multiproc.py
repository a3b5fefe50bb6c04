"""The row-block sharded HIP engine under a REAL process group: world 2 / 3 processes on the box's one GPU, each a
HipShardBackend rank of one attack, ShardedStepper over gloo with host-staged arena slices.
The children come from a fork server started in conftest.py before this process touched the GPU."""
import json
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest

from tests import helpers as H
from tests import _hip_shard_worker as W

pytestmark = pytest.mark.gpu

NXN_ONLY = (0.01, 0.01, 0, 0, 0, 10, 10, 0, 0, 0)


@pytest.fixture(scope="module")
def pkg():
    import mcgra_loader
    return mcgra_loader.load()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(target, world, args, out, timeout=600):
    ctx = mp.get_context("forkserver")
    port = _free_port()
    ps = [ctx.Process(target=target, args=(r, world, port) + args + (out,)) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(timeout)
    errs = []
    for r, p in enumerate(ps):
        if p.is_alive():
            p.kill()
            errs.append(f"rank {r}: timeout")
        ep = f"{out}.rank{r}.err"
        if os.path.exists(ep):
            errs.append(f"rank {r}:\n" + open(ep).read())
        elif p.exitcode != 0:
            errs.append(f"rank {r}: exit code {p.exitcode}")
    assert not errs, "\n".join(errs)


def _two_devices():
    import torch
    return torch.cuda.device_count() >= 2      # (counting devices does not initialise the GPU)


needs_two_gpus = pytest.mark.skipif(not _two_devices(), reason="needs two MI355X: one device per rank, backend nccl (RCCL)")


# (2, 4096, .., "l3"): BASELINE.json configs[4]'s shape family -- a 3-layer victim on 256 attributes -- across a process boundary
# (three layers were sharded at n = 1030 in lockstep only); "kl" / "mse": the elementwise fused steps
@pytest.mark.parametrize("world,n,wp,kind", [(2, 1100, None, ""), (3, 1100, None, ""), (2, 1283, NXN_ONLY, ""), (2, 4096, None, "l3"),
                                             (2, 1100, None, "kl"), (3, 1100, None, "mse"),
                                             pytest.param(2, 1100, None, "rccl", marks=needs_two_gpus),
                                             pytest.param(2, 4096, None, "l3+rccl", marks=needs_two_gpus),
                                             pytest.param(2, 1100, None, "kl+rccl", marks=needs_two_gpus)])
def test_row_block_ranks_as_processes_match_the_monolithic_step(pkg, world, n, wp, kind, tmp_path):
    """Union of the ranks' rows == the monolithic fused step, step by step (3 steps + monitor, adopted forward), with the
    collectives executed by a process group between separate processes; scalars identical on every rank.  The "rccl" cases run
    where the box has two devices: rank k on cuda:k, backend nccl, the exchanges on views of the engine's arena in device memory --
    `all_gather_into_tensor` in place on the rank's own chunk, `all_to_all_single` on byte slices (sharded.py:73-83)."""
    spec = dict(n=n, widths=(16, 16) if wp is None else (16, 8), seed=n, steps=3, weight_param=wp, rccl="rccl" in kind)
    if "l3" in kind:
        spec.update(widths=(16, 16, 16), nfeat=256)
    if "kl" in kind:
        spec["measure"] = "KL"
    if "mse" in kind:
        spec["measure"] = "MSELoss"
    elem = "kl" in kind or "mse" in kind
    out = str(tmp_path / "mp")
    _run_ranks(W.run_rank, world, (spec,), out)
    z = W.case_of(spec)
    mono = H.engine_from(pkg, z)
    lr = float(z["lr"])
    ranks = [np.load(f"{out}.rank{r}.npz") for r in range(world)]
    for t in range(3):
        a = mono.step(want_scalars=True); mono.monitor()
        M = mono.buffer("M").cpu().numpy()
        rows = np.concatenate([r[f"rows{t}"] for r in ranks if r[f"rows{t}"].shape[0] > 0], 0)
        assert rows.shape == M.shape
        assert float((np.abs(rows - M) > 0.05 * lr).mean()) < 2e-3, t        # Adam: +-lr on noise-level gradients
        assert float(np.abs(rows - rows.T).max()) == 0.0, "ranks must agree on mirrored entries bit for bit"
        ref = np.array([a[k] for k in ("loss", "c1", "c2", "c6", "c7", "c9", "c10", "nll", "clamp_sum")])
        for r in ranks:
            assert np.array_equal(r[f"scal{t}"], ranks[0][f"scal{t}"]), "scalars are identical on every rank"
            # ("l3": 256 binary attributes through three un-normalised layers -- the H_A prior runs on the raw adjacency -- give
            # activations of 1e3 ... 1e4 and c9 = |Xc^T Yc|^2 = 5e14, a squared CENTRED cross-covariance: the ranks' forward sums its
            # products' slabs in another order than the monolithic step and the centring amplifies that to 7e-5 of the term)
            assert np.allclose(r[f"scal{t}"], ref, rtol=3e-4 if "l3" in kind else 3e-5, atol=1e-6 * max(1.0, abs(a["loss"]))), (t, r[f"scal{t}"], ref)
    assert all(int(r["fused_steps"]) == 3 and int(r["general_steps"]) == 0 for r in ranks) and mono.fused_steps() == 3
    nl = len(spec["widths"])
    # 8 per step + monitor at L = 2, + 1: loss terms asked for, + the first step's own forward (no monitor call in front of it);
    # the elementwise steps have no product to hand over and no low-rank factors: 6 (MSELoss) / 7 (KL: + the row statistics)
    per_step = 2 * nl + (4 if not elem else (3 if "kl" in kind else 2))
    assert all(int(r["exchanges"]) == 3 * (per_step + 1) + (nl + 1) for r in ranks), [int(r["exchanges"]) for r in ranks]


def test_row_block_processes_masked_steps_then_fused_again(pkg, tmp_path, monkeypatch):
    """Two decode-masked steps (M / am / av all-gathered between the processes, the general path redoes the step replicated
    on every rank), then the decode stops masking and the ranks take fused steps again."""
    import torch
    wp = (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 0)
    spec = dict(n=600, widths=(16, 16), seed=9, steps=4, weight_param=wp, masked_steps=2, env={"MCGRA_SPLIT_BF16": "3"})
    out = str(tmp_path / "mpm")
    _run_ranks(W.run_rank, 2, (spec,), out)
    monkeypatch.setenv("MCGRA_SPLIT_BF16", "3")
    z = W.case_of(spec)
    mono = H.engine_from(pkg, z)
    w = W.masked_weights(z)
    mono.set_model(w.W, w.b, w.Wlin, w.blin, w.Ws)
    ranks = [np.load(f"{out}.rank{r}.npz") for r in range(2)]
    lr = float(z["lr"])
    for t in range(4):
        if t == 2:
            w0 = H.weights_from(z)
            mono.set_model(w0.W, w0.b, w0.Wlin, w0.blin, w0.Ws)
        mono.step(); mono.monitor()
        M = mono.buffer("M").cpu().numpy()
        rows = np.concatenate([r[f"rows{t}"] for r in ranks], 0)
        if t < 2:
            assert np.array_equal(rows, M), t            # the general path, replicated: bit for bit
        else:
            assert float((np.abs(rows - M) > 0.05 * lr).mean()) < 2e-3, t
    assert all(int(r["general_steps"]) == 2 and int(r["fused_steps"]) == 2 for r in ranks)
    assert mono.path_stats()["general_steps"] == 2 and mono.fused_steps() == 2


def _one_rank_auc(workload, steps, seed=0):
    """The same workload, start and step count on ONE rank (monolithic fused step), through bench.py's own helpers."""
    import torch
    import mcgra_loader
    import bench
    pkg = mcgra_loader.load()
    dev = torch.device("cuda:0")
    eng, inp, adj_dev = bench.build_engine(pkg, torch, dev, workload, seed)
    for _ in range(steps):
        eng.step(); eng.monitor()
    lab = torch.as_tensor(inp["labels"], device=dev)
    final = eng.finalize(0, eng.buffer("HA"), eng.buffer("YA"), (lab[:, None] == lab[None, :]).float())
    auc = bench.gpu_auc(adj_dev, final, torch)
    fused = eng.fused_steps()
    del eng, final
    torch.cuda.empty_cache()
    return auc, fused


def test_bench_world2_branch_on_a_shared_gpu(tmp_path):
    """bench.py's `world > 1` branch (RowBlockPlan + HipShardBackend + ShardedStepper inside the timing contract) driven
    by two processes for 2 timed steps; MCGRA_BENCH_SHARED_GPU=1 selects gloo + host staging on cuda:0.  The recovered-
    adjacency AUC of the 2-rank attack equals the 1-rank attack's on the same workload, start and step count."""
    out = str(tmp_path / "bench2")
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "synthetic-4k-hsic", "--no-shard-probe"]
    _run_ranks(W.run_bench_rank, 2, (argv,), out, timeout=900)
    line = json.load(open(out + ".json"))
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "strong"
    assert line["value"] > 0 and abs(line["value"] * line["ms_per_step"] - 1e3) < 1e-6 * 1e3
    assert line["config"]["fused_steps"] == 3 and line["config"]["general_steps"] == 0
    assert line["collectives_per_step"] == 8           # per timed step + monitor at L = 2 (round 3: 11)
    auc1, fused1 = _one_rank_auc("synthetic-4k-hsic", 3)
    assert fused1 == 3 and 0.5 < auc1 < 1.0
    assert abs(line["auc"] - auc1) <= 1e-6, (line["auc"], auc1)
    # the fields that make a multi-GPU line explain itself: collective time by kind (HIP events, mean / max over ranks), every
    # rank's product time, the compute-only pass (collectives answered by the rank's own data), the exposed communication, and
    # the state check: rank 0 replays the attack on one rank
    m = line["multi_rank"]
    assert m["collectives_timed_per_step"] == 8 and m["comm_ms_per_step"]["max"] >= m["comm_ms_per_step"]["mean"] > 0
    assert m["alltoall_ms_per_step"]["mean"] > 0 and m["allgather_ms_per_step"]["mean"] > 0
    assert len(m["product_ms_per_rank"]) == 2 and all(x > 0 for x in m["product_ms_per_rank"])
    assert m["compute_only_ms_per_step"] > 0 and m["compute_only_steps_all_fused"]
    assert abs(m["exposed_comm_ms_per_step"] - (line["ms_per_step"] - m["compute_only_ms_per_step"])) < 1e-9
    sc = m["state_check"]
    assert sc["ok"] is True and sc["abs_diff"] <= 1e-6 and abs(sc["auc_one_rank"] - auc1) <= 1e-9 and sc["steps"] == 3


@pytest.mark.parametrize("workload,ncoll", [("cora-shape-mse", 6), ("cora-shape-kl", 7)])
def test_bench_world2_branch_with_the_fused_mseloss_step(tmp_path, workload, ncoll):
    """The same branch on an MSELoss workload (Cora-shaped: BASELINE.json configs[0]'s measure) and on a KL one (round 6): the
    row-block MSELoss / KL steps exchange no N x N data -- six / seven all-gathers per step + monitor, no all-to-all, no product --
    and the 2-rank attack's AUC is the 1-rank attack's (rank 0's replay behind the timed region says so too)."""
    out = str(tmp_path / "bench2mse")
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload, "--no-shard-probe"]
    _run_ranks(W.run_bench_rank, 2, (argv,), out, timeout=900)
    line = json.load(open(out + ".json"))
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "strong"
    assert line["config"]["fused_steps"] == 4 and line["config"]["general_steps"] == 0
    assert line["collectives_per_step"] == ncoll
    auc1, fused1 = _one_rank_auc(workload, 4)
    assert fused1 == 4 and abs(line["auc"] - auc1) <= 1e-5, (line["auc"], auc1)
    m = line["multi_rank"]
    assert m["collectives_timed_per_step"] == ncoll and m["alltoall_ms_per_step"]["max"] == 0 and m["allgather_ms_per_step"]["mean"] > 0
    assert m["product_ms_per_rank"] == [None, None] and m["compute_only_steps_all_fused"] and not m.get("errors")
    assert m["state_check"]["ok"] is True


@needs_two_gpus
def test_bench_gpus2_over_rccl_on_two_devices(tmp_path):
    """`python bench.py --gpus 2 --steps 5` as the driver's multi-GPU tier runs it, on a box with two devices: no MCGRA_SHARED_GPU,
    backend nccl (RCCL over xGMI), rank k on cuda:k.  The line explains itself (collective time by kind, the all-to-all of the
    product's tile blocks among them) and rank 0's replay of the attack on one rank agrees with the two-rank state."""
    out = str(tmp_path / "rccl2")
    argv = ["--gpus", "2", "--steps", "5", "--warmup", "2", "--workload", "synthetic-4k-hsic", "--no-shard-probe"]
    ctx = mp.get_context("forkserver")
    p = ctx.Process(target=W.run_bench_plain, args=(argv, out, {"OMP_NUM_THREADS": "2"}))
    p.start()
    p.join(900)
    if p.is_alive():
        p.kill()
        pytest.fail("timeout")
    err = out + ".rank0.err"
    assert p.exitcode == 0, open(err).read() if os.path.exists(err) else f"exit code {p.exitcode}"
    lines = [ln for ln in open(out + ".stdout").read().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["scaling"] == "strong"
    assert line["config"]["fused_steps"] == 7 and line["config"]["general_steps"] == 0
    assert "gloo" not in line["config"]["parallelism"]
    m = line["multi_rank"]
    assert m["state_check"]["ok"] is True
    assert m["alltoall_ms_per_step"]["mean"] > 0 and m["allgather_ms_per_step"]["mean"] > 0


def test_plain_bench_gpus2_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` with NO launcher environment (--gpus was parsed and ignored,
    so the command measured one GPU): bench.main starts the two ranks itself (torch.distributed.run from a process that has
    not touched the GPU), prints rank 0's line as the only stdout line, and that line says n_gpus == 2.  The caller here is a
    fork-server child (this pytest process has initialised the GPU and must not fork + exec)."""
    out = str(tmp_path / "plain2")
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "synthetic-4k-hsic", "--no-shard-probe"]
    ctx = mp.get_context("forkserver")
    p = ctx.Process(target=W.run_bench_plain, args=(argv, out, {"MCGRA_BENCH_SHARED_GPU": "1", "OMP_NUM_THREADS": "2"}))
    p.start()
    p.join(900)
    if p.is_alive():
        p.kill()
        pytest.fail("timeout")
    err = out + ".rank0.err"
    assert p.exitcode == 0, open(err).read() if os.path.exists(err) else f"exit code {p.exitcode}"
    lines = [ln for ln in open(out + ".stdout").read().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "strong"
    assert line["config"]["fused_steps"] == 3 and line["config"]["general_steps"] == 0
    assert "gloo" in line["config"]["parallelism"]          # the shared-GPU test mode says so in the line
    assert 0.5 < line["auc"] < 1.0


def test_row_block_processes_at_the_headline_size(pkg, tmp_path):
    """Two row-block ranks of `synthetic-10k-hsic` as separate processes (gloo, host-staged arena slices) against the
    monolithic fused step and the float64 fixture -- the sizes at which the product's slab split-K tail, planes_mm
    (n >= 8192) and 20-panel row blocks run, none of which had crossed a process boundary before.
    Bars: those of test_sharded_ranks_match_monolithic_at_10k for the union of rows and the scalars; the first-step
    gradient of the union within 3e-4 of the reference's own code in float64 (bench10k_hsic_ref64.npz: run_g64ref)."""
    import torch
    import bench
    wl = "synthetic-10k-hsic"
    n = bench.WORKLOADS[wl][0]
    z64 = np.load(os.path.join(H.GOLDEN, "bench10k_hsic_ref64.npz"))
    z = np.load(os.path.join(H.GOLDEN, "bench10k_hsic.npz"))
    pi, pj = H.tril_pos(z64["packed_pos"])
    spec = dict(workload=wl, seed=int(z["seed"]), steps=2, pos_i=pi.tolist(), pos_j=pj.tolist())
    out = str(tmp_path / "mp10k")
    _run_ranks(W.run_workload_rank, 2, (spec,), out, timeout=1500)
    ranks = [np.load(f"{out}.rank{r}.npz") for r in range(2)]
    assert all(int(r["fused_steps"]) == 2 and int(r["general_steps"]) == 0 for r in ranks)
    # first-step gradient of the union at the fixture's sampled positions against the float64 evaluation
    g = np.zeros(len(pi), np.float32)
    seen = np.zeros(len(pi), bool)
    for r in ranks:
        g[r["own"]] = r["g0"]; seen |= r["own"]
    assert seen.all()
    gmax = float(z64["run_g64ref_absmax"])
    err = float(np.abs(g - z64["run_g64ref"]).max()) / gmax
    assert err <= 3e-4, err
    dev = torch.device("cuda:0")
    mono, _, _ = bench.build_engine(pkg, torch, dev, wl, int(z["seed"]))
    lr = bench.workload_lr(wl, n)
    for t in range(2):
        a = mono.step(want_scalars=True); mono.monitor()
        rows = torch.cat([torch.as_tensor(np.fromfile(f"{out}.rank{r}.rows{t}.f32", np.float32).reshape(-1, n), device=dev)
                          for r in range(2)], 0)
        ref = mono.buffer("M")
        assert rows.shape == ref.shape
        assert float(((rows - ref).abs() > 0.05 * lr).float().mean()) < 2e-3, f"step {t}"
        assert float((rows - rows.T).abs().max()) == 0.0, "ranks must agree on mirrored entries bit for bit"
        del rows, ref
        names = ("loss", "c1", "c2", "c6", "c7", "c9", "c10", "nll", "clamp_sum")
        assert np.array_equal(ranks[0][f"scal{t}"], ranks[1][f"scal{t}"]), "scalars are identical on every rank"
        for k in ("loss", "c1", "c2", "c9", "c10"):
            assert ranks[0][f"scal{t}"][names.index(k)] == pytest.approx(a[k], rel=1e-4), (t, k)
        if t == 0:      # ... and the monolithic engine's own first gradient at the same positions
            gm = mono.buffer("G_sym")[torch.as_tensor(pi, device=dev), torch.as_tensor(pj, device=dev)].cpu().numpy()
            assert float(np.abs(g - gm).max()) / gmax <= 1e-4
    for r in range(2):
        for t in range(2):
            os.remove(f"{out}.rank{r}.rows{t}.f32")


def test_plain_bench_measures_its_traffic_live(tmp_path):
    """`python bench.py` as the driver runs it at N = 1 (here: the 4k workload, 5 steps): before it touches the GPU the process
    runs two children of itself under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` and reports the product's memory-side
    bytes per launch and the rest of the step's per step from THEM (the line quoted committed profiles) --
    `traffic_source` says so.  The caller is a fork-server child: this pytest process has initialised the GPU."""
    import shutil
    out = str(tmp_path / "live")
    argv = ["--steps", "5", "--warmup", "1", "--workload", "synthetic-4k-hsic", "--no-cpu-baseline"]
    ctx = mp.get_context("forkserver")
    p = ctx.Process(target=W.run_bench_plain, args=(argv, out, {"OMP_NUM_THREADS": "2"}))
    p.start()
    p.join(900)
    if p.is_alive():
        p.kill()
        pytest.fail("timeout")
    err = out + ".rank0.err"
    assert p.exitcode == 0, open(err).read() if os.path.exists(err) else f"exit code {p.exitcode}"
    lines = [ln for ln in open(out + ".stdout").read().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    r = line["roofline"]
    assert line["n_gpus"] == 1 and line["config"]["fused_steps"] == 6 and r["bound"] == "mfma"
    assert "traffic_source" in r
    # (a profiler that is missing, or whose child pass fails on this box, makes the line fall back to the committed passes and
    #  say so: that is the designed behaviour, not a parity failure -- the accounting itself is pinned on the CPU,
    #  tests/test_dist_gloo.py::test_live_traffic_accounting)
    if shutil.which("rocprofv3") and (r["traffic_source"] or "").startswith("rocprofv3 --pmc"):
        n = 4096
        compulsory = 2 * n * n * 4 + n * n * 4           # both operands' planes once + the result
        assert compulsory <= r["traffic"] <= 12 * compulsory, (r["traffic"], compulsory)
        so = line["step_outside_product"]
        assert so["bytes_source"].startswith("rocprofv3 --pmc") and 8 * n * n * 4 <= so["bytes_per_step"] <= 40 * n * n * 4


# ---- the sharded step behind the class surface and main.py (north_star: "keep the PGDAttack / BaseAttack class surface and
# main.py entry" AND "partition ... row-block across up to 8 MI355X") ------------------------------------------------------
@pytest.mark.parametrize("fixture,kind", [("cora_hsic_sparse", ""), ("cora_mse_short", ""), ("cora_hsic_sparse", "kl"),
                                          pytest.param("cora_hsic_sparse", "rccl", marks=needs_two_gpus)])
def test_pgdattack_class_shards_the_attack_under_a_process_group(pkg, tmp_path, fixture, kind):
    """Two processes call PGDAttack.attack on Cora (reference-trained weights of the fixture: HSIC from the sparse start, and the
    README's MSELoss configuration -- BASELINE.json configs[0] -- for 20 epochs) under a process group: the class builds
    RowBlockPlan + HipShardBackend + ShardedStepper itself, every step is a fused row-block step (the MSELoss one exchanges no
    N x N data at all), both ranks return the SAME modified_adj, its AUC equals the 1-process run's to 1e-6 (MSELoss: 1e-5) and the
    REFERENCE's (the fixture's) to 1e-4."""
    # "kl": the fixture's graph, reference-trained weights, sparse start and weights (w1 = w2 = 0.01: both N x N terms) with calc =
    # calc_kl (round 6: the fused KL step, sharded like the MSELoss one) -- no reference AUC for that combination: the two-rank run
    # is held to the one-process run.  (From the MSELoss fixture's start, adj_changes = 0 with lr = 0.01, the first KL step is a
    # sign step on gradients of 1e-7 k / n^2: two evaluations that differ in summation order are 1.5e-4 of AUC apart after 8 epochs.)
    kl = kind == "kl"
    out = str(tmp_path / "cls")
    _run_ranks(W.run_class_rank, 2, (dict(name=fixture, rccl=kind == "rccl", measure="KL" if kl else None, epochs=8 if kl else None),), out)
    z, final1, auc1, model1 = W.run_cora_class(fixture, epochs=8 if kl else None, measure="KL" if kl else None)
    assert model1.history["path"]["sharded_world"] == 1
    ranks = [np.load(f"{out}.rank{r}.npz") for r in range(2)]
    epochs = int(z["epochs"])
    if kl:
        assert model1.history["path"]["fused_steps"] == epochs and model1.history["path"]["general_steps"] == 0
    for r in ranks:
        assert int(r["sharded_world"]) == 2 and int(r["fused_steps"]) == epochs and int(r["general_steps"]) == 0
        assert int(r["collectives"]) >= (7 if kl else 8 if "hsic" in fixture else 6) * epochs
        # (the row-block ranks sum the decode's slabs and the tail's tiles in another order than the one-process step: not the
        # same bits.  HSIC from the sparse start keeps the two runs' AUC within 1e-6; the 20 MSELoss epochs amplify such
        # differences the way they do between the reference, the fp32 and the fp64 oracle -- 6e-6 apart at this horizon,
        # test_cora_mse_checkpoints -- measured 3.8e-6.  Each rank's gradient rows against the one-process step, per step:
        # test_sharded_mse_ranks_match_monolithic_step.)
        assert abs(float(r["auc"]) - auc1) <= (1e-6 if "hsic" in fixture and not kl else 1e-5), (float(r["auc"]), auc1)
        if not kl:
            assert abs(float(r["auc"]) - float(z["auc"])) <= 1e-4, (float(r["auc"]), float(z["auc"]))
        assert len(r["acc_test"]) == epochs and np.allclose(r["acc_test"], model1.history["acc_test"])
    assert np.array_equal(ranks[0]["final_sample"], ranks[1]["final_sample"]), "every rank returns the same modified_adj"
    assert float(ranks[0]["final_sum"]) == float(ranks[1]["final_sum"])
    assert float(ranks[0]["adj_changes_sum"]) == float(ranks[1]["adj_changes_sum"]) > 0      # adj_changes readable after the run
    sp = z["sample_pos"]
    # (entries that moved by a different +-lr somewhere along the run: under 0.1 % from the sparse HSIC start, 0.5 % after the 20
    # MSELoss epochs)
    assert np.mean(np.abs(ranks[0]["final_sample"] - final1[sp[:, 0], sp[:, 1]]) > 1e-3) < (1e-3 if "hsic" in fixture and not kl else 1e-2)


def test_main_entry_under_a_launcher_shards_the_attack(pkg, tmp_path, monkeypatch):
    """`torchrun --nproc-per-node 2 main.py --dataset cora --measure HSIC ...` (two fork-server processes with the launcher's
    environment on the box's one GPU, MCGRA_SHARED_GPU=1): main.py joins the group before it touches the GPU, trains the
    victim, rank 0's weights go to every rank, PGDAttack.attack runs ONE row-block sharded attack; the AUC of both ranks
    equals the plain 1-process `main.py` run's to 1e-6 and only rank 0 writes the log."""
    import scipy.sparse as sp
    z = H.load_cora("cora_hsic_sparse")
    A = sp.csr_matrix(np.triu(z["adj"], 1)); X = sp.csr_matrix(z["features"])
    root = tmp_path / "dataset"; root.mkdir()
    np.savez(root / "cora.npz", adj_data=A.data, adj_indices=A.indices, adj_indptr=A.indptr, adj_shape=A.shape,
             attr_data=X.data, attr_indices=X.indices, attr_indptr=X.indptr, attr_shape=X.shape, labels=z["labels"])
    n = z["adj"].shape[0]
    argv = W.main_args(root)
    cwd2 = tmp_path / "w2"; cwd2.mkdir()
    out = str(tmp_path / "main2")
    ctx = mp.get_context("forkserver")
    port = _free_port()
    ps = [ctx.Process(target=W.run_main_rank, args=(r, 2, port, argv, n, str(cwd2), out)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(900)
    errs = [open(f"{out}.rank{r}.err").read() for r in range(2) if os.path.exists(f"{out}.rank{r}.err")]
    assert not errs and all(p.exitcode == 0 for p in ps), "\n".join(errs)
    res2 = [json.load(open(f"{out}.rank{r}.json")) for r in range(2)]
    cwd1 = tmp_path / "w1"; cwd1.mkdir()
    monkeypatch.chdir(cwd1)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    res1 = W.run_main(argv, n)
    assert res1["path"]["sharded_world"] == 1 and res1["path"]["fused_steps"] == 6
    for r in res2:
        assert r["path"]["sharded_world"] == 2 and r["path"]["fused_steps"] == 6 and r["path"]["general_steps"] == 0
        assert abs(r["auc_all"] - res1["auc_all"]) <= 1e-6, (r["auc_all"], res1["auc_all"])
    assert res2[0]["auc_all"] == res2[1]["auc_all"]
    assert 0.8 < res1["auc_all"] < 0.95
    assert os.path.exists(cwd2 / "results" / "result.txt") and len(open(cwd2 / "results" / "result.txt").read().split("current parameter")) == 2


@pytest.mark.parametrize("name", ["readme_polblogs_hsic_hY", "readme_cora_mse_hy", "readme_citeseer_kl_all", "readme_polblogs_dp_y",
                                  "readme_cora_kde_Y", "readme_aids_cka_Y"])
def test_readme_lines_in_the_production_configuration(name, tmp_path):
    """The parity suite runs with MCGRA_AB=1 and MCGRA_KEEP_GSYM=1 set process-wide (tests/helpers.py) -- VERDICT round 5, weak #4:
    the configuration a user runs was covered by two tests.  Here one README line per measure (HSIC: fused low-rank step; MSELoss
    and KL: the fused elementwise steps; DP, KDE, CKA: the general step) goes through PGDAttack.attack in a child process WITHOUT
    those variables -- switches ignored, no mirrored gradient store -- and is held to the same bars as
    test_readme_line_through_the_class: ensemble sample, sum, AUC within 1e-4 of the reference's."""
    out = str(tmp_path / "prod")
    ctx = mp.get_context("forkserver")
    p = ctx.Process(target=W.run_production_line, args=(name, out))
    p.start()
    p.join(600)
    if p.is_alive():
        p.kill()
        pytest.fail("timeout")
    err = out + ".rank0.err"
    assert p.exitcode == 0 and os.path.exists(out + ".ok"), open(err).read() if os.path.exists(err) else f"exit code {p.exitcode}"
