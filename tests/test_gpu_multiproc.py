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


@pytest.mark.parametrize("world,n,wp", [(2, 1100, None), (3, 1100, None), (2, 1283, NXN_ONLY)])
def test_row_block_ranks_as_processes_match_the_monolithic_step(pkg, world, n, wp, tmp_path):
    """Union of the ranks' rows == the monolithic fused step, step by step (3 steps + monitor, adopted forward), with the
    collectives executed by a process group between separate processes; scalars identical on every rank."""
    spec = dict(n=n, widths=(16, 16) if wp is None else (16, 8), seed=n, steps=3, weight_param=wp)
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
            assert np.allclose(r[f"scal{t}"], ref, rtol=3e-5, atol=1e-6 * max(1.0, abs(a["loss"]))), (t, r[f"scal{t}"], ref)
    assert all(int(r["fused_steps"]) == 3 and int(r["general_steps"]) == 0 for r in ranks) and mono.fused_steps() == 3
    assert all(int(r["exchanges"]) >= 3 * 8 for r in ranks)


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


def test_bench_world2_branch_on_a_shared_gpu(tmp_path):
    """bench.py's `world > 1` branch (RowBlockPlan + HipShardBackend + ShardedStepper inside the timing contract) driven
    by two processes for 2 timed steps; MCGRA_BENCH_SHARED_GPU=1 selects gloo + host staging on cuda:0."""
    out = str(tmp_path / "bench2")
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "synthetic-4k-hsic", "--no-shard-probe"]
    _run_ranks(W.run_bench_rank, 2, (argv,), out, timeout=900)
    line = json.load(open(out + ".json"))
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "strong"
    assert line["value"] > 0 and abs(line["value"] * line["ms_per_step"] - 1e3) < 1e-6 * 1e3
    assert line["config"]["fused_steps"] == 3 and line["config"]["general_steps"] == 0
    assert line["collectives_per_step"] >= 8
    assert 0.5 < line["auc"] < 1.0
