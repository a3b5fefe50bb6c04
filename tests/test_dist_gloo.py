"""N > 1 plumbing on CPU (gloo): (i) bench.py's timing contract on two ranks (barrier, exactly K timed steps, MAX over
ranks) with a stand-in sleep as the per-rank work; (ii) the product's ShardedStepper / run_exchange over gloo at world
2 / 3 / 4 with a numpy rank that walks the engine's arena conventions.  The HIP row-block engine itself under a real
process group runs on the GPU box: tests/test_gpu_multiproc.py."""
import os
import socket
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    calls = {"n": 0}
    delay = 0.02 * (1 + rank)          # rank 1 is the slow rank

    def step():
        calls["n"] += 1
        time.sleep(delay)

    dt = bench.timed_region(step, steps=5, warmup=2, sync=lambda: None, world=world, dist=dist,
                            device=torch.device("cpu"), torch=torch)
    q.put((rank, calls["n"], dt, bench.aggregate_value(world, 5, dt)))
    dist.destroy_process_group()


def test_two_rank_timing_contract():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, dt0, v0), (r1, n1, dt1, v1) = out
    assert n0 == n1 == 7                              # W + K calls on every rank
    assert dt0 == dt1                                 # MAX over ranks is what every rank reports
    assert dt0 >= 5 * 0.04 * 0.95                     # bounded below by the slow rank's 5 timed steps
    assert abs(v0 - 2 * 5 / dt0) < 1e-9               # whole-job aggregate over both replicas


# ---- row-block sharded step: the exchange protocol over gloo ----------------------------------
def _run_sharded(world, n, steps, tmp_path):
    import subprocess
    out = str(tmp_path / "shard")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_shard_worker.py"), str(n), str(steps), out]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    import numpy as np
    return [np.load(f"{out}.rank{k}.npz") for k in range(world)]


@pytest.mark.parametrize("world,n", [(2, 600), (3, 600), (4, 700)])
def test_sharded_exchange_protocol_over_gloo(world, n, tmp_path):
    """ShardedStepper (product code) over gloo with a numpy rank that walks through the protocol's three collectives in
    the engine's arena conventions: the all-gathered node array, the all-reduced scalar and -- the step the engine's
    mirrored gradient depends on -- the all-to-all that turns every rank's COLUMN block of a product into its ROW block.
    world = 3 / 4 leave the last rank with fewer rows / without rows (it still joins every collective)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _shard_worker as W
    ranks = _run_sharded(world, n, 2, tmp_path)
    A, V, K = W.make_problem(n)
    Y = A @ V
    Cfull = K @ A
    for k, r in enumerate(ranks):
        r0, r1, n_pad = [int(x) for x in r["rows"]]
        assert n_pad % (256 * world) == 0 and int(r["exchanges"]) == 2 * 3
        assert np.allclose(r["Y"], Y, rtol=1e-5, atol=1e-5), f"rank {k}: all-gather"
        assert abs(float(r["s"]) - float((Y.astype(np.float64) ** 2).sum())) <= 1e-6 * float((Y.astype(np.float64) ** 2).sum())
        assert r["Crow"].shape == (max(min(r1, n) - r0, 0), n)
        if r1 > r0:
            assert np.allclose(r["Crow"], Cfull[r0:r1], rtol=1e-4, atol=1e-4), f"rank {k}: all-to-all row block"
    rows = np.array([r["rows"] for r in ranks])
    assert (rows[1:, 0] >= rows[:-1, 1]).all() and rows[rows[:, 0] < n, 1].max() == n


def test_row_block_plan():
    import mcgra_loader
    mcgra_loader.load()
    from mc_gra_amd.sharded import RowBlockPlan
    p = [RowBlockPlan(10000, 8, r) for r in range(8)]
    assert all(q.rows_per_rank == 1280 and q.n_pad == 10240 for q in p)
    assert [q.row_begin for q in p] == [1280 * r for r in range(8)] and p[7].row_end == 10000 and p[7].has_rows
    q = RowBlockPlan(200, 4, 3)
    assert q.rows_per_rank == 256 and q.n_pad == 1024 and not q.has_rows and q.row_end == q.row_begin
    assert RowBlockPlan(2708, 1, 0).row_end == 2708


# ---- bench.py --gpus N without a launcher: the ranks are started by bench.py itself ---------------------------------
def test_bench_gpus_flag_starts_ranks_and_refuses_a_mismatch(monkeypatch, capsys):
    """`python bench.py --gpus 4` with no RANK / WORLD_SIZE must start 4 ranks (the driver's own torch.distributed.run command
    line on 127.0.0.1) and forward rank 0's JSON line as the only stdout line; under a launcher --gpus must equal
    WORLD_SIZE.  (The launch itself runs on the GPU box: tests/test_gpu_multiproc.py::test_plain_bench_gpus2_...)"""
    import json
    import subprocess
    import bench
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        rank0 = json.dumps({"metric": "attack-steps/sec", "value": 1.0, "n_gpus": 4})
        return subprocess.CompletedProcess(cmd, 0, stdout="noise from a rank\n{not json\n" + rank0 + "\n", stderr=None)

    monkeypatch.setattr(subprocess, "run", fake_run)
    line = bench.main(["--gpus", "4", "--steps", "3", "--workload", "synthetic-4k-hsic"])
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(out) == 1 and json.loads(out[0]) == line and line["n_gpus"] == 4
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "3", "--workload", "synthetic-4k-hsic"]
    # a rank that fails, or a launch without a result line, is an error -- not a silent 1-GPU number
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: subprocess.CompletedProcess(cmd, 1, stdout="", stderr=None))
    with pytest.raises(SystemExit):
        bench.main(["--gpus", "2"])
    # under a launcher: --gpus must agree with WORLD_SIZE
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit, match="must agree"):
        bench.main(["--gpus", "4", "--no-cpu-baseline"])


def test_live_traffic_accounting(monkeypatch):
    """bench.live_traffic: the two `rocprofv3 --pmc` child passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only beside them) and
    the accounting of their CSVs -- 2 x FETCH_SIZE + WRITE_SIZE per kernel, the product's main grid per launch, everything
    else a step launches per step, one-off launches excluded -- on a faked profiler output."""
    import csv
    import shutil
    import subprocess
    import bench
    steps = 3
    nl = steps + 1                     # warm-up step + timed steps
    rows = {   # kernel -> (grid, launches, FETCH KB per launch, WRITE KB per launch)
        "void mcgra::(anonymous namespace)::split2_m16_kernel<0>(char const*)": (786432, nl, 3000.0, 400.0),
        "void mcgra::(anonymous namespace)::split2_m16_kernel<0>(char const*, int)": (131072, nl, 200.0, 60.0),      # split-K tail
        "void mcgra::(anonymous namespace)::split2_m16_kernel<0>(char const*, long)": (655360, nl, 500.0, 80.0),    # second part of a cut launch
        "mcgra::(anonymous namespace)::k_split3_reduce(float const*)": (131072, nl, 30.0, 20.0),                     # sum of the tail's slabs
        "mcgra::k_tail_adam(int)": (6310144, nl, 400.0, 900.0),
        "mcgra::k_planes_mm<2>(int)": (266240, 3 * nl, 200.0, 16.0),
        "mcgra::k_dd2_accum(int)": (2560000, nl, 999.0, 999.0),                      # finalize: not part of a step
        "mcgra::k_center_cols(int)": (2560000, 1, 999.0, 999.0),                     # set_graph: once
        "void at::native::vectorized_elementwise_kernel<4>(int)": (1000, 50, 5.0, 5.0),
    }
    seen = []

    def fake_run(cmd, **kw):
        seen.append(cmd)
        ctr = cmd[cmd.index("--pmc") + 1]
        assert "--kernel-trace" in cmd and not any(x in cmd for x in ("-s", "--sys-trace", "-r", "--runtime-trace", "--hip-trace"))
        # the program behind `--` is the interpreter ITSELF (resolved, an ELF binary): a shim that execs it would be an exec after
        # the profiler's library initialised the GPU
        assert cmd[cmd.index("--") + 1] == os.path.realpath(sys.executable) and "--no-live-traffic" in cmd and "--no-split-probe" in cmd
        d = os.path.join(cmd[cmd.index("-d") + 1], "host")
        os.makedirs(d)
        with open(os.path.join(d, "1_counter_collection.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Dispatch_Id", "Grid_Size", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
            i = 0
            for k, (g, n, fe, wr) in rows.items():
                for _ in range(n):
                    i += 1
                    w.writerow([i, g, k, ctr, fe if ctr == "FETCH_SIZE" else wr, 0, 1])
        return subprocess.CompletedProcess(cmd, 0)

    monkeypatch.setattr(shutil, "which", lambda name: "/opt/rocm/bin/rocprofv3" if name == "rocprofv3" else None)
    monkeypatch.setattr(subprocess, "run", fake_run)
    out = bench.live_traffic("synthetic-10k-hsic", 0, steps=steps)
    assert [c[c.index("--pmc") + 1] for c in seen] == ["FETCH_SIZE", "WRITE_SIZE"]
    # one product per step = every launch of the split kernel (parts of a cut launch, split-K tail) + the sum of its slabs
    assert out["product_bytes_per_launch"] == pytest.approx(((2 * 3000.0 + 400.0) + (2 * 200.0 + 60.0) + (2 * 500.0 + 80.0) + (2 * 30.0 + 20.0)) * 1024)
    assert out["outside_product_bytes_per_step"] == ((2 * 400.0 + 900.0) + 3 * (2 * 200.0 + 16.0)) * 1024
    # a pass that cannot run says why (the bench line then quotes the committed passes and carries `live_traffic_error`)
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: subprocess.CompletedProcess(cmd, 3))
    assert "exit code 3" in bench.live_traffic("synthetic-10k-hsic", 0)["error"]
    monkeypatch.setattr(sys, "executable", __file__)             # not an ELF binary: a launcher script
    assert "ELF" in bench.live_traffic("synthetic-10k-hsic", 0)["error"]
    monkeypatch.setattr(shutil, "which", lambda name: None)
    assert "rocprofv3" in bench.live_traffic("synthetic-10k-hsic", 0)["error"]


def _class_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mcgra_loader
    mcgra_loader.load()
    from mc_gra_amd import topology_attack as TA
    d, w, r, staged = TA._dist_group()
    t = torch.full((3,), float(rank + 1))
    TA._bcast(d, t, staged)
    q.put((rank, w, r, staged, t.tolist()))
    dist.destroy_process_group()


def test_class_finds_the_process_group_and_knows_what_it_can_shard():
    """PGDAttack.attack's view of torch.distributed (no GPU): no group -> one rank; a gloo group -> world, rank, host-staged
    exchanges, rank 0's tensors on every rank; and which configurations the row-block sharded fused step covers (the rest
    run replicated and say so)."""
    import mcgra_loader
    mcgra_loader.load()
    from mc_gra_amd import topology_attack as TA
    assert TA._dist_group() == (None, 1, 0, False)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_class_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out == [(0, 2, 0, True, [1.0, 1.0, 1.0]), (1, 2, 1, True, [1.0, 1.0, 1.0])]
    why = TA.PGDAttack._replicated_reason
    ok = dict(measure="HSIC", eps=0.0, ori_np=None, Ws=None, act="relu", head_act="none", loss_type="CE", n=2708,
              dims=[1433, 16, 16], w1=0.01, w2=0.01, num_edges=1e30)
    assert why(**ok) is None
    assert why(**dict(ok, measure="MSELoss", n=300, w1=0, w2=0)) is None          # the fused MSELoss step: any n >= 256
    assert why(**dict(ok, measure="KL", n=300)) is None                           # the fused KL step (round 6) likewise
    assert why(**dict(ok, dims=[1433, 16, 16, 16, 16])) is None                   # four 16-wide layers: summed widths 64
    # the create-time rule of csrc/attack.hip, term by term (`torchrun main.py --nlayers 5`, a victim with nhid = 24)
    for change, word in ((dict(dims=[1433, 16, 16, 16, 16, 16]), "summed layer widths 80"), (dict(dims=[1433, 24, 24]), "embedding width 24"),
                         (dict(dims=[1433, 32, 32, 32]), "summed"), (dict(dims=[1433, 16, 16, 16], emb_nlayer=3), None),
                         (dict(measure="MSELoss", dims=[1433, 24, 24]), "embedding width 24")):
        got = why(**dict(ok, **change))
        assert (got is None) if word is None else (got is not None and word in got), (change, got)
    for change, word in ((dict(measure="DP"), "DP"), (dict(measure="MSELoss", n=200), "256"), (dict(eps=0.1), "eps"), (dict(ori_np=object()), "ori_adj"),
                         (dict(Ws=[1]), "GraphSAGE"), (dict(act="elu"), "GAT"), (dict(n=300), "1024"),
                         (dict(dims=[10, 64, 64]), "width"), (dict(w1=0, w2=0), "w1"), (dict(num_edges=5.0), "projection"),
                         (dict(loss_type="CW"), "CW")):
        assert word in why(**dict(ok, **change)), change
