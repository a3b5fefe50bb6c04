"""N > 1 plumbing of bench.py on CPU: two gloo ranks run the timing contract (barrier, exactly K timed steps,
MAX over ranks, whole-job aggregate).  The per-rank work is a stand-in sleep: the HIP step itself cannot run
here, and round 1 ships independent replicas per rank (no data-path collective to test)."""
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


# ---- row-block sharded step -------------------------------------------------------------
def _run_sharded(world, case, steps, tmp_path):
    import subprocess
    out = str(tmp_path / "shard")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_shard_worker.py"), case, str(steps), out]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    import numpy as np
    return [np.load(f"{out}.rank{k}.npz") for k in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_step_matches_single_process(world, tmp_path):
    """ShardedStepper (product code) over gloo: every rank ends with the same adjacency as the unsharded oracle.
    world=3 on N=200 leaves the last rank without rows (it still joins the collectives)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    case, steps = "s200_hsic_init", 3
    ranks = _run_sharded(world, case, steps, tmp_path)
    z = helpers.load_case(case)
    ref = helpers.oracle_from(z)
    ref_losses = [ref.step(helpers.noise_of(z, t))["loss"] for t in range(steps)]
    for k, r in enumerate(ranks):
        assert np.array_equal(r["M"], ranks[0]["M"]), f"rank {k} diverged from rank 0"       # replicas stay bit-identical
        assert np.abs(r["M"] - ref.M).max() < 2e-5
        assert np.allclose(r["losses"], ref_losses, rtol=2e-5)
    rows = np.array([r["rows"] for r in ranks])
    assert rows[0, 2] % (128 * world) == 0 and rows[-1, 1] == rows[0, 2]                     # equal whole-tile blocks cover n_pad
    assert (rows[1:, 0] == rows[:-1, 1]).all()


def test_row_block_plan():
    import mcgra_loader
    mcgra_loader.load()
    from mc_gra_amd.sharded import RowBlockPlan
    p = [RowBlockPlan(10000, 8, r) for r in range(8)]
    assert all(q.rows_per_rank == 1280 and q.n_pad == 10240 for q in p)
    assert [q.row_begin for q in p] == [1280 * r for r in range(8)] and p[7].row_end == 10240 and p[7].has_rows
    q = RowBlockPlan(200, 4, 3)
    assert q.rows_per_rank == 256 and q.n_pad == 1024 and not q.has_rows
    assert RowBlockPlan(2708, 1, 0).row_end == 2816
