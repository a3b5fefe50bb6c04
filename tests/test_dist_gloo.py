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
