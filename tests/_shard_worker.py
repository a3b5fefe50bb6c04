"""World-size-N gloo worker: the product's ShardedStepper (mc-gra_amd/sharded.py) driving a numpy stand-in for a
row-block rank that goes through the three collectives of the protocol in the engine's own arena conventions:
all-gather of row chunks of an n_pad x c node array, all-reduce of fp64 scalars, all-to-all of [world][rpr][rpr] tile
blocks that turns the column block of a product into the row block.  Writes each rank's results to <out>.rank<k>.npz."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import torch.distributed as dist

import mcgra_loader


def make_problem(n, seed=3):
    rng = np.random.RandomState(seed)
    A = rng.rand(n, n).astype(np.float32)
    A = (A + A.T) * 0.5                     # symmetric, like the learnable adjacency
    V = rng.randn(n, 5).astype(np.float32)
    K = rng.randn(n, n).astype(np.float32)  # stands for the constant left factor of the N x N x N product
    return A, V, K


class ToyRank:
    """begin / next / scalars / arena of the stepper's backend protocol.  Program of one "step":
         Y[rows]  = A[rows, :] V                 -> ALLGATHER (chunk = rpr x c floats)
         s        = sum(Y[rows]^2)               -> ALLREDUCE_F64
         C[:, rows] = K A[:, rows] (= K A[rows, :]^T by symmetry), packed as [world][rpr][rpr] blocks
                                                 -> ALLTOALL, which delivers C[rows, :]
    """

    def __init__(self, plan, n):
        from mc_gra_amd import sharded as S
        self.S, self.plan, self.n = S, plan, n
        self.A, self.V, self.K = make_problem(n)
        w, rpr, c = plan.world, plan.rows_per_rank, self.V.shape[1]
        self.c = c
        self.off_y, self.off_s = 0, plan.n_pad * c * 4
        self.off_send = self.off_s + 256
        self.off_recv = self.off_send + w * rpr * rpr * 4
        self.arena = torch.zeros(self.off_recv + w * rpr * rpr * 4, dtype=torch.uint8)
        self.np = self.arena.numpy()
        self.state = 0
        self.out = {}

    def _f32(self, off, count):
        return self.np[off:off + 4 * count].view(np.float32)

    def begin(self, what, want_scalars):
        self.state = 0

    def next(self):
        p, S = self.plan, self.S
        r0, r1, rpr, w = p.row_begin, p.row_end, p.rows_per_rank, p.world
        self.state += 1
        if self.state == 1:
            Y = self._f32(self.off_y, p.n_pad * self.c).reshape(p.n_pad, self.c)
            Y[:] = np.nan
            Y[r0:r0 + rpr] = 0
            if r1 > r0:
                Y[r0:r1] = self.A[r0:r1] @ self.V
            return (S.XCHG_ALLGATHER, 0, self.off_y, 0, rpr * self.c * 4)
        if self.state == 2:
            Y = self._f32(self.off_y, p.n_pad * self.c).reshape(p.n_pad, self.c)
            self.out["Y"] = Y[:self.n].copy()
            self.np[self.off_s:self.off_s + 8].view(np.float64)[0] = float((Y[r0:r1].astype(np.float64) ** 2).sum())
            return (S.XCHG_ALLREDUCE_F64, 1, self.off_s, 0, 0)
        if self.state == 3:
            self.out["s"] = float(self.np[self.off_s:self.off_s + 8].view(np.float64)[0])
            send = self._f32(self.off_send, w * rpr * rpr).reshape(w, rpr, rpr)
            send[:] = 0
            if r1 > r0:
                Ccol = self.K @ self.A[r0:r1].T                        # C[:, rows]
                for s in range(w):
                    s0, s1 = s * rpr, min((s + 1) * rpr, self.n)
                    if s1 > s0:
                        send[s, :s1 - s0, :r1 - r0] = Ccol[s0:s1]      # rows of rank s, my columns
            return (S.XCHG_ALLTOALL, 0, self.off_send, self.off_recv, rpr * rpr * 4)
        if self.state == 4:
            recv = self._f32(self.off_recv, w * rpr * rpr).reshape(w, rpr, rpr)
            Crow = np.zeros((max(r1 - r0, 0), self.n), np.float32)
            for s in range(w):
                s0, s1 = s * rpr, min((s + 1) * rpr, self.n)
                if s1 > s0 and r1 > r0:
                    Crow[:, s0:s1] = recv[s, :r1 - r0, :s1 - s0]       # peer s packed its C[my rows, its columns]
            self.out["Crow"] = Crow
        return (S.XCHG_DONE, 0, 0, 0, 0)

    def scalars(self):
        return None


def main(n, steps, out):
    mcgra_loader.load()
    from mc_gra_amd.sharded import RowBlockPlan, ShardedStepper
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    plan = RowBlockPlan(n, world, rank)
    b = ToyRank(plan, n)
    st = ShardedStepper(b, plan, dist=dist, clone_input=True)
    for _ in range(steps):
        st.step()
    np.savez(f"{out}.rank{rank}.npz", Y=b.out["Y"], s=b.out["s"], Crow=b.out["Crow"],
             rows=np.array([plan.row_begin, plan.row_end, plan.n_pad]), exchanges=st.exchanges)
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
