"""World-size-N gloo worker: the product's ShardedStepper driving an oracle-backed stand-in for the HIP engine
(same phase protocol, same exchanged buffers).  Writes the final adjacency of each rank to out.npz."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import torch.distributed as dist

import helpers
import mcgra_loader


class OracleShardBackend:
    """phase(k) protocol of HipShardBackend on top of PGDAttackOracle.step_iter (phases 0+1 run to its first yield)."""
    needs_exchange = True

    def __init__(self, orc, plan):
        self.orc, self.plan = orc, plan
        self.np_ex = {k: np.full((plan.n_pad, orc.n), np.nan, np.float32) for k in ("KX", "KY", "G_adjn", "G_A1")}
        self.exchanged = {k: torch.from_numpy(v) for k, v in self.np_ex.items()}     # shared memory
        orc.shard, orc.exchanged = (plan.row_begin, plan.row_end), self.np_ex
        self.it = None

    def exchange_names(self, k):
        return {1: ("KX", "KY"), 2: ("G_adjn", "G_A1")}.get(k, ())

    def phase(self, k, noise=None, want_scalars=False):
        if k == 0:
            for v in self.np_ex.values():
                v[:] = np.nan                          # stale rows from the previous step must not be reused
            self.it = self.orc.step_iter(noise)
            assert next(self.it) == "gram"
        elif k == 2:
            assert next(self.it) == "grad"
        elif k == 3:
            try:
                next(self.it)
            except StopIteration as e:
                return e.value
            raise AssertionError("step_iter yielded more than twice")
        return None


def main(case, steps, out):
    pkg = mcgra_loader.load()
    from mc_gra_amd.sharded import RowBlockPlan, ShardedStepper
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    z = helpers.load_case(case)
    orc = helpers.oracle_from(z)
    a0 = helpers.a0_of(z)
    if a0 is not None:
        orc.set_adj_changes(a0)
    plan = RowBlockPlan(orc.n, world, rank)
    st = ShardedStepper(OracleShardBackend(orc, plan), plan, dist=dist, clone_input=True)
    losses = []
    for t in range(steps):
        r = st.step(noise=helpers.noise_of(z, t))
        losses.append(r["loss"])
    np.savez(f"{out}.rank{rank}.npz", M=orc.M, losses=np.array(losses), rows=np.array([plan.row_begin, plan.row_end, plan.n_pad]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3])
