"""Row-block sharded attack step over ranks (one process per GPU, torch.distributed; backend "nccl" is RCCL).

Only the N x N x N products of linear_HSIC / linear_CKA are sharded (they are ~90 % of a step at N = 10 000):
rank r computes the tile rows [row_begin, row_end) of the two centred Grams and of the two gradient products;
one all-gather of row blocks follows each.  Everything else is O(n^2) and stays replicated, so every rank holds
the full learnable adjacency and no parameter exchange is needed.  The backend is duck-typed (``phase(k, noise)``
plus the four exchanged tensors) so that the same orchestration runs on the HIP engine and, in the gloo CPU
tests, on a numpy stand-in supplied by the test.
"""
import torch

TILE = 256          # row blocks are whole 256-row panels (2 x SYM_TILE of csrc/common.h; panel of split_symm_bf16.hip)
EXCHANGED_AFTER_PHASE = {1: ("KX", "KY"), 2: ("G_adjn", "G_A1")}
EXCHANGE_BIT = {"KX": 1, "KY": 2, "G_adjn": 4, "G_A1": 8}          # MCGRA_EXCHANGE_* of include/mcgra.h


class RowBlockPlan:
    """Equal row blocks of whole tiles: rows_per_rank = ceil(n / (TILE * world)) * TILE; the exchanged buffers
    have n_pad = rows_per_rank * world rows so that all_gather chunks are equal."""

    def __init__(self, n, world, rank):
        self.n, self.world, self.rank = int(n), int(world), int(rank)
        tiles = (self.n + TILE - 1) // TILE
        self.rows_per_rank = ((tiles + self.world - 1) // self.world) * TILE
        self.n_pad = self.rows_per_rank * self.world
        self.row_begin = min(self.rank * self.rows_per_rank, self.n_pad)
        self.row_end = self.row_begin + self.rows_per_rank
        # a trailing rank may own no real rows (n small against world * TILE): it still takes part in collectives
        self.has_rows = self.row_begin < self.n


class ShardedStepper:
    def __init__(self, backend, plan, dist=None, clone_input=False):
        """backend: .phase(k, noise) and .exchanged[name] -> tensor [n_pad, ld]; dist: torch.distributed or None
        (world 1); clone_input: gloo needs a non-aliased all_gather input."""
        self.b, self.plan, self.dist, self.clone_input = backend, plan, dist, clone_input

    def _all_gather_rows(self, t):
        if self.dist is None or self.plan.world == 1:
            return
        p = self.plan
        mine = t[p.rank * p.rows_per_rank:(p.rank + 1) * p.rows_per_rank]
        if self.clone_input:
            mine = mine.clone()
        self.dist.all_gather_into_tensor(t.view(-1), mine.reshape(-1))

    def step(self, noise=None, want_scalars=False):
        out = None
        for k in range(4):
            r = self.b.phase(k, noise) if k < 3 else self.b.phase(k, noise, want_scalars)
            if k == 3:
                out = r
            if self.b.needs_exchange:
                for name in self.b.exchange_names(k):
                    self._all_gather_rows(self.b.exchanged[name])
        return out


class HipShardBackend:
    """AttackEngine restricted to this rank's row block, with the exchanged buffers owned by torch."""

    def __init__(self, engine, plan):
        self.eng, self.plan = engine, plan
        ld = engine.leading_dim()
        dev = engine.device
        self.needs_exchange = engine.cfg.measure in (0, 3)          # HSIC, CKA: the N x N x N products
        self.exchanged = {}
        if self.needs_exchange:
            for name in ("KX", "KY", "G_adjn", "G_A1"):
                t = torch.zeros(plan.n_pad, ld, device=dev, dtype=torch.float32)
                engine.bind_buffer(name, t)
                self.exchanged[name] = t

    def phase(self, k, noise=None, want_scalars=False):
        return self.eng.step_phase(k, noise=noise, want_scalars=want_scalars)

    def exchange_names(self, k):
        """Buffers to gather after phase k of the step in flight: the engine picks, per step, between the
        low-rank evaluation (one product: only KX rows travel) and the Gram evaluation (all four)."""
        mask = self.eng.exchange_mask()
        return [nm for nm in EXCHANGED_AFTER_PHASE.get(k, ()) if mask & EXCHANGE_BIT[nm]]
