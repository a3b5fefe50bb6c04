"""Row-block sharded attack step over ranks (one process per GPU, torch.distributed; backend "nccl" is RCCL).

Rank r owns rows [row_begin, row_end) of the learnable adjacency and of the Adam moments and does 1/world of every
N x N pass of the fused low-rank step (csrc/attack_fused.hip).  The engine runs to the next
exchange point and describes the collective (include/mcgra.h: mcgra_exchange_t); this module executes it on views of
ONE device arena the engine was bound to:

  ALLGATHER      rows of an n x c node array (skinny products M[rows, :] V; r, d, gd, decode backward) -- `world`
                 equal chunks, this rank's chunk filled
  ALLREDUCE_F64  a handful of scalars (|adj_changes|^2, mask count, loss terms)
  ALLTOALL       tile blocks of the N x N x N product: every rank computed the COLUMN block P1[:, rows] and receives
                 its ROW block P1[rows, :] (the mirrored gradient needs P1_ij and P1_ji) -- n^2 / world floats per rank
                 and step, the only N x N-sized traffic

The backend is duck-typed (arena / begin / next / scalars / plan) so that the same orchestration drives the HIP engine
(HipShardBackend) and, in the gloo CPU tests, a numpy stand-in.
"""
import torch

PANEL = 256          # row blocks are whole 256-row panels (split_symm_bf16.hip), i.e. whole 64-row tiles of the tail
XCHG_DONE, XCHG_ALLGATHER, XCHG_ALLREDUCE_F64, XCHG_ALLTOALL = 0, 1, 2, 3
SHARD_STEP, SHARD_MONITOR, SHARD_MONITOR_LAST = 0, 1, 2


class RowBlockPlan:
    """Equal row blocks of whole panels: rows_per_rank = ceil(panels / world) * 256; exchanged node arrays have
    n_pad = rows_per_rank * world rows so that all-gather chunks are equal.  Trailing ranks may own fewer rows or none
    (they still take part in every collective)."""

    def __init__(self, n, world, rank):
        self.n, self.world, self.rank = int(n), int(world), int(rank)
        panels = (self.n + PANEL - 1) // PANEL
        self.rows_per_rank = ((panels + self.world - 1) // self.world) * PANEL
        self.n_pad = self.rows_per_rank * self.world
        self.row_begin = self.rank * self.rows_per_rank
        self.row_end = min(self.row_begin + self.rows_per_rank, self.n) if self.row_begin < self.n else self.row_begin
        self.has_rows = self.row_begin < self.n


def run_exchange(ex, arena, plan, dist=None, group=None, clone_input=False, always=False, host_staged=False):
    """One collective of the protocol on the rank's arena (uint8 tensor).  dist None / world 1: nothing to move
    (always=True issues the world-1 collectives anyway: a hardware smoke test of the RCCL calls).

    host_staged=True: the arena lives on a device but the process group is a host one (gloo) -- the slices a collective
    touches are copied to host tensors, exchanged there and copied back.  For ranks that SHARE one GPU (RCCL refuses two
    ranks on one device): the multi-process tests of the engine on a 1-GPU box; never the fast path."""
    kind, count, off, off2, chunk = ex
    if kind == XCHG_DONE or dist is None or (plan.world == 1 and not always):
        return
    w, r = plan.world, plan.rank
    if host_staged:
        # (.cpu() orders itself behind the engine's launches on the current stream and waits; the copies back are enqueued
        # on the same stream, in front of the engine's next launches)
        if kind == XCHG_ALLGATHER:
            full = arena[off:off + w * chunk]
            mine = full[r * chunk:(r + 1) * chunk].cpu()
            host = torch.empty(w * chunk, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, mine, group=group)
            full.copy_(host)
        elif kind == XCHG_ALLREDUCE_F64:
            dev = arena[off:off + 8 * count].view(torch.float64)
            host = dev.cpu()
            dist.all_reduce(host, group=group)
            dev.copy_(host)
        elif kind == XCHG_ALLTOALL:
            send = arena[off:off + w * chunk].cpu()
            host = torch.empty(w * chunk, dtype=torch.uint8)
            dist.all_to_all_single(host, send, group=group)
            arena[off2:off2 + w * chunk].copy_(host)
        else:
            raise ValueError(f"exchange kind {kind}")
        return
    if kind == XCHG_ALLGATHER:
        full = arena[off:off + w * chunk]
        mine = full[r * chunk:(r + 1) * chunk]
        dist.all_gather_into_tensor(full, mine.clone() if clone_input else mine, group=group)
    elif kind == XCHG_ALLREDUCE_F64:
        dist.all_reduce(arena[off:off + 8 * count].view(torch.float64), group=group)
    elif kind == XCHG_ALLTOALL:
        send, recv = arena[off:off + w * chunk], arena[off2:off2 + w * chunk]
        dist.all_to_all_single(recv, send, group=group)
    else:
        raise ValueError(f"exchange kind {kind}")


class ShardedStepper:
    def __init__(self, backend, plan=None, dist=None, group=None, clone_input=False, always=False, host_staged=False):
        """backend: .arena, .begin(what, want_scalars), .next() -> (kind, count, offset, offset2, chunk_bytes),
        .scalars(); dist: torch.distributed or None (world 1); clone_input: gloo needs a non-aliased all_gather
        input; host_staged: device arena, host process group (see run_exchange)."""
        self.b, self.plan = backend, plan or backend.plan
        self.dist, self.group, self.clone_input, self.always = dist, group, clone_input, always
        self.host_staged = host_staged
        self.exchanges = 0
        self._timed = None          # time_exchanges(): [(kind, event before, event after)] of the collectives since

    def time_exchanges(self, on=True):
        """Bracket every collective with HIP events on the caller's stream from now on (comm_ms reads and clears them).  The
        event behind a collective is recorded once the caller's stream has been made to wait for it (torch.distributed's
        synchronous collectives), so the pair measures what the compute stream loses to it."""
        self._timed = [] if on else None

    def comm_ms(self):
        """{"allgather": ms, "alltoall": ms, "allreduce": ms, "count": n} of the collectives since time_exchanges() / the last
        call; synchronises the device."""
        out = {"allgather": 0.0, "alltoall": 0.0, "allreduce": 0.0, "count": 0}
        if not self._timed:
            return out
        torch.cuda.synchronize()
        names = {XCHG_ALLGATHER: "allgather", XCHG_ALLTOALL: "alltoall", XCHG_ALLREDUCE_F64: "allreduce"}
        for kind, e0, e1 in self._timed:
            out[names[kind]] += e0.elapsed_time(e1)
            out["count"] += 1
        self._timed = []
        return out

    def _run(self, what, want_scalars):
        self.b.begin(what, want_scalars)
        while True:
            ex = self.b.next()
            if ex[0] == XCHG_DONE:
                break
            if self._timed is not None and self.plan.world > 1 and self.dist is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run_exchange(ex, self.b.arena, self.plan, self.dist, self.group, self.clone_input, self.always, self.host_staged)
                e1.record()
                self._timed.append((ex[0], e0, e1))
            else:
                run_exchange(ex, self.b.arena, self.plan, self.dist, self.group, self.clone_input, self.always, self.host_staged)
            self.exchanges += 1
        return self.b.scalars() if want_scalars else None

    def step(self, want_scalars=False):
        """One iteration of the loop (topology_attack.py:161-283) on this rank's row block."""
        return self._run(SHARD_STEP, want_scalars)

    def monitor(self, want_sparsity=False, last=False):
        """The monitoring forward (:290-296); the next step adopts it.  Returns mean(modified_adj) if asked.
        last: no step follows (the final epoch) -- the rank's forward then does not fork the next step's product."""
        out = self._run(SHARD_MONITOR_LAST if last else SHARD_MONITOR, want_sparsity)
        return out[0] if want_sparsity else None


def run_lockstep(backends, what=SHARD_STEP, want_scalars=False):
    """All ranks of one attack inside ONE process (one GPU, or numpy stand-ins): the backends advance in lockstep and
    the collectives are plain copies between their arenas.  For tests and for per-rank timing on a single device."""
    w = len(backends)
    for b in backends:
        b.begin(what, want_scalars)
    n_ex = 0
    joint = getattr(backends[0], "joint", None)
    if joint is not None and not all(getattr(b, "joint", None) is joint for b in backends):
        joint = None
    while True:
        exs = [b.next() for b in backends]
        kinds = {e[0] for e in exs}
        assert len(kinds) == 1, f"ranks disagree on the next exchange: {exs}"
        kind, count, off, off2, chunk = exs[0]
        assert all(e == exs[0] for e in exs), f"ranks disagree on the exchange geometry: {exs}"
        if kind == XCHG_DONE:
            break
        n_ex += 1
        arenas = [b.arena for b in backends]
        if joint is not None and kind in (XCHG_ALLGATHER, XCHG_ALLTOALL):
            # arenas are rows of ONE [world, bytes] tensor (lockstep_backends): a collective is one strided copy, so the
            # emulation's own launches do not grow with world^2 (per-rank timing, scripts/shard_emulate.py)
            if kind == XCHG_ALLGATHER:
                full = joint[:, off:off + w * chunk].view(w, w, chunk)              # [holder, chunk owner, bytes]
                own = torch.diagonal(full, dim1=0, dim2=1).t().clone()         # [owner, bytes]: every rank's own chunk
                full.copy_(own.unsqueeze(0).expand(w, w, chunk))
            else:
                send = joint[:, off:off + w * chunk].view(w, w, chunk)              # [src, dst, bytes]
                joint[:, off2:off2 + w * chunk].view(w, w, chunk).copy_(send.transpose(0, 1))
            continue
        if kind == XCHG_ALLGATHER:
            for src in range(w):
                piece = arenas[src][off + src * chunk: off + (src + 1) * chunk]
                for dst in range(w):
                    if dst != src:
                        arenas[dst][off + src * chunk: off + (src + 1) * chunk].copy_(piece)
        elif kind == XCHG_ALLREDUCE_F64:
            vs = [a[off:off + 8 * count].view(torch.float64) for a in arenas]
            tot = vs[0].clone()
            for v in vs[1:]:
                tot += v
            for v in vs:
                v.copy_(tot)
        elif kind == XCHG_ALLTOALL:
            for src in range(w):
                for dst in range(w):
                    arenas[dst][off2 + src * chunk: off2 + (src + 1) * chunk].copy_(
                        arenas[src][off + dst * chunk: off + (dst + 1) * chunk])
    return [b.scalars() for b in backends] if want_scalars else n_ex


def run_echo(backend, what=SHARD_STEP, want_scalars=False):
    """ONE rank of a `world`-rank attack by itself, for TIMING only: every collective is answered with the rank's own data (an
    all-gather fills every peer's chunk with the own chunk, an all-to-all hands the send blocks back as the received ones), so
    the rank runs exactly the launches of a real step -- its share of every N x N pass, the replicated node chain -- with
    nothing exchanged.  The state it leaves is NOT the attack's.  For the per-rank compute of configurations whose `world`
    engines do not fit one GPU side by side (scripts/shard_emulate.py --echo) and for bench.py's compute-only pass."""
    plan, arena = backend.plan, backend.arena
    w, r = plan.world, plan.rank
    backend.begin(what, want_scalars)
    n_ex = 0
    while True:
        kind, count, off, off2, chunk = backend.next()
        if kind == XCHG_DONE:
            break
        n_ex += 1
        if kind == XCHG_ALLGATHER:
            full = arena[off:off + w * chunk].view(w, chunk)
            full.copy_(full[r].clone().unsqueeze(0).expand(w, chunk))
        elif kind == XCHG_ALLTOALL:
            arena[off2:off2 + w * chunk].copy_(arena[off:off + w * chunk])
    return n_ex


class HipShardBackend:
    """An AttackEngine created as a row-block rank, with its exchange arena owned by torch."""

    def __init__(self, engine, plan, arena=None, joint=None):
        """arena: a caller-owned uint8 device tensor of engine.exchange_bytes() bytes (default: allocated here); joint: the
        [world, bytes] tensor the arena is a row of (lockstep_backends)."""
        self.eng, self.plan = engine, plan
        nbytes = engine.exchange_bytes()
        self.arena = torch.zeros(nbytes, device=engine.device, dtype=torch.uint8) if arena is None else arena
        assert self.arena.numel() >= nbytes and self.arena.is_contiguous()
        self.joint = joint
        engine.bind_exchange(self.arena)

    def begin(self, what, want_scalars):
        self.eng.shard_begin(what, want_scalars)

    def next(self):
        return self.eng.shard_next()

    def scalars(self):
        return self.eng.shard_scalars()


def lockstep_backends(engines, plans):
    """HipShardBackends of all ranks of one attack inside one process with their arenas as the rows of ONE tensor, so that
    run_lockstep executes a collective as a single strided device copy."""
    nbytes = max(e.exchange_bytes() for e in engines)
    nbytes = (nbytes + 255) // 256 * 256
    joint = torch.zeros(len(engines), nbytes, device=engines[0].device, dtype=torch.uint8)
    return [HipShardBackend(e, p, arena=joint[k], joint=joint) for k, (e, p) in enumerate(zip(engines, plans))]
