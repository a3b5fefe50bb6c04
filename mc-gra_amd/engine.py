"""Thin host wrapper over the C-ABI attack engine and standalone ops.

PyTorch is used only as plumbing: it owns the caller-side device buffers and
the HIP stream.  All arithmetic happens in libmcgra_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import AttackConfig, check, lib


import functools


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_operand_device(fn):
    """Standalone ops launch on the device (and its current stream) of their first tensor argument, not on
    whatever device happens to be current."""
    @functools.wraps(fn)
    def run(first, *a, **k):
        with torch.cuda.device(first.device):
            return fn(first, *a, **k)
    return run


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _dev_f32(x, device):
    if isinstance(x, torch.Tensor):
        if x.is_sparse:
            x = x.to_dense()
        return x.detach().to(device=device, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float32)), device=device)


def _dev_i32(x, device):
    if isinstance(x, torch.Tensor):
        return x.detach().to(device=device, dtype=torch.int32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.int32)), device=device)


# ------------------------------------------------------------- standalone ops
@_on_operand_device
def sgemm(A, B, ta=False, tb=False, alpha=1.0, beta=0.0, out=None):
    """torch.mm on the fp32 MFMA kernel (mcgra_sgemm)."""
    m = A.shape[1] if ta else A.shape[0]
    k = A.shape[0] if ta else A.shape[1]
    n = B.shape[0] if tb else B.shape[1]
    assert (B.shape[1] if tb else B.shape[0]) == k
    if out is None:
        out = torch.zeros(m, n, device=A.device, dtype=torch.float32)
    check(lib.mcgra_sgemm(_stream(), int(ta), int(tb), m, n, k, float(alpha), _p(A), A.stride(0), _p(B),
                          B.stride(0), float(beta), _p(out), out.stride(0)))
    return out


@_on_operand_device
def ssyrk_lower(A, out=None):
    """C = A A^T on the lower tile storage (128 x 128 tiles on or below the diagonal); the rest of
    `out` is left untouched (mcgra_ssyrk_lower)."""
    n, k = A.shape
    if out is None:
        out = torch.zeros(n, n, device=A.device, dtype=torch.float32)
    check(lib.mcgra_ssyrk_lower(_stream(), n, k, 1.0, _p(A), A.stride(0), 0.0, _p(out), out.stride(0)))
    return out


@_on_operand_device
def ssymm_lower(S, B, beta=0.0, out=None):
    """C = S B with symmetric S read from its lower tile storage only (mcgra_ssymm_lower)."""
    n, m = S.shape[0], B.shape[1]
    if out is None:
        out = torch.zeros(n, m, device=S.device, dtype=torch.float32)
    check(lib.mcgra_ssymm_lower(_stream(), n, m, 1.0, _p(S), S.stride(0), _p(B), B.stride(0), float(beta), _p(out),
                                out.stride(0)))
    return out


@_on_operand_device
def ssymm_split_bf16(S, X, rowsub=None, out=None):
    """C = S (X - rowsub 1^T)^T through the 3-plane bf16 split kernel (mcgra_ssymm_split_bf16); S symmetric."""
    n = S.shape[0]
    if out is None:
        out = torch.empty(n, n, device=S.device, dtype=torch.float32)
    check(lib.mcgra_ssymm_split_bf16(_stream(), n, _p(S), S.stride(0), _p(X), X.stride(0), _p(rowsub), _p(out), out.stride(0)))
    return out


@_on_operand_device
def ssymm_split_f16(S, X, rowsub=None, out=None):
    """The same product through the 2-plane fp16 split kernel (mcgra_ssymm_split_f16); S symmetric."""
    n = S.shape[0]
    if out is None:
        out = torch.empty(n, n, device=S.device, dtype=torch.float32)
    check(lib.mcgra_ssymm_split_f16(_stream(), n, _p(S), S.stride(0), _p(X), X.stride(0), _p(rowsub), _p(out), out.stride(0)))
    return out


@_on_operand_device
def normalize_adj_tensor(adj):
    """utils.normalize_adj_tensor, dense branch (utils.py:211-230)."""
    out = torch.empty_like(adj)
    check(lib.mcgra_normalize_adj(_stream(), adj.shape[0], _p(adj), _p(out)))
    return out


@_on_operand_device
def get_modified_adj(adj_changes, ori_adj, n):
    out = torch.empty(n, n, device=adj_changes.device, dtype=torch.float32)
    check(lib.mcgra_get_modified_adj(_stream(), n, _p(adj_changes), _p(ori_adj), _p(out)))
    return out


@_on_operand_device
def info_entropy(prob):
    out = torch.zeros(1, device=prob.device, dtype=torch.float32)
    check(lib.mcgra_info_entropy(_stream(), prob.shape[0], _p(prob), _p(out)))
    return out[0]


@_on_operand_device
def dot_product_decode(Z):
    n, d = Z.shape
    out = torch.empty(n * (n - 1) // 2, device=Z.device, dtype=torch.float32)
    check(lib.mcgra_dot_product_decode(_stream(), n, d, _p(Z), _p(out)))
    return out


@_on_operand_device
def dot_product_decode2(Z, mode):
    """PGDAttack.dot_product_decode2 (topology_attack.py:421-467); `mode` as topology_attack._decode_mode gives it."""
    n, d = Z.shape
    out = torch.empty(n, n, device=Z.device, dtype=torch.float32)
    check(lib.mcgra_dot_product_decode2(_stream(), n, d, _p(Z), int(mode), _p(out)))
    return out


@_on_operand_device
def linear_hsic(X, Y):
    out = torch.zeros(1, device=X.device, dtype=torch.float32)
    check(lib.mcgra_linear_hsic(_stream(), X.shape[0], X.shape[1], Y.shape[1], _p(X), _p(Y), _p(out)))
    return out[0]


@_on_operand_device
def mutual_information(X, Y, want_grad=False):
    """utils.MutualInformation(sigma=0.4, num_bins=X.shape[1], normalize=True)(X, Y)[0] (utils.py:980-1049); want_grad: also
    its gradients w.r.t. X and Y (mcgra_mutual_information)."""
    m, c = X.shape
    assert Y.shape == X.shape
    X, Y = X.contiguous(), Y.contiguous()
    out = torch.zeros(1, device=X.device, dtype=torch.float32)
    gX = torch.empty_like(X) if want_grad else None
    gY = torch.empty_like(Y) if want_grad else None
    check(lib.mcgra_mutual_information(_stream(), m, c, _p(X), _p(Y), _p(out), _p(gX), _p(gY)))
    return (out[0], gX, gY) if want_grad else out[0]


@_on_operand_device
def hsic_regular(x, y, sigma):
    """hsic.hsic_regular (hsic.py:117-124) with a given sigma."""
    out = torch.zeros(1, device=x.device, dtype=torch.float32)
    check(lib.mcgra_hsic_regular(_stream(), x.shape[0], x.shape[1], y.shape[1], _p(x), _p(y), float(sigma), _p(out)))
    return out[0]


@_on_operand_device
def hsic_normalized(x, y, sigma):
    """hsic.hsic_normalized (hsic.py:127-135) with a given sigma."""
    out = torch.zeros(1, device=x.device, dtype=torch.float32)
    check(lib.mcgra_hsic_normalized(_stream(), x.shape[0], x.shape[1], y.shape[1], _p(x), _p(y), float(sigma), _p(out)))
    return out[0]


@_on_operand_device
def mse(X, Y):
    out = torch.zeros(1, device=X.device, dtype=torch.float32)
    check(lib.mcgra_mse(_stream(), X.numel(), _p(X), _p(Y), _p(out)))
    return out[0]


@_on_operand_device
def gcn_forward(X, adj, W, b, Wlin, blin, emb_nlayer=0):
    """GCN.forward (eval) and, when emb_nlayer > 0, embedding_GCN.forward."""
    n, nfeat = X.shape
    L = len(W)
    dims = (C.c_int32 * (L + 1))(*([nfeat] + [w.shape[1] for w in W]))
    Wp = (C.c_void_p * L)(*[w.data_ptr() for w in W])
    bp = (C.c_void_p * L)(*[x.data_ptr() for x in b])
    nclass = Wlin.shape[0]
    out = torch.empty(n, nclass, device=X.device, dtype=torch.float32)
    emb = torch.empty(n, W[emb_nlayer - 1].shape[1], device=X.device, dtype=torch.float32) if emb_nlayer else None
    check(lib.mcgra_gcn_forward(_stream(), n, nfeat, L, dims, _p(X), _p(adj), Wp, bp, _p(Wlin), _p(blin), nclass,
                                emb_nlayer, _p(emb), _p(out)))
    return out, emb


# ---------------------------------------------------------------- the engine
class AttackEngine:
    """One mcgra_attack_t.  Mirrors the state PGDAttack keeps across the loop of
    topology_attack.py:161-298 (adj_changes + Adam moments) in HBM."""

    def __init__(self, n, dims, nclass, emb_nlayer, measure, weight_sup, weight_param, lr, num_edges,
                 n_attack, eps=0.0, device="cuda:0", act="relu", head_act="none", has_self=False, fin_layers=(1, 2),
                 plan=None):
        _lib.require_device()
        self.device = torch.device(device)
        self.n, self.nclass, self.dims = int(n), int(nclass), list(int(d) for d in dims)
        cfg = AttackConfig()
        cfg.n, cfg.nfeat, cfg.nclass = self.n, self.dims[0], self.nclass
        cfg.nlayer, cfg.emb_nlayer = len(self.dims) - 1, int(emb_nlayer)
        for i, d in enumerate(self.dims):
            cfg.dims[i] = d
        if measure not in _lib.MEASURES:
            raise ValueError(f"measure {measure!r}: topology_attack.py:194-208 knows {sorted(_lib.MEASURES)}")
        cfg.measure = _lib.MEASURES[measure]
        cfg.n_attack = int(n_attack)
        cfg.weight_sup = float(weight_sup)
        for i in range(10):
            cfg.w[i] = float(weight_param[i])
        cfg.lr, cfg.eps = float(lr), float(eps)
        cfg.num_edges = float(min(num_edges, 1e300))
        # plan: a sharded.RowBlockPlan makes this engine one of plan.world row-block ranks (include/mcgra.h)
        cfg.row_begin, cfg.row_end = (0, self.n) if plan is None else (int(plan.row_begin), int(plan.row_end))
        cfg.shard_world, cfg.shard_rows = (0, 0) if plan is None else (int(plan.world), int(plan.rows_per_rank))
        cfg.act = {"relu": 0, "elu": 1}[act]
        cfg.head_act = {"none": 0, "elu": 1}[head_act]
        cfg.has_self = int(bool(has_self))
        cfg.fin_layers[0], cfg.fin_layers[1] = int(fin_layers[0]), int(fin_layers[1])
        self._h = C.c_void_p(0)
        with torch.cuda.device(self.device):
            check(lib.mcgra_attack_create(C.byref(self._h), C.byref(cfg)))
        self.cfg = cfg
        self._keep = []

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib.mcgra_attack_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_model(self, W, b, Wlin, blin, Ws=None):
        dev = self.device
        Ws = [_dev_f32(w, dev) for w in Ws] if Ws is not None else None
        W = [_dev_f32(w, dev) for w in W]
        b = [_dev_f32(x, dev) for x in b]
        Wlin, blin = _dev_f32(Wlin, dev), _dev_f32(blin, dev)
        L = len(W)
        Wp = (C.c_void_p * L)(*[w.data_ptr() for w in W])
        bp = (C.c_void_p * L)(*[x.data_ptr() for x in b])
        Wsp = (C.c_void_p * L)(*[w.data_ptr() for w in Ws]) if Ws is not None else None
        with torch.cuda.device(dev):
            check(lib.mcgra_attack_set_model(self._h, _stream(), Wp, bp, _p(Wlin), _p(blin), Wsp))
            torch.cuda.current_stream().synchronize()

    def set_graph(self, features, adj, ori_adj, feature_adj, labels, idx_attack):
        dev = self.device
        X = _dev_f32(features, dev)
        A = _dev_f32(adj, dev)
        F = _dev_f32(feature_adj, dev)
        O = None
        if ori_adj is not None:
            O = _dev_f32(ori_adj, dev)
            if not bool((O != 0).any()):
                O = None          # dataset.init_matrix (dataset.py:433): all zeros
        lab = _dev_i32(labels, dev)
        idx = _dev_i32(idx_attack, dev)
        with torch.cuda.device(dev):
            check(lib.mcgra_attack_set_graph(self._h, _stream(), _p(X), _p(A), _p(O), _p(F), _p(lab), _p(idx)))

    def set_adj_changes(self, packed):
        t = _dev_f32(packed, self.device)
        with torch.cuda.device(self.device):
            check(lib.mcgra_attack_set_adj_changes(self._h, _stream(), _p(t)))
            torch.cuda.current_stream().synchronize()

    def get_adj_changes(self):
        n = self.n
        out = torch.empty(n * (n - 1) // 2, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            check(lib.mcgra_attack_get_adj_changes(self._h, _stream(), _p(out)))
        return out

    def step(self, want_scalars=False, noise=None):
        with torch.cuda.device(self.device):
            if want_scalars:
                buf = (C.c_double * 10)()
                check(lib.mcgra_attack_step(self._h, _stream(), _p(noise), buf))
                keys = ("loss", "origin_loss", "c1", "c2", "c6", "c7", "c9", "c10", "clamp_sum", "nll")
                return dict(zip(keys, list(buf)))
            check(lib.mcgra_attack_step(self._h, _stream(), _p(noise), None))
        return None

    # ---- row-block sharded step (mcgra_attack_shard_*, driven by mc-gra_amd/sharded.py) ----------------------------
    def exchange_bytes(self):
        return int(lib.mcgra_attack_exchange_bytes(self._h))

    def bind_exchange(self, arena):
        assert arena.dtype == torch.uint8 and arena.is_contiguous() and arena.device == self.device
        self._keep.append(arena)
        with torch.cuda.device(self.device):
            check(lib.mcgra_attack_bind_exchange(self._h, _p(arena), arena.numel()))

    def shard_begin(self, what, want_scalars=False):
        with torch.cuda.device(self.device):
            check(lib.mcgra_attack_shard_begin(self._h, _stream(), int(what), int(bool(want_scalars))))

    def shard_next(self):
        ex = _lib.Exchange()
        with torch.cuda.device(self.device):
            check(lib.mcgra_attack_shard_next(self._h, _stream(), C.byref(ex)))
        return (ex.kind, ex.count, ex.offset, ex.offset2, ex.chunk_bytes)

    def shard_scalars(self):
        buf = (C.c_double * 10)()
        with torch.cuda.device(self.device):
            check(lib.mcgra_attack_shard_scalars(self._h, _stream(), buf))
        keys = ("loss", "origin_loss", "c1", "c2", "c6", "c7", "c9", "c10", "clamp_sum", "nll")
        return dict(zip(keys, list(buf)))

    def get_rows(self):
        """Rows [row_begin, row_end) of the learnable adjacency (dense form of adj_changes)."""
        r0, r1 = self.cfg.row_begin, (self.cfg.row_end or self.n)
        out = torch.empty(max(r1 - r0, 0), self.n, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            check(lib.mcgra_attack_get_rows(self._h, _stream(), _p(out)))
        return out

    def product_mode(self):
        """0: fp32 MFMA SYMM, 1: bf16 split through hipBLASLt, 2: bf16 split, hand-written kernel (mcgra_attack_product_mode)."""
        return int(lib.mcgra_attack_product_mode(self._h))

    def path_stats(self):
        a, b = C.c_longlong(0), C.c_longlong(0)
        check(lib.mcgra_attack_path_stats(self._h, C.byref(a), C.byref(b)))
        return {"lowrank_steps": a.value, "general_steps": b.value}

    def fused_steps(self):
        """Low-rank steps that ran as the fused step (mcgra_attack_fused_steps)."""
        return int(lib.mcgra_attack_fused_steps(self._h))

    def gram_split_steps(self):
        """Gram-evaluation steps whose products ran on the fp16 split kernel (mcgra_attack_gram_split_steps)."""
        return int(lib.mcgra_attack_gram_split_steps(self._h))

    def leading_dim(self):
        ptr, r, c, ld = C.c_void_p(0), C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib.mcgra_attack_buffer(self._h, b"M", C.byref(ptr), C.byref(r), C.byref(c), C.byref(ld)))
        return ld.value

    def monitor(self, want_sparsity=False):
        out = torch.empty(self.n, self.nclass, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            if want_sparsity:
                s = C.c_double(0)
                check(lib.mcgra_attack_monitor(self._h, _stream(), _p(out), C.byref(s)))
                return out, s.value
            check(lib.mcgra_attack_monitor(self._h, _stream(), _p(out), None))
        return out, None

    def finalize(self, decode_mode, H_A=None, Y_A=None, label_adj=None):
        dev = self.device
        H_A = _dev_f32(H_A, dev) if H_A is not None else None
        Y_A = _dev_f32(Y_A, dev) if Y_A is not None else None
        label_adj = _dev_f32(label_adj, dev) if label_adj is not None else None
        out = torch.empty(self.n, self.n, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            check(lib.mcgra_attack_finalize(self._h, _stream(), int(decode_mode), _p(H_A), _p(Y_A), _p(label_adj),
                                            _p(out)))
            torch.cuda.current_stream().synchronize()
        return out

    def buffer(self, name):
        """Copy of a named intermediate (parity tests)."""
        ptr, r, c, ld = C.c_void_p(0), C.c_int(0), C.c_int(0), C.c_int(0)
        check(lib.mcgra_attack_buffer(self._h, name.encode(), C.byref(ptr), C.byref(r), C.byref(c), C.byref(ld)))
        out = torch.empty(r.value, c.value, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            check(lib.mcgra_attack_copy_buffer(self._h, _stream(), name.encode(), _p(out), c.value))
            torch.cuda.current_stream().synchronize()
        return out

    def masked_fused_steps(self):
        """Fused steps whose decode relu-masked pairs of live embedding rows (they stand; only a dead row falls back)."""
        return int(lib.mcgra_attack_masked_fused_steps(self._h))

    def cut_product_steps(self):
        """Steps whose product was cut in two: a row-block rank's so that the P1 all-to-all runs beside its own row panels, a
        large monolithic graph's so that the tail's first pass runs beside the product's last rounds."""
        return int(lib.mcgra_attack_cut_product_steps(self._h))

    def product_replay(self, reps=10):
        """Mean launch time [ms] of `reps` back-to-back launches of the last fused step's N x N x N product, nothing beside
        them (mcgra_attack_product_replay: a measurement aid, no engine state changes)."""
        ms = C.c_double(0)
        with torch.cuda.device(self.device):
            check(lib.mcgra_attack_product_replay(self._h, _stream(), int(reps), C.byref(ms)))
        return ms.value

    def test_mutate(self, what):
        """TEST ONLY (mcgra_attack_test_mutate): 'p1' wipes the product's result, 'rk' drops the tail's rank-k terms, None disarms."""
        check(lib.mcgra_attack_test_mutate(self._h, {None: 0, "p1": 1, "rk": 2}[what]))

    def profile(self, enable=True):
        check(lib.mcgra_attack_profile(self._h, int(enable)))

    def gemm_stats(self, reset=True):
        n, ms, fl = C.c_int64(0), C.c_double(0), C.c_double(0)
        check(lib.mcgra_attack_gemm_stats(self._h, int(reset), C.byref(n), C.byref(ms), C.byref(fl)))
        return dict(launches=n.value, ms=ms.value, flops=fl.value)
