"""PGDAttack on the MI355X hot path.

Same constructor and ``attack`` signature as the reference class
(/root/reference/MC-GRA/topology_attack.py:55-324), so ``main.py`` can import
it in place of the reference's.  The loop body (:161-298), its backward and the
post-loop ensemble (:300-324) run as hand-written HIP kernels through the
C ABI of include/mcgra.h; this file only validates arguments, moves the inputs
to HBM and drives the engine.  There is no CPU path: without a HIP device or
without libmcgra_hip.so it raises.

Arguments the reference accepts but this path does not cover raise
NotImplementedError naming the reference line (an embedding whose weights differ
from victim_model.gc).

Several GPUs: when ``torch.distributed`` is initialised with more than one rank (``torchrun ... main.py``: one process
per GPU, backend "nccl" = RCCL), ``attack`` runs ONE attack row-block sharded over the ranks (mc-gra_amd/sharded.py,
DESIGN.md section 6) whenever the configuration is one a fused step covers (measure HSIC -- n >= 1024, w1 or w2 non-zero -- or
MSELoss -- n >= 256 --, ReLU GCN victim, eps == 0, ori_adj == 0, hidden widths <= 32, a projection budget that cannot bind); every
rank returns the same ``modified_adj``.  Any other configuration says on stderr that it runs replicated.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

from .base_attack import BaseAttack
from ._lib import McgraNotSupported
from .engine import AttackEngine

# dot_product_decode2 branch -> mcgra_attack_finalize decode_mode (topology_attack.py:421-467)
def _decode_mode(args):
    ds = args.dataset
    if ds in ('cora', 'AIDS'):
        return 0
    if ds == 'citeseer':
        return 1
    if ds == 'brazil':
        return 2
    if ds in ('polblogs', 'usair'):
        H, YA, Y = bool(args.useH_A), bool(args.useY_A), bool(args.useY)
        if ds == 'polblogs' and H and YA and Y:
            return 3
        if ds == 'usair' and Y and not H and not YA:
            return 5            # F.normalize(p=3) then Z Z^T
        if ds == 'usair' and not Y and H and YA:
            return 4            # p=2
        if ds == 'usair' and Y and H and not YA:
            return 6            # p=5
        return 3
    raise ValueError(f"dot_product_decode2 has no branch for dataset {ds!r} (topology_attack.py:421-467)")


def _dist_group():
    """(torch.distributed, world, rank, host_staged) of an initialised default process group with more than one rank, else
    (None, 1, 0, False).  host_staged: the group is a host one (gloo) -- ranks that SHARE a GPU (RCCL refuses two ranks on one
    device): the exchanged arena slices are copied through the host (sharded.run_exchange), a test mode, never the fast path."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return None, 1, 0, False
    return dist, dist.get_world_size(), dist.get_rank(), str(dist.get_backend()).lower() != "nccl"


def _bcast(dist, t, host_staged):
    """rank 0's copy of a float / int tensor on every rank (in place on a contiguous tensor; returns it)."""
    if host_staged:
        c = t.detach().cpu().contiguous()
        dist.broadcast(c, 0)
        t.copy_(c)
    else:
        dist.broadcast(t, 0)
    return t


def _dense_np(x, dtype=np.float32):
    if sp.issparse(x):
        return np.asarray(x.todense(), dtype=dtype)
    if isinstance(x, torch.Tensor):
        if x.is_sparse:
            x = x.to_dense()
        return x.detach().cpu().numpy().astype(dtype, copy=False)
    return np.asarray(x, dtype=dtype)


class PGDAttack(BaseAttack):
    """topology_attack.PGDAttack (topology_attack.py:55-81)."""

    def __init__(self, model=None, embedding=None, H_A=None, Y_A=None, nnodes=None, loss_type='CE',
                 feature_shape=None, attack_structure=True, attack_features=False, device='cpu'):
        super(PGDAttack, self).__init__(model, nnodes, attack_structure, attack_features, device)
        assert attack_features or attack_structure, 'attack_features or attack_structure cannot be both False'
        self.loss_type = loss_type
        self.modified_adj = None
        self.modified_features = None
        self.edge_select = None
        self.complementary = None
        self.complementary_after = None
        self.embedding = embedding
        self.H_A = H_A
        self.Y_A = Y_A
        self.engine = None
        self.history = {}
        if attack_structure:
            assert nnodes is not None, 'Please give nnodes='
            # the reference keeps Parameter(zeros(n(n-1)/2)) here (:77-78); on this path the
            # learnable adjacency lives in HBM inside the engine and is exposed by the property below
            self._adj_changes_init = None
        if attack_features:
            assert True, 'Topology Attack does not support attack feature'

    # ---- adj_changes in the reference's packed order --------------------------
    @property
    def adj_changes(self):
        if self.engine is not None:
            return self.engine.get_adj_changes()
        n = self.nnodes
        if self._adj_changes_init is None:
            return torch.zeros(int(n * (n - 1) / 2))
        return self._adj_changes_init

    @adj_changes.setter
    def adj_changes(self, value):
        if self.engine is not None:
            self.engine.set_adj_changes(value)
        else:
            self._adj_changes_init = torch.as_tensor(value, dtype=torch.float32)

    # ---- helpers ---------------------------------------------------------------
    @staticmethod
    def _weights(victim_model, embedding):
        """Victim weights in the engine's unified layer form  P_l = adj @ (H W_l) + H Ws_l + b_l, H_l = act(P_l).
        Returns (W, b, Wlin, blin, Ws, act, head_act, emb_full): models/gcn.py GCN, the dense GAT of
        models/gat.py (its attention product is overwritten by adj @ h, gat.py:44-45) and models/graphsage.py."""
        if hasattr(victim_model, "attentions"):                       # GAT: heads concatenated, no bias, ELU
            W = [torch.cat([a.W.detach() for a in heads], dim=1) for heads in victim_model.attentions]
            b = [torch.zeros(w.shape[1]) for w in W]
            Wlin, blin = victim_model.out_att.weight.detach(), victim_model.out_att.bias.detach()
            emb_att = getattr(embedding, "attentions", None)
            if emb_att is not None and emb_att is not victim_model.attentions:
                for he, hv in zip(emb_att, victim_model.attentions):
                    for ae, av in zip(he, hv):
                        if not torch.equal(ae.W.detach().cpu(), av.W.detach().cpu()):
                            raise NotImplementedError("embedding.attentions differs from victim_model.attentions "
                                                      "(main.py:231 shares them)")
            return W, b, Wlin, blin, None, "elu", "elu", True
        gcs = victim_model.gc
        sage = gcs[0].weight.shape[0] == 2 * victim_model.nfeat       # graphsage.py:20: weight is [2*in, out]
        W, Ws, b = [], ([] if sage else None), []
        for l in gcs:
            w = l.weight.detach()
            if sage:
                half = w.shape[0] // 2
                Ws.append(w[:half].contiguous())
                w = w[half:].contiguous()
            W.append(w)
            b.append(torch.zeros(w.shape[1]) if l.bias is None else l.bias.detach())
        Wlin = victim_model.linear1.weight.detach()
        blin = (victim_model.linear1.bias.detach() if victim_model.linear1.bias is not None
                else torch.zeros(Wlin.shape[0]))
        emb_gc = getattr(embedding, "gc", None)
        if emb_gc is not None:
            for le, lv in zip(emb_gc, gcs):
                same = torch.equal(le.weight.detach().cpu(), lv.weight.detach().cpu())
                if le.bias is not None and lv.bias is not None:
                    same = same and torch.equal(le.bias.detach().cpu(), lv.bias.detach().cpu())
                if not same:
                    raise NotImplementedError(
                        "embedding.gc differs from victim_model.gc; main.py:190 deep-copies them and the HIP path "
                        "shares one GCN chain between embedding(features, modified_adj) and victim(features, "
                        "modified_adj)")
        return W, b, Wlin, blin, Ws, "relu", "none", False

    @staticmethod
    def _replicated_reason(measure, eps, ori_np, Ws, act, head_act, loss_type, n, dims, w1, w2, num_edges, emb_nlayer=None):
        """None when the row-block sharded fused step covers this configuration (include/mcgra.h: mcgra_attack_shard_*),
        else why it does not.  Mirrors the create-time rule of csrc/attack.hip (`fused_ok` / `fused_mse` / `fused_kl` under
        shard_world > 0) term by term: a configuration this accepts and mcgra_attack_create refuses would fail on every rank
        instead of running replicated (attack() also catches that refusal, should the two ever drift apart)."""
        if measure not in ("HSIC", "MSELoss", "KL"):
            return f"measure {measure} (the fused HSIC, MSELoss and KL steps are the sharded ones)"
        if loss_type != "CE":
            return "loss_type 'CW' takes no step"
        if eps != 0:
            return "eps != 0 (adding_noise makes modified_adj asymmetric: general step)"
        if ori_np is not None:
            return "a non-zero ori_adj (general step)"
        if Ws is not None or act != "relu" or head_act != "none":
            return "a GAT / GraphSAGE victim (Gram evaluation of linear_HSIC)"
        split = os.environ.get("MCGRA_SPLIT_BF16", "")
        if measure == "HSIC" and split == "0":
            return "MCGRA_SPLIT_BF16=0 (the product runs on the fp32 kernel: nothing to shard)"
        if measure == "HSIC" and n < 1024 and split not in ("2", "3"):
            return f"n = {n} < 1024 (the product runs on the fp32 kernel: nothing to shard)"
        if n < 256:
            return f"n = {n} < 256"
        widths = [int(w) for w in dims[1:]]
        le = min(2, len(widths)) if emb_nlayer is None else int(emb_nlayer)
        he = widths[le - 1]
        if max(widths) > 32:
            return f"hidden width {max(widths)} > 32"
        if he not in (8, 16, 32):                                  # lr_decode_supported: the per-pair decode's register tiles
            return f"embedding width {he} (the per-pair decode is built for widths 8, 16 and 32)"
        hsum = sum((w + 3) & ~3 for w in widths)                   # the concatenated node buffers: rank-k depth of the tail
        if max(hsum, 2 * he) > 64:                                 # fl_tail_supported: kmax <= 64
            return (f"summed layer widths {hsum} / twice the embedding width {2 * he} > 64 (rank-k depth of the tail's "
                    f"panels: e.g. more than four 16-wide layers)")
        if measure == "HSIC":
            fc = max([2 * he + 1 + widths[-1]] + [2 * w + 1 for w in widths])
        else:
            fc = max(2 * w for w in widths)
        if ((fc + 3) & ~3) > 64:
            return f"skinny products of {fc} columns > 64"
        if measure == "HSIC" and w1 == 0 and w2 == 0:
            return "w1 == w2 == 0 (no N x N HSIC term)"
        if num_edges < 0.5 * float(n) * float(n):
            return "a projection budget that can bind (host-driven bisection)"
        return None

    def test(self, idx_attack, idx_val, idx_test, adj, features, labels, victim_model):
        """topology_attack.py:83-93 through mcgra_gcn_forward / mcgra_normalize_adj."""
        from . import engine as E
        dev = torch.device(self.device)
        adj_t = torch.as_tensor(_dense_np(adj), device=dev)
        X = torch.as_tensor(_dense_np(features), device=dev)
        W, b, Wlin, blin, Ws, act, head_act, _ = self._weights(victim_model, None)
        if Ws is not None or act != "relu":
            raise NotImplementedError("PGDAttack.test() through mcgra_gcn_forward covers the GCN victim only")
        to = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        out, _ = E.gcn_forward(X, E.normalize_adj_tensor(adj_t), [to(w) for w in W], [to(x) for x in b], to(Wlin), to(blin))
        lab = torch.as_tensor(np.asarray(labels), device=dev)
        pred = out[torch.as_tensor(np.asarray(idx_test), device=dev)].max(1)[1]
        return (pred == lab[torch.as_tensor(np.asarray(idx_test), device=dev)]).double().mean().item()

    def attack(self, args, index_delete, lr_ori, weight_aux, weight_supervised, weight_param, feature_adj,
               aux_adj, aux_feature, aux_num_edges, idx_train, idx_val, idx_test, adj,
               ori_features, ori_adj, labels, idx_attack, num_edges,
               dropout_rate, epochs=200, sample=False, **kwargs):
        """Same parameters as the reference (topology_attack.py:95-116).

        Extra keyword arguments (ignored by the reference through **kwargs):
          label_adj : array, used instead of np.load('./saved_data/<dataset>.npy') (:133)
          monitor   : bool, run the per-step test-accuracy forward of :290-296 (default True)
        """
        dev = torch.device(self.device)
        if dev.type != 'cuda':
            raise RuntimeError("mc-gra_amd PGDAttack runs on an MI355X ('cuda:N' device) only; there is no CPU path")
        if self.loss_type not in ('CE', 'CW'):
            raise NotImplementedError(f"loss_type {self.loss_type!r} (topology_attack.py:326-335 knows 'CE' and 'CW')")
        # loss_type 'CW': the reference computes the Carlini-Wagner margin loss and back-propagates it, but only calls
        # optimizer.step() for 'CE' (:277-280) -- adj_changes never moves, and the run's result is the post-loop
        # ensemble of the untouched adjacency.  The same happens here: no step is taken.  (With a non-zero starting
        # adj_changes the reference's per-iteration projection (:282) could still move it; that start is a test hook.)
        if self.loss_type == 'CW' and self._adj_changes_init is not None:
            raise NotImplementedError("loss_type 'CW' with a non-zero starting adj_changes (projection without steps, :282)")
        if args.max_eval == 1:                      # (:118-119)
            lr_ori = 10 ** args.lr
        self.args = args
        victim_model = self.surrogate
        w1, w2, _, _, _, w6, w7, w8, w9, w10 = weight_param     # (:151)
        if args.max_eval == 1:                      # (:152-159), including the w7 <- args.w8 quirk
            w1, w2, w6 = args.w1, args.w2, args.w6
            w7 = args.w7
            w7 = args.w8
            w9, w10 = args.w9, args.w10
        measure = args.measure
        if measure not in ("HSIC", "MSELoss", "KL", "DP", "CKA", "KDE"):
            raise ValueError(f"measure {measure!r}: topology_attack.py:194-208 knows HSIC, MSELoss, KL, KDE, CKA, DP")
        eps = float(getattr(args, "eps", 0) or 0)

        n = self.nnodes
        adj_np = _dense_np(adj)
        # ori_adj: zeros from main.py (dataset.init_matrix, dataset.py:433-437); anything else takes the engine's general
        # path (modified_adj = clamp(adj_changes + ori_adj), embedding on modified_adj - ori_adj: :164-165, :185)
        ori_np = _dense_np(ori_adj)
        ori_np = ori_np if np.any(ori_np != 0) else None
        fadj = _dense_np(feature_adj)
        if w1 != 0 and not (fadj.max() != fadj.min()):            # (:212)
            w1 = 0
        lab = np.asarray(labels.cpu() if isinstance(labels, torch.Tensor) else labels).astype(np.int64)
        idx = np.asarray(idx_attack).astype(np.int64)
        W, b, Wlin, blin, Ws, act, head_act, emb_full = self._weights(victim_model, self.embedding)
        dims = [W[0].shape[0]] + [w.shape[1] for w in W]
        # GCN / GraphSAGE embeddings: whatever embedding.nlayer holds at call time, the reference loop resets it with
        # set_layers(2) (:179) before `em = embedding(...)` (:185), the w9 loop (:238-240) and the post-loop decode
        # (:300) use it, so every embedding inside the loop is the 2-layer one.  embedding_gat.forward ignores
        # set_layers (gat.py:170-174): full depth.
        emb_nlayer = len(W) if emb_full else min(2, len(W))
        fin_layers = (len(W), len(W)) if emb_full else (1, min(2, len(W)))
        if measure == "KDE" and max(dims[emb_nlayer], int(Wlin.shape[0])) > 32:
            # utils.MutualInformation(num_bins = H_A1.shape[1]) on the embeddings (:244-249) / num_bins = classes (:261-265): the
            # c x c joint pdf of kde_kernels.hip is built for c <= 32 -- said here, before any device memory is allocated
            raise NotImplementedError(
                f"measure 'KDE' with an embedding of width {dims[emb_nlayer]} / {int(Wlin.shape[0])} classes: the c x c joint pdf of "
                f"utils.MutualInformation (topology_attack.py:244-249, :261-265; utils.py:980-1049) is built for widths <= 32 on this "
                f"path (a GAT victim's 5 x 16 = 80-wide embedding is not covered)")
        label_adj = kwargs.get("label_adj", None)
        if label_adj is None and getattr(args, "useY", False):
            label_adj = np.load("./saved_data/" + args.dataset + ".npy")   # (:133)
        monitor = bool(kwargs.get("monitor", True))

        if self.engine is not None:
            self.engine.close()
        # ---- several ranks (torchrun main.py: main.py:298-307 is called by every rank): one attack, row-block sharded
        dist, world, rank, host_staged = _dist_group()
        plan = stepper = None
        if world > 1:
            why = self._replicated_reason(measure, eps, ori_np, Ws, act, head_act, self.loss_type, n, dims, w1, w2, num_edges,
                                          emb_nlayer)
            if why is None:
                from .sharded import HipShardBackend, RowBlockPlan, ShardedStepper
                plan = RowBlockPlan(n, world, rank)
            elif rank == 0:
                print(f"[mc-gra_amd] PGDAttack.attack under {world} ranks runs REPLICATED (every rank the whole attack): {why}",
                      file=sys.stderr, flush=True)
        to_dev = lambda t: torch.as_tensor(np.asarray(t.detach().cpu() if isinstance(t, torch.Tensor) else t),
                                           dtype=torch.float32).to(dev).contiguous()
        if plan is not None:
            # every rank must hold rank 0's victim bit for bit (the node-level part of the step is replicated and the ranks
            # must agree on it): a victim trained per rank on the same seed is not guaranteed to be
            W = [_bcast(dist, to_dev(w), host_staged) for w in W]
            b = [_bcast(dist, to_dev(x), host_staged) for x in b]
            Wlin, blin = _bcast(dist, to_dev(Wlin), host_staged), _bcast(dist, to_dev(blin), host_staged)
            idx_t = _bcast(dist, torch.as_tensor(idx, device=dev), host_staged)
            idx = idx_t.cpu().numpy()
        mk = lambda pl: AttackEngine(n, dims, int(Wlin.shape[0]), emb_nlayer, measure, weight_supervised,
                                     (w1, w2, 0, 0, 0, w6, w7, w8, w9, w10), lr_ori, num_edges, len(idx), eps=eps, device=dev,
                                     act=act, head_act=head_act, has_self=Ws is not None, fin_layers=fin_layers, plan=pl)
        try:
            eng = mk(plan)
        except McgraNotSupported as e:
            # the create-time rule decides the same way on every rank (configuration only): all of them fall back together
            if plan is None:
                raise
            if rank == 0:
                print(f"[mc-gra_amd] PGDAttack.attack under {world} ranks runs REPLICATED: the engine refused the row-block "
                      f"plan ({e})", file=sys.stderr, flush=True)
            plan = None
            eng = mk(None)
        eng.set_model(W, b, Wlin, blin, Ws)
        eng.set_graph(_dense_np(ori_features), adj_np, ori_np, fadj, lab, idx)
        if self._adj_changes_init is not None:
            a0 = torch.as_tensor(self._adj_changes_init, dtype=torch.float32).to(dev).contiguous()
            eng.set_adj_changes(_bcast(dist, a0, host_staged) if plan is not None else a0)
        self.engine = eng
        if plan is not None:
            stepper = ShardedStepper(HipShardBackend(eng, plan), plan, dist=dist, host_staged=host_staged)
        self.sharded_world = world if plan is not None else 1

        idx_test_t = None
        if monitor and idx_test is not None:
            idx_test_t = torch.as_tensor(np.asarray(idx_test).astype(np.int64), device=dev)
            lab_t = torch.as_tensor(lab, device=dev)
        acc_test_list, sparsity_list = [], []
        for t in range(epochs):
            # adding_noise (:474-478): torch.randn_like on the attack device, as the reference draws it
            if self.loss_type == 'CE':
                if stepper is not None:
                    stepper.step()
                else:
                    eng.step(noise=torch.randn(n, n, device=dev) if eps != 0 else None)
            if monitor:
                if stepper is not None:
                    stepper.monitor(last=(t == epochs - 1))              # (:290-296), the next step adopts it
                    out2 = eng.buffer("logp") if idx_test_t is not None else None
                else:
                    out2, spars = eng.monitor(want_sparsity=False)       # (:290-296)
                if idx_test_t is not None:
                    acc_test_list.append((out2[idx_test_t].max(1)[1] == lab_t[idx_test_t]).double().mean())
        if acc_test_list:
            self.history["acc_test"] = [float(a) for a in torch.stack(acc_test_list).cpu()]
        self.history["path"] = dict(eng.path_stats(), fused_steps=eng.fused_steps(), sharded_world=self.sharded_world,
                                    collectives=stepper.exchanges if stepper is not None else 0)

        use_HA, use_YA, use_Y = bool(args.useH_A), bool(args.useY_A), bool(args.useY)
        H_A = self.H_A.detach() if (use_HA and self.H_A is not None) else None
        Y_A = self.Y_A.detach() if (use_YA and self.Y_A is not None) else None
        if use_HA and H_A is None:
            raise ValueError("args.useH_A needs H_A= at construction (main.py:298)")
        if use_YA and Y_A is None:
            raise ValueError("args.useY_A needs Y_A= at construction (main.py:298)")
        if H_A is not None and tuple(H_A.shape) != (n, dims[emb_nlayer]):
            raise ValueError(f"H_A has shape {tuple(H_A.shape)}; the {emb_nlayer}-layer embedding of this victim gives "
                             f"({n}, {dims[emb_nlayer]}) (main.py:238-241)")
        if Y_A is not None and tuple(Y_A.shape) != (n, int(Wlin.shape[0])):
            raise ValueError(f"Y_A has shape {tuple(Y_A.shape)}, expected ({n}, {int(Wlin.shape[0])})")
        if plan is not None:
            # the post-loop ensemble (:300-324) needs only what every rank holds (the last iteration's embedding and the
            # priors): it runs replicated, on rank 0's priors, and every rank returns the same modified_adj
            H_A = _bcast(dist, to_dev(H_A), host_staged) if H_A is not None else None
            Y_A = _bcast(dist, to_dev(Y_A), host_staged) if Y_A is not None else None
        self.modified_adj = eng.finalize(_decode_mode(args), H_A, Y_A, label_adj if use_Y else None).detach()
        return 0, 0, 0, 0
