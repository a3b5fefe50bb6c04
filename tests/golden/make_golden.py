#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE itself on CPU.

Runs only in the build container (needs /root/reference); the fixtures it
writes under tests/golden/ are data (inputs + the reference's outputs) and are
what travels to the GPU box.  Usage:  python tests/golden/make_golden.py [--only small|ops|cw|cora|cora_sparse|mid|bench10k|citeseer]

Environment shims applied before importing the reference (none of them is on
the computed path):
  * ``np.int = int``        - alias removed in numpy>=1.24, used by utils.py:505
  * ``torchmetrics.AUROC``  - imported at topology_attack.py:10 but only used by
                              ``metric()`` (:15-21), which main.py never calls;
                              the package is not installed here, so an empty
                              module object satisfies the import.
  * matplotlib Agg backend  - plt.cla() at topology_attack.py:122.
  * ``torch.linspace(..., device='cuda:0')`` - utils.MutualInformation (measure KDE) asks for its bins on 'cuda:0'
                              (utils.py:990-991) although every other tensor of the run follows ``self.device``; on this
                              CPU-only torch the keyword is dropped (the bins are then where the operands are).  No
                              arithmetic changes: the same linspace values, the same float32 ops.
The attack needs ./saved_data/<dataset>.npy (label adjacency, main.prepare()
main.py:440-450; the zip that ships it is a missing large blob), so the script
works in a temp cwd and writes label_adj[i,j] = (labels[i] == labels[j]) there.
"""
import argparse
import time as _time
import os
import random
import sys
import tempfile
import types
import zlib
from copy import deepcopy

import numpy as np

REF = "/root/reference/MC-GRA"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    np.int = int
    tm = types.ModuleType("torchmetrics")
    tm.AUROC = None
    sys.modules.setdefault("torchmetrics", tm)
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    import torch  # noqa
    _linspace = torch.linspace

    def linspace_where_the_operands_are(*a, **k):
        k.pop("device", None)
        return _linspace(*a, **k)

    torch.linspace = linspace_where_the_operands_are
    import utils  # noqa  (reference utils.py)
    import topology_attack  # noqa
    from models.gcn import GCN, embedding_GCN  # noqa
    return torch, utils, topology_attack, GCN, embedding_GCN


torch, rutils, rta, GCN, embedding_GCN = _import_reference()
from sklearn.metrics import auc, roc_curve  # noqa: E402


def ref_dot_product_decode_main(Z, dataset):
    """main.dot_product_decode (main.py:44-55) - feature_adj construction."""
    import torch.nn.functional as F
    if dataset in ("cora", "citeseer", "AIDS"):
        Z = torch.matmul(Z, Z.t())
        return torch.sigmoid(torch.relu(Z - torch.eye(Z.shape[0])))
    Z = F.normalize(Z, p=2, dim=1)
    Z = torch.matmul(Z, Z.t())
    return torch.relu(Z - torch.eye(Z.shape[0]))


def metric_pool(adj, inference_adj, idx):
    """main.metric_pool (main.py:66-75)."""
    real = adj[idx, :][:, idx].reshape(-1)
    pred = inference_adj[idx, :][:, idx].reshape(-1)
    fpr, tpr, _ = roc_curve(real, pred)
    return float(auc(fpr, tpr))


def weights_of(victim):
    """Victim weights in the unified layer form of oracle.GCNWeights (W = neighbour part, Ws = self part)."""
    d = {}
    if hasattr(victim, "attentions"):                       # models/gat.py GAT: heads concatenated, no bias
        for l, heads in enumerate(victim.attentions):
            W = torch.cat([a.W for a in heads], dim=1).detach().numpy().copy()
            d[f"W{l}"] = W
            d[f"b{l}"] = np.zeros(W.shape[1], np.float32)
        d["Wlin"] = victim.out_att.weight.detach().numpy().copy()
        d["blin"] = victim.out_att.bias.detach().numpy().copy()
        d["act"], d["head_act"] = "elu", "elu"
        return d
    sage = victim.gc[0].weight.shape[0] == 2 * victim.nfeat     # models/graphsage.py: weight is [2*in, out]
    for l, layer in enumerate(victim.gc):
        W = layer.weight.detach().numpy().copy()
        if sage:
            half = W.shape[0] // 2
            d[f"Ws{l}"] = W[:half].copy()                   # rows that multiply `input` (graphsage.py:44)
            W = W[half:].copy()                             # rows that multiply adj @ input
        d[f"W{l}"] = W
        d[f"b{l}"] = (layer.bias.detach().numpy().copy() if layer.bias is not None
                      else np.zeros(W.shape[1], np.float32))
    d["Wlin"] = victim.linear1.weight.detach().numpy().copy()
    d["blin"] = (victim.linear1.bias.detach().numpy().copy() if victim.linear1.bias is not None
                 else np.zeros(d["Wlin"].shape[0], np.float32))
    d["act"], d["head_act"] = "relu", "none"
    return d


def run_reference_attack(adj, features, labels, victim, idx_attack, measure, weight_param,
                         weight_sup, lr, epochs, dataset, use, num_edges, eps=0.0,
                         capture_steps=True, a0=None, loss_type="CE", ori_adj=None, noise_seed=None, f64=False):
    """Drive topology_attack.PGDAttack.attack exactly as main.objective does
    (main.py:298-307) and capture per-step adj_changes / grads through a global
    optimizer post-hook."""
    device = torch.device("cpu")
    n = adj.shape[0]
    if f64:
        # the REFERENCE's own code evaluated in float64 (default dtype float64, inputs / weights / adj_changes as doubles): what
        # its algorithm gives without fp32 rounding -- a truth for large-graph gradients that does not lean on oracle/
        torch.set_default_dtype(torch.float64)
        victim = deepcopy(victim).double()
        adj, features = adj.double(), features.double()
        # utils.to_tensor (utils.py:106-118; called at topology_attack.py:126-129) wraps its inputs in torch.FloatTensor: a type
        # conversion of data, replaced here by the same conversion to doubles
        orig_to_tensor = rutils.to_tensor

        def to_tensor64(adj, features, labels=None, device="cpu"):
            a_, f_ = torch.as_tensor(np.asarray(adj), dtype=torch.float64), torch.as_tensor(np.asarray(features), dtype=torch.float64)
            return (a_, f_) if labels is None else (a_, f_, torch.LongTensor(labels))

        rutils.to_tensor = to_tensor64
    try:
        return _run_reference_attack(adj, features, labels, victim, idx_attack, measure, weight_param, weight_sup, lr, epochs,
                                     dataset, use, num_edges, eps, capture_steps, a0, loss_type, ori_adj, noise_seed, f64, device, n)
    finally:
        torch.set_default_dtype(torch.float32)
        if f64:
            rutils.to_tensor = orig_to_tensor


def _run_reference_attack(adj, features, labels, victim, idx_attack, measure, weight_param, weight_sup, lr, epochs, dataset, use,
                          num_edges, eps, capture_steps, a0, loss_type, ori_adj, noise_seed, f64, device, n):
    fdt = np.float64 if f64 else np.float32
    if hasattr(victim, "attentions"):                                   # main.py:213-231
        from models.gat import embedding_gat
        nl = len(victim.attentions)
        embedding = embedding_gat(nfeat=features.shape[1], nclass=victim.nclass, nhid=victim.hidden_sizes[0],
                                  nlayer=nl, dropout=0.5, alpha=0.1, nheads=len(victim.attentions[0]), device=device)
        embedding.attentions = victim.attentions
    elif victim.gc[0].weight.shape[0] == 2 * victim.nfeat:              # main.py:193-210
        from models.graphsage import embedding_graphsage
        nl = len(victim.gc)
        embedding = embedding_graphsage(nfeat=features.shape[1], nhid=victim.hidden_sizes[0], nlayer=nl, device=device)
        embedding.gc = deepcopy(victim.gc)
    else:
        nl = len(victim.gc)
        embedding = embedding_GCN(nfeat=features.shape[1], nhid=victim.hidden_sizes[0], nlayer=nl, device=device)
        embedding.gc = deepcopy(victim.gc)                              # main.py:190
    victim.eval()
    H_A = embedding(features, adj); Y_A = victim(features, adj)         # main.py:235-236
    embedding.set_layers(1); embedding(features, adj)
    embedding.set_layers(2); H_A2 = embedding(features, adj)            # main.py:238-241
    feature_adj = ref_dot_product_decode_main(features, dataset)        # main.py:165
    init_adj = torch.zeros(n, n)                                        # dataset.init_matrix (dataset.py:433)
    if ori_adj is not None:     # a non-zero ori_adj (never produced by main.py; the class accepts it: :164, :185, :188, :302)
        init_adj = torch.tensor(np.asarray(ori_adj, dtype=fdt))
    args = argparse.Namespace(max_eval=100, lr=0, dataset=dataset, eps=eps, measure=measure,
                              useH_A=use[0], useY_A=use[1], useY=use[2],
                              w1=0, w2=0, w6=0, w7=0, w8=0, w9=0, w10=0)
    model = rta.PGDAttack(model=victim, embedding=embedding, H_A=H_A2, Y_A=Y_A, nnodes=n,
                          loss_type=loss_type, device=device).to(device)
    if f64:
        model = model.double()
    if a0 is not None:      # start away from the origin: adj_changes is a public Parameter (topology_attack.py:77)
        model.adj_changes.data = torch.tensor(np.asarray(a0, dtype=fdt))
    steps_a, steps_g, noises = [], [], []
    orig_randn_like = torch.randn_like

    def rec_randn_like(t, *a_, **k_):      # adding_noise's torch.randn_like (topology_attack.py:475), recorded
        if noise_seed is not None:
            # large graphs (5.6 - 29 MB of noise per step): the noise handed to the reference is a platform-independent
            # stream (numpy's legacy RandomState, bit-reproducible by its compatibility policy), so the fixture holds the
            # seed, checksums and a sample instead of the matrices; tests/helpers.py:noise_of regenerates and verifies it
            zn = seeded_noise(noise_seed, len(noises), tuple(t.shape))
            noises.append(noise_digest(zn))
            return torch.from_numpy(zn)
        z = orig_randn_like(t, *a_, **k_)
        noises.append(z.detach().numpy().copy())
        return z

    def hook(opt, a, k):
        p = opt.param_groups[0]["params"][0]
        steps_a.append(p.detach().numpy().copy())
        steps_g.append(p.grad.detach().numpy().copy())

    from torch.optim.optimizer import register_optimizer_step_post_hook
    handle = register_optimizer_step_post_hook(hook) if capture_steps else None
    if eps != 0:
        torch.randn_like = rec_randn_like
    try:
        model.attack(args, None, lr, 0, weight_sup, weight_param, feature_adj, 0, 0, 0,
                     None, None, np.arange(min(8, n)), adj, features, init_adj, labels, idx_attack,
                     num_edges, 0, epochs=epochs)
    finally:
        torch.randn_like = orig_randn_like
        if handle is not None:
            handle.remove()
    final = model.modified_adj.detach().numpy().copy()
    return dict(final=final, steps_a=steps_a, steps_g=steps_g, noises=noises, H_A2=H_A2.detach().numpy(),
                Y_A=Y_A.detach().numpy(), feature_adj=feature_adj.numpy(),
                auc=metric_pool(adj.numpy(), final, idx_attack))


def seeded_noise(seed, t, shape):
    """Step t's noise of a seeded eps != 0 run (same generator as tests/helpers.py:seeded_noise)."""
    return np.random.RandomState(int(seed) + int(t)).standard_normal(shape).astype(np.float32)


def noise_digest(zn):
    """[sum, sum of squares, first, last, centre entry] of a noise matrix in float64: what the fixture keeps of it."""
    z8 = zn.astype(np.float64)
    return np.array([z8.sum(), (z8 ** 2).sum(), z8[0, 0], z8[-1, -1], z8[zn.shape[0] // 2, zn.shape[1] // 3]])


def init_adj_changes(n, seed, scale):
    """Seeded non-zero start for adj_changes (tests regenerate it from the seed)."""
    return (np.random.RandomState(seed).rand(n * (n - 1) // 2) * scale).astype(np.float32)


def make_synth(n, f, c, hid, nlayer, seed, p_edge=0.08, arch="gcn"):
    rng = np.random.RandomState(seed)
    torch.manual_seed(seed); random.seed(seed)
    labels = rng.randint(0, c, size=n)
    # class-correlated features + homophilous edges so the attack has signal
    centers = rng.randn(c, f).astype(np.float32)
    feats = (centers[labels] * 0.6 + rng.randn(n, f) * 0.8).astype(np.float32)
    feats = (feats > 0.5).astype(np.float32)           # binary bag-of-words like cora
    same = labels[:, None] == labels[None, :]
    prob = np.where(same, p_edge * 3, p_edge * 0.5)
    up = np.triu(rng.rand(n, n) < prob, 1)
    adj = (up | up.T).astype(np.float32)
    device = torch.device("cpu")
    if arch == "gat":
        from models.gat import GAT
        victim = GAT(nfeat=f, nclass=c, nhid=hid, nlayer=nlayer, dropout=0.5, alpha=0.1, nheads=3, device=device)
    elif arch == "sage":
        from models.graphsage import graphsage
        victim = graphsage(nfeat=f, nclass=c, nhid=hid, nlayer=nlayer, dropout=0.5, weight_decay=5e-4, device=device)
    else:
        victim = GCN(nfeat=f, nclass=c, nhid=hid, nlayer=nlayer, dropout=0.5, weight_decay=5e-4, device=device)
    # reference reset_parameters / xavier init (models/gcn.py:28-33, gat.py:27-30); no training
    return torch.FloatTensor(adj), torch.FloatTensor(feats), torch.LongTensor(labels), victim


def gen_cw(tmp):
    """loss_type='CW' (topology_attack.py:329-335): the reference back-propagates the margin loss but never calls
    optimizer.step() (:277-280); the fixture pins what its run returns."""
    os.chdir(tmp)
    os.makedirs("saved_data", exist_ok=True)
    name, n, f, c, hid, nl = "s48_mse_cw", 48, 24, 4, 16, 2
    wp = (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)
    adj, feats, labels, victim = make_synth(n, f, c, hid, nl, seed=zlib.crc32(name.encode()) % 10000, arch="gcn")
    lab = labels.numpy()
    np.save("saved_data/cora.npy", (lab[:, None] == lab[None, :]).astype(np.float32))
    random.seed(7)
    idx_attack = np.array(random.sample(range(n), n))
    res = run_reference_attack(adj, feats, labels, victim, idx_attack, "MSELoss", wp, 1.0, 0.01, 3, "cora",
                               (True, True, True), 1e12, capture_steps=False, loss_type="CW")
    out = dict(adj=adj.numpy(), features=feats.numpy(), labels=lab, idx_attack=idx_attack, measure="MSELoss",
               weight_param=np.array(wp, dtype=np.float64), weight_sup=1.0, lr=0.01, epochs=3, num_edges=1e12, nlayer=nl,
               final=res["final"], H_A2=res["H_A2"], Y_A=res["Y_A"], feature_adj=res["feature_adj"], auc=res["auc"],
               **weights_of(victim))
    np.savez_compressed(os.path.join(OUT, f"cw_{name}.npz"), **out)
    print(name, "auc", res["auc"])


def gen_small(tmp, only=None):
    os.chdir(tmp)
    os.makedirs("saved_data", exist_ok=True)
    cases = []
    base_wp = (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)
    spec = [
        # name, n, f, c, hid, nlayer, measure, weight_param, wsup, lr, epochs, num_edges
        ("s48_mse", 48, 24, 4, 16, 2, "MSELoss", base_wp, 1.0, 0.01, 4, 1e12),
        ("s48_hsic", 48, 24, 4, 16, 2, "HSIC", base_wp, 1.0, 0.01, 4, 1e12),
        ("s48_kl", 48, 24, 4, 16, 2, "KL", base_wp, 1.0, 0.01, 3, 1e12),
        # CKA with w9/w10 is 0/0 at adj_changes == 0 (identical em / softmax rows): the reference returns
        # rounding noise there, so the pinned CKA case uses the N x N terms only (README polblogs/AIDS CKA lines)
        ("s48_cka", 48, 24, 4, 16, 2, "CKA", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 0, 0), 1.0, 0.01, 3, 1e12),
        ("s48_dp", 48, 24, 4, 16, 2, "DP", base_wp, 1.0, 0.01, 3, 1e12),
        ("s80_hsic_l3", 80, 40, 5, 16, 3, "HSIC", (0.1, 0.1, 0, 0, 0, 1, 1, 0, 100, 10), 0.0, 0.003, 3, 1e12),
        ("s80_mse_proj", 80, 40, 5, 16, 2, "MSELoss", (1, 0.1, 0, 0, 0, 100, 1, 0, 10, 10), 1.0, 0.1, 5, 30.0),
        ("s200_mse", 200, 64, 6, 16, 2, "MSELoss", base_wp, 1.0, 0.01, 3, 1e12),
        ("s200_hsic", 200, 64, 6, 16, 2, "HSIC", base_wp, 1.0, 0.01, 3, 1e12),
        # random (seeded) initial adj_changes: generic-sign gradients through every N x N term
        ("s48_hsic_init", 48, 24, 4, 16, 2, "HSIC", base_wp, 1.0, 0.01, 4, 1e12),
        ("s200_hsic_init", 200, 64, 6, 16, 2, "HSIC", base_wp, 1.0, 0.01, 4, 1e12),
        # CKA with the small-operand terms: non-degenerate away from the origin (distinct em rows)
        ("s48_cka_init", 48, 24, 4, 16, 2, "CKA", base_wp, 1.0, 0.01, 3, 1e12),
        # args.eps != 0: adding_noise (:474-478) makes modified_adj asymmetric and gates the gradient at the clamp
        ("s48_hsic_eps", 48, 24, 4, 16, 2, "HSIC", base_wp, 1.0, 0.01, 3, 1e12),
        ("s48_mse_eps", 48, 24, 4, 16, 2, "MSELoss", base_wp, 1.0, 0.01, 3, 1e12),
        ("s48_kl_eps", 48, 24, 4, 16, 2, "KL", base_wp, 1.0, 0.01, 3, 1e12),
        # other victims of main.py --arch: the dense "GAT" (gat.py:36-50) and GraphSAGE (graphsage.py:37-50)
        ("s48_gat_hsic_init", 48, 24, 4, 16, 2, "HSIC", base_wp, 1.0, 0.01, 3, 1e12),
        ("s48_gat_mse", 48, 24, 4, 16, 2, "MSELoss", base_wp, 1.0, 0.01, 3, 1e12),
        ("s48_sage_hsic_init", 48, 24, 4, 16, 2, "HSIC", base_wp, 1.0, 0.01, 3, 1e12),
        ("s48_sage_kl", 48, 24, 4, 16, 2, "KL", base_wp, 1.0, 0.01, 3, 1e12),
        ("s200_mse_init", 200, 64, 6, 16, 2, "MSELoss", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000), 1.0, 0.01, 4, 1e12),
        # eps != 0 at n >= 256: the gated branch of the one-pass tail kernel (rank-k + mirror + Adam) and the asymmetric
        # normalisation backward at the sizes where they are the default
        ("s300_hsic_eps", 300, 64, 6, 16, 2, "HSIC", base_wp, 1.0, 0.01, 2, 1e12),
        ("s300_mse_eps", 300, 64, 6, 16, 2, "MSELoss", base_wp, 1.0, 0.01, 2, 1e12),
        # a non-zero ori_adj (random symmetric 0/1): modified_adj = clamp(adj_changes + ori_adj) gates the gradient where
        # it saturates, the embedding runs on modified_adj - ori_adj (:185) while output2 runs on modified_adj (:259), and
        # modified_adj1 / the post-loop adjacency carry + ori_adj (:188, :302)
        ("s48_hsic_ori", 48, 24, 4, 16, 2, "HSIC", base_wp, 1.0, 0.01, 3, 1e12),
        ("s48_mse_ori", 48, 24, 4, 16, 2, "MSELoss", base_wp, 1.0, 0.01, 3, 1e12),
        # measure KDE (utils.MutualInformation as `calc`, topology_attack.py:199-201 and its four call sites): from the
        # origin, from seeded starts (every term with a generic gradient), with noise; + the first gradient of the reference's
        # own code in float64 (`step0_g64`)
        ("s48_kde", 48, 24, 4, 16, 2, "KDE", base_wp, 1.0, 0.01, 3, 1e12),
        ("s48_kde_init", 48, 24, 4, 16, 2, "KDE", base_wp, 1.0, 0.01, 4, 1e12),
        ("s200_kde_init", 200, 64, 6, 16, 2, "KDE", base_wp, 1.0, 0.01, 3, 1e12),
        ("s48_kde_eps", 48, 24, 4, 16, 2, "KDE", base_wp, 1.0, 0.01, 3, 1e12),
    ]
    if only is not None:
        spec = [c for c in spec if c[0] in only]
    for (name, n, f, c, hid, nl, measure, wp, wsup, lr, epochs, ne) in spec:
        arch = "gat" if "_gat_" in name else ("sage" if "_sage_" in name else "gcn")
        adj, feats, labels, victim = make_synth(n, f, c, hid, nl, seed=zlib.crc32(name.encode()) % 10000, arch=arch)
        lab = labels.numpy()
        np.save("saved_data/cora.npy", (lab[:, None] == lab[None, :]).astype(np.float32))
        random.seed(7)
        idx_attack = np.array(random.sample(range(n), n if "l3" not in name else int(n * 0.75)))
        a0 = None
        extra = {}
        eps = 0.0
        if arch == "gat":      # embedding_gat.forward ignores set_layers (gat.py:170-174): every embedding is full depth
            extra.update(arch=arch, emb_nlayer=nl, fin_layers=np.array([nl, nl]))
        elif arch == "sage":
            extra.update(arch=arch)
        ori = None
        if name.endswith("_ori"):
            rs = np.random.RandomState(zlib.crc32(name.encode()) % 10000 + 1)
            up = np.triu(rs.rand(n, n) < 0.06, 1)
            ori = (up | up.T).astype(np.float32)
            extra.update(ori_adj=ori, a0_seed=123, a0_scale=0.3)
            a0 = init_adj_changes(n, 123, 0.3)
        if name.endswith("_init") or name.endswith("_eps"):
            extra.update(a0_seed=123, a0_scale=0.05)
            a0 = init_adj_changes(n, 123, 0.05)
        if name.endswith("_eps"):
            eps = 0.02
        res = run_reference_attack(adj, feats, labels, victim, idx_attack, measure, wp, wsup, lr,
                                   epochs, "cora", (True, True, True), ne, a0=a0, eps=eps, ori_adj=ori)
        if eps != 0:
            extra.update(eps=eps, noise=np.stack(res["noises"]))
        if measure == "KDE" and eps == 0:
            r64 = run_reference_attack(adj, feats, labels, victim, idx_attack, measure, wp, wsup, lr, 1, "cora", (True, True, True),
                                       ne, a0=a0, ori_adj=ori, f64=True)
            extra.update(step0_g64=r64["steps_g"][0])
            print(name, "reference fp32 vs its own float64 run, first gradient:",
                  np.abs(res["steps_g"][0] - r64["steps_g"][0]).max() / np.abs(r64["steps_g"][0]).max())
        out = dict(adj=adj.numpy(), **extra, features=feats.numpy(), labels=lab, idx_attack=idx_attack,
                   measure=measure, weight_param=np.array(wp, dtype=np.float64), weight_sup=wsup,
                   lr=lr, epochs=epochs, num_edges=ne, nlayer=nl, final=res["final"],
                   steps_a=np.stack(res["steps_a"]), steps_g=np.stack(res["steps_g"]),
                   H_A2=res["H_A2"], Y_A=res["Y_A"], feature_adj=res["feature_adj"], auc=res["auc"],
                   **weights_of(victim))
        np.savez_compressed(os.path.join(OUT, f"attack_{name}.npz"), **out)
        print(name, "auc", res["auc"], "steps", len(res["steps_a"]))
        cases.append(name)
    return cases


def gen_ops(tmp):
    """Op-level known answers from the reference's own functions (+autograd)."""
    rng = np.random.RandomState(3)
    out = {}
    # normalize_adj_tensor (utils.py:211) incl. an isolated... (d >= 1 always with +I)
    A = rng.rand(37, 37).astype(np.float32); A = (A + A.T) / 2; np.fill_diagonal(A, 0)
    out["norm_in"] = A
    out["norm_out"] = rutils.normalize_adj_tensor(torch.tensor(A)).numpy()
    # CudaCKA linear_HSIC / linear_CKA (utils.py:1085-1096) with autograd
    cka = rutils.CudaCKA(device="cpu")
    for tag, (m, dx, dy) in {"a": (29, 16, 7), "b": (40, 40, 40)}.items():
        X = torch.tensor(rng.randn(m, dx).astype(np.float32), requires_grad=True)
        Y = torch.tensor(rng.randn(m, dy).astype(np.float32), requires_grad=True)
        for nm, fn in (("hsic", cka.linear_HSIC), ("cka", cka.linear_CKA)):
            X.grad = Y.grad = None
            v = fn(X, Y); v.backward()
            out[f"{nm}_{tag}_X"] = X.detach().numpy(); out[f"{nm}_{tag}_Y"] = Y.detach().numpy()
            out[f"{nm}_{tag}_val"] = v.item()
            out[f"{nm}_{tag}_gX"] = X.grad.numpy().copy(); out[f"{nm}_{tag}_gY"] = Y.grad.numpy().copy()
    # Info_entropy (topology_attack.py:44)
    P = torch.tensor(rng.rand(31, 31).astype(np.float32) * 1.2 - 0.1, requires_grad=True)
    v = rta.Info_entropy(P); v.backward()
    out["ie_in"] = P.detach().numpy(); out["ie_val"] = v.item(); out["ie_grad"] = P.grad.numpy().copy()
    # calc_kl / dot_product (topology_attack.py:480-487) with autograd; MSELoss
    import warnings
    warnings.simplefilter("ignore")
    dummy = rta.PGDAttack.__new__(rta.PGDAttack)
    for nm, fn in (("kl", lambda a, b: rta.PGDAttack.calc_kl(dummy, a, b)),
                   ("dp", lambda a, b: rta.PGDAttack.dot_product(dummy, a, b)),
                   ("mse", torch.nn.MSELoss())):
        X = torch.tensor(rng.randn(23, 9).astype(np.float32), requires_grad=True)
        Y = torch.tensor(rng.randn(23, 9).astype(np.float32), requires_grad=True)
        v = fn(X, Y); v.backward()
        out[f"{nm}_X"] = X.detach().numpy(); out[f"{nm}_Y"] = Y.detach().numpy()
        out[f"{nm}_val"] = v.item(); out[f"{nm}_gX"] = X.grad.numpy().copy(); out[f"{nm}_gY"] = Y.grad.numpy().copy()
    # dot_product_decode2 variants (topology_attack.py:421-467)
    Zr = rng.rand(19, 6).astype(np.float32)
    out["dd2_Z"] = Zr
    for ds, use in (("cora", (1, 1, 1)), ("AIDS", (1, 0, 0)), ("citeseer", (1, 1, 1)), ("brazil", (1, 1, 1)),
                    ("polblogs", (1, 1, 1)), ("polblogs", (1, 0, 1)), ("usair", (0, 0, 1)),
                    ("usair", (1, 1, 0)), ("usair", (1, 0, 1)), ("usair", (1, 1, 1))):
        dummy.args = argparse.Namespace(dataset=ds, useH_A=bool(use[0]), useY_A=bool(use[1]), useY=bool(use[2]))
        dummy.device = "cpu"
        out[f"dd2_{ds}_{use[0]}{use[1]}{use[2]}"] = rta.PGDAttack.dot_product_decode2(dummy, torch.tensor(Zr)).numpy()
    # projection with bisection (topology_attack.py:338-347, 397-412)
    class _P(torch.nn.Module):
        pass
    pm = rta.PGDAttack(model=None, embedding=None, nnodes=30, device="cpu")
    vec = (rng.rand(435).astype(np.float32) * 1.6 - 0.3)
    pm.adj_changes.data = torch.tensor(vec)
    pm.projection(40)
    out["proj_in"] = vec; out["proj_edges"] = 40.0; out["proj_out"] = pm.adj_changes.detach().numpy().copy()
    # get_modified_adj (topology_attack.py:365-379)
    pm.adj_changes.data = torch.tensor(rng.rand(435).astype(np.float32))
    ori = (rng.rand(30, 30) < 0.1).astype(np.float32)
    out["gma_a"] = pm.adj_changes.detach().numpy().copy(); out["gma_ori"] = ori
    out["gma_out"] = pm.get_modified_adj(torch.tensor(ori)).detach().numpy()
    # dot_product_decode packed (topology_attack.py:414-419)
    pm.nnodes = 19
    out["dd_out"] = pm.dot_product_decode(torch.tensor(Zr)).numpy()
    # hsic.py (Gaussian-kernel HSIC) with explicit sigma
    import hsic as rhsic
    hx = rng.randn(45, 8).astype(np.float32); hy = (hx[:, :5] * 0.5 + rng.randn(45, 5) * 0.7).astype(np.float32)
    out["ghsic_x"], out["ghsic_y"] = hx, hy
    for sg in (1.0, 5.0):
        out[f"ghsic_reg_{sg}"] = rhsic.hsic_regular(torch.tensor(hx), torch.tensor(hy), sigma=sg).item()
        out[f"ghsic_norm_{sg}"] = rhsic.hsic_normalized(torch.tensor(hx), torch.tensor(hy), sigma=sg).item()
    # the rest of hsic.py: sigma=None (median heuristic), distmat, distcorr, mmd, mmd_pxpy_pxy
    tx, ty = torch.tensor(hx), torch.tensor(hy)
    out["ghsic_sigma_xx"] = float(rhsic.sigma_estimation(tx, tx)); out["ghsic_sigma_yy"] = float(rhsic.sigma_estimation(ty, ty))
    hz = (hx[:, :5] * 0.3 + rng.randn(45, 5) * 0.5).astype(np.float32)          # same width as y: mmd needs x, y in one space
    out["ghsic_z"] = hz
    tz = torch.tensor(hz)
    out["ghsic_sigma_yz"] = float(rhsic.sigma_estimation(ty, tz))
    out["ghsic_reg_auto"] = rhsic.hsic_regular(tx, ty).item()
    out["ghsic_norm_auto"] = rhsic.hsic_normalized(tx, ty).item()
    out["ghsic_distmat"] = rhsic.distmat(tx).numpy()
    out["ghsic_distcorr_2.0"] = rhsic.distcorr(tx, sigma=2.0).item()
    for sg in (None, 1.5):
        out[f"ghsic_mmd_{sg}"] = rhsic.mmd(ty, tz, sigma=sg).item()
        out[f"ghsic_mmdp_{sg}"] = rhsic.mmd_pxpy_pxy(tx, ty, sigma=sg, use_cuda=False).item()
    # hsic_normalized_cca (hsic.py:138-151): the reference's fp32 value and the same formula in float64 (the two
    # regularised inverses are ill-conditioned; the fp32 value is up to 1e-2 off)
    def cca64(x, y, sx, sy):
        def R(X, s_):
            X = X.astype(np.float64); m = len(X)
            r = (X * X).sum(1)
            Kc = np.exp(-(r[:, None] - 2 * X @ X.T + r[None, :]) / (2 * s_ * s_)) @ (np.eye(m) - np.ones((m, m)) / m)
            return Kc @ np.linalg.inv(Kc + 1e-5 * m * np.eye(m))
        return float((R(x, sx) * R(y, sy).T).sum())
    for sg in (1.0, 5.0, None):
        out[f"ghsic_cca_{sg}"] = rhsic.hsic_normalized_cca(tx, ty, sigma=sg).item()
        sxx, syy = (sg, sg) if sg else (out["ghsic_sigma_xx"], out["ghsic_sigma_yy"])
        out[f"ghsic_cca64_{sg}"] = cca64(hx, hy, sxx, syy)
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)
    print("ops.npz", len(out), "arrays")


def gen_ops_kde(tmp):
    """utils.MutualInformation (utils.py:980-1049) as topology_attack.py constructs it (sigma=0.4, normalize=True, num_bins =
    the operands' width) on 2-D operands of the shapes its call sites have: value and autograd gradients in float32 and --
    the same module on double tensors -- in float64 -> ops_kde.npz."""
    rng = np.random.RandomState(11)
    out = {}
    cases = [("nxn48", 48, 48, 0.0, 1.0), ("nxn160", 160, 160, 0.0, 1.0), ("nxn160_ori", 160, 160, 0.0, 2.0),
             ("em16", 60, 16, 0.0, 4.0), ("sm7", 60, 7, None, None), ("wide20", 40, 20, -1.0, 21.0)]
    for tag, m, c, lo, hi in cases:
        if lo is None:      # Y_A (log-probs) against softmax(output2): the operands of c10 (:261-265)
            X = torch.log_softmax(torch.tensor(rng.randn(m, c).astype(np.float32)) * 2, 1).numpy()
            Y = torch.softmax(torch.tensor(rng.randn(m, c).astype(np.float32)) * 2, 1).numpy()
        else:
            X = (rng.rand(m, c) * (hi - lo) + lo).astype(np.float32)
            Y = (rng.rand(m, c) * (hi - lo) + lo).astype(np.float32)
        mi = rutils.MutualInformation(sigma=0.4, num_bins=c, normalize=True)
        for dt, sfx in ((torch.float32, ""), (torch.float64, "64")):
            tx, ty = torch.tensor(X, dtype=dt, requires_grad=True), torch.tensor(Y, dtype=dt, requires_grad=True)
            v = mi(tx, ty)
            assert tuple(v.shape) == (1,)
            v[0].backward()
            out.update({f"{tag}_val{sfx}": v.detach().numpy().copy(), f"{tag}_gx{sfx}": tx.grad.numpy().astype(np.float32),
                        f"{tag}_gy{sfx}": ty.grad.numpy().astype(np.float32)})
        out.update({f"{tag}_x": X, f"{tag}_y": Y})
        print(tag, float(out[f"{tag}_val"][0]), float(out[f"{tag}_val64"][0]),
              np.abs(out[f"{tag}_gx"] - out[f"{tag}_gx64"]).max() / np.abs(out[f"{tag}_gx64"]).max())
    out["cases"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(OUT, "ops_kde.npz"), **out)


def gen_cora(tmp, only=None):
    """Cora through the reference's own data path and victim training
    (main.py:148-190), README headline MSELoss config + an HSIC config."""
    os.chdir(tmp)
    if not os.path.exists("dataset"):
        os.symlink(os.path.join(REF, "dataset"), "dataset")
    os.makedirs("saved_data", exist_ok=True)
    from dataset import Dataset
    seed = 15
    np.random.seed(seed); random.seed(seed); torch.manual_seed(seed)
    data = Dataset(root="./dataset", name="cora", setting="GCN")
    adj, features, labels = data.adj, data.features, data.labels
    idx_train, idx_val, idx_test = data.idx_train, data.idx_val, data.idx_test
    idx_attack = np.array(random.sample(range(adj.shape[0]), int(adj.shape[0] * 1.0)))   # main.py:155
    adj, features, labels = rutils.preprocess(adj, features, labels, preprocess_adj=False, onehot_feature=False)
    device = torch.device("cpu")
    victim = GCN(nfeat=features.shape[1], nclass=labels.max().item() + 1, nhid=16, nlayer=2,
                 dropout=0.5, weight_decay=5e-4, device=device).to(device)
    victim.fit(features, adj, labels, idx_train, idx_val, verbose=False)
    idx_attack = np.array(random.sample(range(adj.shape[0]), int(adj.shape[0] * 1.0)))   # main.py:244
    num_edges = int(0.5 * 1e7 * adj.sum() / adj.shape[0] ** 2 * len(idx_attack) ** 2)
    lab = labels.numpy()
    np.save("saved_data/cora.npy", (lab[:, None] == lab[None, :]).astype(np.float32))
    rng = np.random.RandomState(0)
    n = adj.shape[0]
    samp = rng.randint(0, n, size=(8192, 2))
    fx = features.numpy()
    assert set(np.unique(fx)) <= {0.0, 1.0}
    ei = np.argwhere(np.triu(adj.numpy(), 1) > 0).astype(np.int32)
    assert np.array_equal(adj.numpy(), adj.numpy().T) and np.trace(adj.numpy()) == 0
    common = dict(idx_attack=idx_attack, idx_test=idx_test, num_edges=float(num_edges), sample_pos=samp,
                  features_bits=np.packbits(fx.astype(np.uint8), axis=1), nfeat=fx.shape[1],
                  adj_edges=ei, labels=lab, **weights_of(victim))
    readme = (0.01, 0, 0, 0, 0, 10, 10, 0, 10, 1000)
    runs = [
        # README.md "K = {X, H_A, Y^, Y}" cora line: --w1=0.01 --w6=10 --w7=10 --w9=10 --w10=1000 --lr=-2 MSELoss.
        # 100 epochs is the README horizon; Adam amplifies fp32 rounding noise on near-zero gradients, so the
        # trajectory is only reproducible to ~1e-3 in AUC there (an fp64 run of the same algorithm differs from
        # its fp32 run by 1e-7 / 4e-6 / 1.2e-4 / ~1e-3 at 10 / 20 / 40 / 100 epochs).  The 20-epoch run is the
        # one held to 1e-4; the 100-epoch run is re-run with another thread count to record the reference's
        # own spread under a changed summation order.
        ("cora_mse_readme", "MSELoss", readme, 1.0, 10 ** -2, 100, None),
        ("cora_mse_short", "MSELoss", readme, 1.0, 10 ** -2, 20, None),
        # HSIC: at adj_changes == 0 every c1/c2 gradient is >= 0 (PSD Gram), the origin is a fixed point in exact
        # arithmetic and the reference leaves it on rounding noise only; start from a seeded random adj_changes.
        ("cora_hsic", "HSIC", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000), 1.0, 10 ** -2, 8, (123, 0.05)),
        # the same from the bench's start rule (adj_changes_0 = U[0, 1) / n, lr = 1 / (50 n)), where the N x N terms carry
        # the gradient (from the dense start above they are 5e-13 of it: VERDICT round 2, item 1c)
        ("cora_hsic_sparse", "HSIC", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000), 1.0, 1.0 / (50 * 2708), 4, (123, 1.0 / 2708)),
    ]
    if only is not None:
        runs = [r for r in runs if r[0] in only]
    for name, measure, wp, wsup, lr, epochs, init in runs:
        a0 = init_adj_changes(adj.shape[0], *init) if init else None
        res = run_reference_attack(adj, features, labels, victim, idx_attack, measure, wp, wsup, lr, epochs,
                                   "cora", (True, True, True), num_edges, a0=a0)
        extra = dict(a0_seed=init[0], a0_scale=init[1]) if init else {}
        if name == "cora_mse_readme":
            torch.set_num_threads(3)
            alt = run_reference_attack(adj, features, labels, victim, idx_attack, measure, wp, wsup, lr, epochs,
                                       "cora", (True, True, True), num_edges, capture_steps=False)
            torch.set_num_threads(8)
            extra["auc_alt_threads"] = alt["auc"]
            print("  same run with 3 threads: auc", alt["auc"])
        sa = np.stack(res["steps_a"])
        out = dict(measure=measure, weight_param=np.array(wp, dtype=np.float64), weight_sup=wsup, lr=lr,
                   epochs=epochs, auc=res["auc"], **extra,
                   final_sample=res["final"][samp[:, 0], samp[:, 1]],
                   final_sum=float(res["final"].astype(np.float64).sum()),
                   step_sum=sa.astype(np.float64).sum(1), step_sqsum=(sa.astype(np.float64) ** 2).sum(1),
                   step_a_sample=sa[:, :: max(1, sa.shape[1] // 4096)],
                   **common)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
        print(name, "auc", res["auc"])


def gen_mid(tmp):
    """A reference run at a size where the default path is the fused low-rank step with the fp16-split product (n >= 1024)
    and from a start where the N x N terms carry the gradient: n = 1200, the bench's start
    rule (adj_changes_0 = U[0, 1) / n, lr = 1 / (50 n)).  Per-step gradient / adj_changes on 8k sampled packed positions,
    fp64 sums, a sample of the final ensemble, AUC; the graph travels as bits."""
    os.chdir(tmp)
    os.makedirs("saved_data", exist_ok=True)
    name, n, f, c, hid, nl = "s1200_hsic_sparse", 1200, 64, 6, 16, 2
    wp = (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)
    adj, feats, labels, victim = make_synth(n, f, c, hid, nl, seed=zlib.crc32(name.encode()) % 10000, p_edge=0.01)
    lab = labels.numpy()
    np.save("saved_data/cora.npy", (lab[:, None] == lab[None, :]).astype(np.float32))
    random.seed(7)
    idx_attack = np.array(random.sample(range(n), n))
    scale, lr, epochs = 1.0 / n, 1.0 / (50.0 * n), 3
    a0 = init_adj_changes(n, 123, scale)
    res = run_reference_attack(adj, feats, labels, victim, idx_attack, "HSIC", wp, 1.0, lr, epochs, "cora",
                               (True, True, True), 1e12, a0=a0)
    npk = n * (n - 1) // 2
    rng = np.random.RandomState(99)
    pk = np.unique(np.concatenate([np.arange(0, npk, max(1, npk // 4096)), rng.randint(0, npk, 4096)])).astype(np.int64)
    samp = rng.randint(0, n, size=(8192, 2))
    sa = np.stack(res["steps_a"]); sg = np.stack(res["steps_g"])
    fx = feats.numpy()
    assert set(np.unique(fx)) <= {0.0, 1.0}
    ei = np.argwhere(np.triu(adj.numpy(), 1) > 0).astype(np.int32)
    out = dict(features_bits=np.packbits(fx.astype(np.uint8), axis=1), nfeat=fx.shape[1], adj_edges=ei, labels=lab,
               idx_attack=idx_attack, measure="HSIC", weight_param=np.array(wp, dtype=np.float64), weight_sup=1.0, lr=lr,
               epochs=epochs, num_edges=1e12, nlayer=nl, a0_seed=123, a0_scale=scale, packed_pos=pk, sample_pos=samp,
               step_a=sa[:, pk], step_g=sg[:, pk], step_g_absmax=np.abs(sg).max(1),
               step_g_sum=sg.astype(np.float64).sum(1), step_g_sqsum=(sg.astype(np.float64) ** 2).sum(1),
               step_a_clip_sum=np.clip(sa, 0, 1).astype(np.float64).sum(1),
               final_sample=res["final"][samp[:, 0], samp[:, 1]], final_sum=float(res["final"].astype(np.float64).sum()),
               H_A2=res["H_A2"], Y_A=res["Y_A"], auc=res["auc"], **weights_of(victim))
    np.savez_compressed(os.path.join(OUT, f"mid_{name}.npz"), **out)
    print(name, "auc", res["auc"], "gmax per step", np.abs(sg).max(1), flush=True)


# README.md command lines of the reference, every one its CPU path can run: (tag, README line number, measure, use flags
# (H_A, Y_A, Y), weight_sup, lr exponent, eps, w1..w10 by position, start).  `start` None = the README's own start
# (adj_changes = 0); (seed, kappa) = a seeded start U[0, 1) * kappa / n with lr = kappa / (50 n) (scripts/nxn_share.py: where
# the N x N terms carry the gradient).  Not here: line 29 (tests/golden/cora_mse_*.npz).  The two --measure=KDE lines (13, 133)
# run with the `device='cuda:0'` keyword of utils.py:990-991 dropped (see the shims at the top).  The eps != 0 lines at n > 1000 (104, 112, 120: 5.6 - 29 MB of
# noise per step) run on SEEDED noise (NOISE_SEEDS) instead of recorded matrices; brazil's line 149 and the n <= 300 cases
# keep theirs.  Lines 116 and 120 sit in the README's usair section but name no dataset: they run on cora, as written.
# eps != 0 lines whose noise is a seeded platform-independent stream instead of recorded matrices (run_reference_attack)
NOISE_SEEDS = {104: 5104, 112: 5112, 120: 5120}

README_RUNS = {
    "brazil": [
        ("kl_h", 125, "KL", (1, 0, 0), 0.0, -1.0, 0.0, {1: 0.001, 2: 0.1, 6: 100, 7: 1000, 9: 0.01}, None),
        ("mse_y", 129, "MSELoss", (0, 1, 0), 0.0, -1.5, 0.0, {1: 0.0001, 6: 0.1, 10: 1}, None),
        ("mse_hy", 137, "MSELoss", (1, 1, 0), 0.0, -2.5, 0.0, {1: 0.0001, 2: 1, 6: 0.0001, 7: 0.001, 9: 100, 10: 1000}, None),
        ("kl_hY", 141, "KL", (1, 0, 1), 1.0, -2.0, 0.0, {1: 0.001, 2: 100, 7: 0.01, 9: 0.001}, None),
        ("dp_yy", 145, "DP", (0, 1, 1), 1.0, -1.0, 0.0, {6: 10000, 10: 1}, None),
        ("kl_all_eps", 149, "KL", (1, 1, 1), 1.0, 0.0, 0.077458886396933, {1: 10, 2: 0.001, 6: 0.1, 9: 0.1, 10: 100}, None),
        ("kde_Y", 133, "KDE", (0, 0, 1), 1.0, -2.5, 0.0, {1: 0.0001, 6: 0.001}, None),
    ],
    "usair": [
        ("mse_h", 96, "MSELoss", (1, 0, 0), 0.0, -3.0, 0.0, {2: 100, 6: 100, 7: 10000, 9: 100}, None),
        ("mse_y", 100, "MSELoss", (0, 1, 0), 0.0, -2.0, 0.0, {1: 1000, 6: 0.01, 10: 0.001}, None),
        # line 108 from the sparse start.  From its own start (adj_changes = 0) the reference's c9 / c10 gradients are
        # rounding noise on this dataset: with identity attributes and an empty graph H_A-vs-em and Y_A-vs-softmax HSIC
        # are 1e-18 / 1e-11 in float64, the fp32 Gram evaluation returns a gradient of largest magnitude 88 where the
        # exact one is 40 (c2 alone) -- nothing can be pinned on that output
        ("hsic_hy_sparse", 108, "HSIC", (1, 1, 0), 0.0, None, 0.0, {1: 10000, 2: 0.0001, 6: 0.0001, 7: 0.0001, 9: 0.01, 10: 0.0001}, (123, 1.0)),
        # the eps != 0 lines at n > 1000 (round 4): seeded noise (NOISE_SEEDS), the asymmetric general-path tail and the
        # column-mean centring at their large-graph launch shapes
        ("dp_Y_eps", 104, "DP", (0, 0, 1), 1.0, -2.5, 0.001986024928134464, {6: 0.001}, None),
        ("hsic_hY_eps", 112, "HSIC", (1, 0, 1), 1.0, -2.5, 0.01189830603305939, {1: 0.01, 2: 0.001, 6: 1000, 7: 0.1, 9: 0.001}, None),
    ],
    "polblogs": [
        ("kl_h", 65, "KL", (1, 0, 0), 0.0, -2.5, 0.0, {1: 0.0001, 7: 100, 9: 1000}, None),
        ("dp_y", 69, "DP", (0, 1, 0), 0.0, -2.5, 0.0, {6: 100, 10: 1}, None),
        ("mse_Y", 73, "MSELoss", (0, 0, 1), 1.0, -2.5, 0.0, {1: 10, 6: 1000}, None),
        ("mse_hy", 77, "MSELoss", (1, 1, 0), 0.0, -3.0, 0.0, {1: 100, 2: 1, 6: 0.1, 7: 0.01, 9: 1000}, None),
        ("hsic_hY", 81, "HSIC", (1, 0, 1), 1.0, -1.0, 0.0, {1: 0.1, 2: 0.01, 6: 100, 7: 10000, 9: 0.001}, None),
        ("cka_yy", 85, "CKA", (0, 1, 1), 1.0, 0.0, 0.0, {1: 0.01, 6: 100}, None),
        # line 90 from the sparse start (c2 through the fused low-rank step at n = 1490; c1 is off on this dataset: identity
        # attributes make feature_adj constant, topology_attack.py:212).  From its own start the reference's first gradient
        # has largest magnitude 4.1e7 where the exact one has 4.0e3 (rounding noise of c9 / c10 times w10 = 1000, as on
        # usair line 108): not a fixture
        ("hsic_all_sparse", 90, "HSIC", (1, 1, 1), 1.0, None, 0.0, {1: 0.01, 2: 0.01, 6: 10000, 7: 100, 9: 0.001, 10: 1000}, (123, 1.0)),
    ],
    "AIDS": [
        ("kl_h", 154, "KL", (1, 0, 0), 0.0, -3.0, 0.0, {1: 1, 2: 1000, 6: 10000, 7: 0.01, 9: 1000}, None),
        ("mse_y", 158, "MSELoss", (0, 1, 0), 0.0, -2.5, 0.0, {1: 1, 6: 1, 10: 0.01}, None),
        ("cka_Y", 162, "CKA", (0, 0, 1), 1.0, -2.5, 0.0, {1: 1, 6: 0.0001}, None),
        ("mse_hy", 166, "MSELoss", (1, 1, 0), 0.0, 0.0, 0.0, {6: 0.001, 7: 10, 9: 1, 10: 100}, None),
        ("mse_hY", 170, "MSELoss", (1, 0, 1), 1.0, -1.0, 0.0, {1: 10, 6: 0.0001, 7: 1, 9: 0.1}, None),
        ("mse_yY", 174, "MSELoss", (0, 1, 1), 1.0, -2.5, 0.0, {1: 100}, None),
        ("kl_all", 179, "KL", (1, 1, 1), 1.0, -3.0, 0.0, {1: 0.0001, 2: 1, 7: 100, 9: 1000, 10: 0.0001}, None),
        # not a README line: HSIC with the Cora weights on the one small dataset whose attributes make feature_adj
        # non-constant, from the sparse start -- the N x N x N product on real data at n = 1429
        ("hsic_all_sparse", 0, "HSIC", (1, 1, 1), 1.0, None, 0.0, {1: 0.01, 2: 0.01, 6: 10, 7: 10, 9: 10, 10: 1000}, (123, 1.0)),
    ],
    "cora": [
        ("mse_h", 5, "MSELoss", (1, 0, 0), 0.0, -2.5, 0.0, {1: 0.1, 2: 0.1, 6: 10000, 7: 100, 9: 1000}, None),
        ("kl_y", 9, "KL", (0, 1, 0), 0.0, -2.5, 0.0, {6: 100, 10: 0.1}, None),
        ("mse_hy", 17, "MSELoss", (1, 1, 0), 0.0, -2.0, 0.0, {1: 1000, 2: 0.001, 6: 0.1, 7: 0.1, 9: 100, 10: 100}, None),
        ("mse_hY", 21, "MSELoss", (1, 0, 1), 1.0, -2.0, 0.0, {1: 100, 2: 0.0001, 6: 0.0001, 7: 1, 9: 10}, None),
        ("mse_yY", 25, "MSELoss", (0, 1, 1), 1.0, -3.0, 0.0, {1: 0.1, 6: 10, 10: 0.01}, None),
        ("kde_Y", 13, "KDE", (0, 0, 1), 1.0, -3.0, 0.0, {1: 1000, 6: 0.01}, None),
        ("cka_yY", 116, "CKA", (0, 1, 1), 1.0, -2.0, 0.0, {1: 1000, 6: 1000, 10: 0.01}, None),
        # line 120: a NEGATIVE eps (seeded noise)
        ("dp_all_eps_neg", 120, "DP", (1, 1, 1), 1.0, 0.0, -0.010192774962135321, {1: 10000, 2: 1, 6: 10, 7: 0.1, 9: 0.01, 10: 0.01}, None),
    ],
    "citeseer": [
        ("kl_h", 35, "KL", (1, 0, 0), 0.0, -2.5, 0.0, {1: 10, 2: 0.1, 6: 0.01, 7: 0.001, 9: 10}, None),
        ("kl_y", 39, "KL", (0, 1, 0), 0.0, -1.5, 0.0, {1: 0.0001, 10: 1}, None),
        ("mse_Y", 43, "MSELoss", (0, 0, 1), 1.0, -3.0, 0.0, {1: 100, 6: 100}, None),
        ("mse_hy", 47, "MSELoss", (1, 1, 0), 0.0, -2.5, 0.0, {1: 100, 2: 0.001, 6: 10, 7: 100, 9: 100, 10: 0.001}, None),
        ("kl_hY", 51, "KL", (1, 0, 1), 1.0, -1.0, 0.0, {1: 0.001, 2: 10000, 6: 0.0001, 7: 100, 9: 100}, None),
        ("kl_yY", 55, "KL", (0, 1, 1), 1.0, -2.0, 0.0, {1: 10, 6: 1, 10: 10000}, None),
        ("kl_all", 59, "KL", (1, 1, 1), 1.0, -1.5, 0.0, {1: 100, 2: 0.0001, 6: 0.001, 9: 1000, 10: 0.001}, None),
    ],
}


def gen_readme(tmp, only=None, epochs=6, horizon=None):
    """The reference on its README.md lines, through its own Dataset, preprocess and GCN.fit (main.py:147-190), `epochs`
    steps each.  (horizon: a dict -- nothing is written per line; the lines are run for `epochs` steps as they are (eps != 0: on seeded noise) and
    once more with the reference's own code in float64, and only what a longer horizon can be held to is kept: both AUCs, a
    sample of the ensemble and its sum -- gen_readme_horizon.)  Per dataset one `readme_<dataset>_graph.npz` with what the loader produced (edges, diagonal -- brazil has
    self loops --, attributes as bits / the identity flag / float32, labels, the three index splits, idx_attack) and the
    trained weights; per line one `readme_<dataset>_<tag>.npz` with the per-step gradient at sampled packed positions,
    adj_changes there after the last step, a sample of the post-loop ensemble, its sum and the AUC."""
    os.chdir(tmp)
    if not os.path.exists("dataset"):
        os.symlink(os.path.join(REF, "dataset"), "dataset")
    os.makedirs("saved_data", exist_ok=True)
    from dataset import Dataset
    device = torch.device("cpu")
    for ds, runs in README_RUNS.items():
        if only is not None and not any(f"{ds.lower()}_{r[0]}" in only or ds.lower() in only for r in runs):
            continue
        seed = 15
        np.random.seed(seed); random.seed(seed); torch.manual_seed(seed)                     # main.py:141-143
        data = Dataset(root="./dataset", name=ds, setting="GCN")
        adj, features, labels = data.adj, data.features, data.labels
        idx_train, idx_val, idx_test = data.idx_train, data.idx_val, data.idx_test
        idx_attack = np.array(random.sample(range(adj.shape[0]), int(adj.shape[0] * 1.0)))   # main.py:155
        adj, features, labels = rutils.preprocess(adj, features, labels, preprocess_adj=False, onehot_feature=False)
        victim = GCN(nfeat=features.shape[1], nclass=labels.max().item() + 1, nhid=16, nlayer=2,
                     dropout=0.5, weight_decay=5e-4, device=device).to(device)
        victim.fit(features, adj, labels, idx_train, idx_val, verbose=False)
        victim_clean = None
        if horizon is not None:      # (fit and attack leave non-leaf tensors on the module: a copy for the float64 run is built from the weights)
            victim_clean = GCN(nfeat=features.shape[1], nclass=labels.max().item() + 1, nhid=16, nlayer=2,
                               dropout=0.5, weight_decay=5e-4, device=device).to(device)
            victim_clean.load_state_dict(victim.state_dict())
            victim_clean.eval()
        idx_attack = np.array(random.sample(range(adj.shape[0]), int(adj.shape[0] * 1.0)))   # main.py:244
        num_edges = int(0.5 * 1e7 * adj.sum() / adj.shape[0] ** 2 * len(idx_attack) ** 2)
        lab = labels.numpy()
        np.save(f"saved_data/{ds}.npy", (lab[:, None] == lab[None, :]).astype(np.float32))
        n = adj.shape[0]
        a = adj.numpy()
        assert np.array_equal(a, a.T) and set(np.unique(a)) <= {0.0, 1.0}
        fx = features.numpy()
        ident = fx.shape == (n, n) and np.array_equal(fx, np.eye(n, dtype=np.float32))
        rng = np.random.RandomState(0)
        samp = rng.randint(0, n, size=(8192, 2))
        npk = n * (n - 1) // 2
        pk = np.unique(np.concatenate([np.arange(0, npk, max(1, npk // 4096)), rng.randint(0, npk, 4096)])).astype(np.int64)
        graph = dict(dataset=ds, idx_attack=idx_attack, idx_train=idx_train, idx_val=idx_val, idx_test=idx_test,
                     num_edges=float(num_edges), sample_pos=samp, packed_pos=pk, features_identity=int(ident),
                     adj_edges=np.argwhere(np.triu(a, 1) > 0).astype(np.int32), adj_diag=np.diag(a).astype(np.uint8).copy(),
                     labels=lab, nlayer=2, **weights_of(victim))
        if not ident and set(np.unique(fx)) <= {0.0, 1.0}:
            graph.update(features_bits=np.packbits(fx.astype(np.uint8), axis=1), nfeat=fx.shape[1])
        elif not ident:
            graph["features_f32"] = fx
        H_A2 = Y_A = None
        for (tag, line, measure, use, wsup, lrexp, eps, w, start) in runs:
            name = f"readme_{ds.lower()}_{tag}"
            if only is not None and not (f"{ds.lower()}_{tag}" in only or ds.lower() in only):
                continue
            wp = tuple(float(w.get(i, 0)) for i in range(1, 11))
            a0, extra = None, {}
            lr = 10.0 ** lrexp if lrexp is not None else None
            if start is not None:
                a0 = init_adj_changes(n, start[0], start[1] / n)
                extra.update(a0_seed=start[0], a0_scale=start[1] / n)
                if lr is None:
                    lr = start[1] / (50.0 * n)
            if horizon is not None:
                # (an eps != 0 line: every step's adding_noise draw is the seeded stream of NOISE_SEEDS -- brazil's line 149, whose
                # six-epoch fixture keeps recorded matrices, gets a seed of its own here -- handed to both runs)
                nseed = (NOISE_SEEDS.get(line, 5000 + line)) if eps != 0 else None
                t0 = _time.time()
                res = run_reference_attack(adj, features, labels, victim, idx_attack, measure, wp, wsup, lr, epochs, ds,
                                           tuple(bool(u) for u in use), num_edges, eps=eps, a0=a0, capture_steps=False, noise_seed=nseed)
                res64 = run_reference_attack(adj, features, labels, deepcopy(victim_clean), idx_attack, measure, wp, wsup, lr, epochs, ds,
                                             tuple(bool(u) for u in use), num_edges, eps=eps, a0=a0, capture_steps=False, f64=True,
                                             noise_seed=nseed)
                if nseed is not None:
                    horizon[f"{name}_noise_seed"] = nseed
                horizon[f"{name}_auc"] = res["auc"]; horizon[f"{name}_auc64"] = res64["auc"]
                horizon[f"{name}_final_sample"] = res["final"][samp[:, 0], samp[:, 1]].astype(np.float32)
                horizon[f"{name}_final_sum"] = float(res["final"].astype(np.float64).sum())
                print(name, epochs, "epochs: auc", res["auc"], "float64", res64["auc"], "diff", abs(res["auc"] - res64["auc"]),
                      "seconds", round(_time.time() - t0, 1), flush=True)
                continue
            torch.manual_seed(1000 + line)       # (the noise of an eps != 0 line; recorded below)
            res = run_reference_attack(adj, features, labels, victim, idx_attack, measure, wp, wsup, lr, epochs, ds,
                                       tuple(bool(u) for u in use), num_edges, eps=eps, a0=a0, noise_seed=NOISE_SEEDS.get(line))
            if eps != 0 and line in NOISE_SEEDS:
                extra.update(noise_seed=NOISE_SEEDS[line], noise_digest=np.stack(res["noises"]))
            elif eps != 0:
                extra.update(noise=np.stack(res["noises"]))
            if H_A2 is None:
                H_A2, Y_A = res["H_A2"], res["Y_A"]
            assert np.array_equal(H_A2, res["H_A2"]) and np.array_equal(Y_A, res["Y_A"])
            sa = np.stack(res["steps_a"]); sg = np.stack(res["steps_g"])
            out = dict(dataset=ds, readme_line=line, measure=measure, use=np.array(use), weight_param=np.array(wp, dtype=np.float64),
                       weight_sup=wsup, lr=lr, eps=eps, epochs=epochs, auc=res["auc"], **extra,
                       step_g=sg[:, pk], step_g_absmax=np.abs(sg).max(1), last_a=sa[-1, pk],
                       step_g_sum=sg.astype(np.float64).sum(1), step_g_sqsum=(sg.astype(np.float64) ** 2).sum(1),
                       step_a_clip_sum=np.clip(sa, 0, 1).astype(np.float64).sum(1),
                       final_sample=res["final"][samp[:, 0], samp[:, 1]],
                       final_sum=float(res["final"].astype(np.float64).sum()))
            np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
            print(name, "auc", res["auc"], "gmax per step", np.abs(sg).max(1), flush=True)
        if H_A2 is not None and horizon is None:
            np.savez_compressed(os.path.join(OUT, f"readme_{ds.lower()}_graph.npz"), H_A2=H_A2, Y_A=Y_A, **graph)


def gen_readme_horizon(tmp, epochs=20, only=None):
    """The README lines (eps == 0) at a longer horizon than the per-step fixtures' six epochs: the reference for `epochs` steps in
    float32 (as it runs) and in float64 (its own code under torch float64) from the same trained victim -- their distance is what
    the reference's own arithmetic leaves of "the" AUC at that horizon (Adam turns rounding noise on near-zero gradients into
    +-lr moves).  One file, horizon<epochs>_readme.npz: per line both AUCs, a sample of the ensemble, its sum.  About an hour
    on 8 cores."""
    path = os.path.join(OUT, f"horizon{epochs}_readme.npz")
    datasets = [d for d in README_RUNS if only is None or any(d.lower() in o for o in only)]
    for ds in datasets:      # one dataset at a time, the file rewritten behind each (hours at 100 epochs: nothing is lost to an interruption)
        out = {}
        sel = [o for o in only if ds.lower() in o] if only is not None else [ds.lower()]
        gen_readme(tmp, only=sel, epochs=epochs, horizon=out)
        if os.path.exists(path):      # keeps the other lines
            old = dict(np.load(path))
            assert int(old["epochs"]) == epochs
            old.update(out)
            out = old
        out["epochs"] = epochs
        np.savez_compressed(path, **out)


def gen_citeseer_gat(tmp, train_iters=6):
    """BASELINE.json configs[2]: Citeseer through the reference's Dataset, the dense GAT victim (models/gat.py:176-206,
    5 heads x 16 as main.py:213-216) trained by the reference's own GAT.fit, embedding_gat sharing its attention layers
    (main.py:226-231), priors H_A + Y, the citeseer branch of dot_product_decode2 (:427-431).  GAT.fit runs
    `train_iters` iterations instead of main.py's 200 (each costs ~10 s on the CPU: the layer materialises an
    N x N x 2F attention input, gat.py:36-41); the weights travel in the fixture, so the attack sees exactly this victim.
    Two configurations: the README's `K = {X, H_A, Y}` citeseer line (measure KL) and an HSIC one (Gram evaluation:
    ELU embeddings of width 80)."""
    os.chdir(tmp)
    if not os.path.exists("dataset"):
        os.symlink(os.path.join(REF, "dataset"), "dataset")
    os.makedirs("saved_data", exist_ok=True)
    from dataset import Dataset
    from models.gat import GAT
    import time as _t
    seed = 15
    np.random.seed(seed); random.seed(seed); torch.manual_seed(seed)
    data = Dataset(root="./dataset", name="citeseer", setting="GCN")
    adj, features, labels = data.adj, data.features, data.labels
    idx_train, idx_val, idx_test = data.idx_train, data.idx_val, data.idx_test
    idx_attack = np.array(random.sample(range(adj.shape[0]), int(adj.shape[0] * 1.0)))   # main.py:155
    adj, features, labels = rutils.preprocess(adj, features, labels, preprocess_adj=False, onehot_feature=False)
    device = torch.device("cpu")
    victim = GAT(nfeat=features.shape[1], nclass=labels.max().item() + 1, nhid=16, nlayer=2, dropout=0.5, alpha=0.1,
                 nheads=5, device=device)
    t0 = _t.time()
    victim.fit(features, adj, labels, idx_train, idx_val, train_iters=train_iters)
    print("GAT.fit", train_iters, "iterations:", round(_t.time() - t0, 1), "s", flush=True)
    idx_attack = np.array(random.sample(range(adj.shape[0]), int(adj.shape[0] * 1.0)))   # main.py:244
    num_edges = int(0.5 * 1e7 * adj.sum() / adj.shape[0] ** 2 * len(idx_attack) ** 2)
    lab = labels.numpy()
    np.save("saved_data/citeseer.npy", (lab[:, None] == lab[None, :]).astype(np.float32))
    n = adj.shape[0]
    rng = np.random.RandomState(0)
    samp = rng.randint(0, n, size=(8192, 2))
    npk = n * (n - 1) // 2
    pk = np.unique(np.concatenate([np.arange(0, npk, max(1, npk // 4096)), rng.randint(0, npk, 4096)])).astype(np.int64)
    fx = features.numpy()
    assert set(np.unique(fx)) <= {0.0, 1.0}
    ei = np.argwhere(np.triu(adj.numpy(), 1) > 0).astype(np.int32)
    assert np.array_equal(adj.numpy(), adj.numpy().T) and np.trace(adj.numpy()) == 0
    common = dict(idx_attack=idx_attack, idx_test=idx_test, num_edges=float(num_edges), sample_pos=samp, packed_pos=pk,
                  features_bits=np.packbits(fx.astype(np.uint8), axis=1), nfeat=fx.shape[1], adj_edges=ei, labels=lab,
                  arch="gat", nheads=5, emb_nlayer=2, fin_layers=np.array([2, 2]), dataset="citeseer",
                  use=np.array([1, 0, 1]), train_iters=train_iters, **weights_of(victim))
    runs = [
        # README.md citeseer "K = {X, H_A, Y}" line: --w1=0.001 --w2=10000 --w6=0.0001 --w7=100 --w9=100 --lr=-1 KL
        ("citeseer_gat_kl", "KL", (0.001, 10000, 0, 0, 0, 0.0001, 100, 0, 100, 0), 1.0, 10 ** -1, 3, None),
        # HSIC on the same victim, seeded start (the origin is a fixed point of the exact HSIC dynamics)
        ("citeseer_gat_hsic", "HSIC", (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 0), 1.0, 10 ** -2, 3, (123, 0.05)),
    ]
    for name, measure, wp, wsup, lr, epochs, init in runs:
        a0 = init_adj_changes(n, *init) if init else None
        t0 = _t.time()
        res = run_reference_attack(adj, features, labels, victim, idx_attack, measure, wp, wsup, lr, epochs,
                                   "citeseer", (True, False, True), num_edges, a0=a0)
        extra = dict(a0_seed=init[0], a0_scale=init[1]) if init else {}
        sa = np.stack(res["steps_a"]); sg = np.stack(res["steps_g"])
        out = dict(measure=measure, weight_param=np.array(wp, dtype=np.float64), weight_sup=wsup, lr=lr, epochs=epochs,
                   auc=res["auc"], **extra,
                   final_sample=res["final"][samp[:, 0], samp[:, 1]], final_sum=float(res["final"].astype(np.float64).sum()),
                   step_a=sa[:, pk], step_g=sg[:, pk], step_g_absmax=np.abs(sg).max(1),
                   step_g_sum=sg.astype(np.float64).sum(1), step_g_sqsum=(sg.astype(np.float64) ** 2).sum(1),
                   step_a_clip_sum=np.clip(sa, 0, 1).astype(np.float64).sum(1),
                   H_A2=res["H_A2"], Y_A=res["Y_A"], **common)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
        print(name, "auc", res["auc"], "seconds", round(_t.time() - t0, 1), flush=True)


def gen_bench(tmp, workload="synthetic-10k-hsic", tag="bench10k_hsic", seed=0, epochs=4, single_starts=((1, 0.5), (2, 2.0))):
    """The bench's OWN workload (bench.WORKLOADS[workload], bench.make_inputs, bench.make_a0) through the
    reference's PGDAttack.attack on CPU (topology_attack.py:161-324).  The inputs are regenerated from the seed by
    the test, so the fixture holds only the reference's outputs: per-step gradient / adj_changes on a fixed sample
    of packed positions, fp64 step sums, a sample of the final ensemble and the AUC.  Besides the `epochs`-step
    run from bench.make_a0(n, seed), `single_starts` = ((seed offset, scale), ...) are one-step runs from other
    seeded starts: each is a step whose starting state the test can rebuild exactly (the 50 M-entry adj_changes of
    later steps of a run cannot be stored), i.e. teacher forcing by construction.  Scales are multiples of the
    workload's own start scale bench.start_scale (kappa / N): the run starts where the loss's N x N terms and its
    small-operand terms both carry the gradient (scripts/nxn_share.py), `one0` at half of it (N x N terms in charge),
    `one1` at twice (small-operand terms in charge); lr is the workload's (bench.workload_lr)."""
    os.chdir(tmp)
    os.makedirs("saved_data", exist_ok=True)
    # bench.py of the repository (the generator and the start scale are its): beside this script's tests/golden, or -- for
    # a copy of this script run elsewhere -- $MCGRA_REPO, or /root/repo
    for root in (os.path.dirname(os.path.dirname(OUT)), os.environ.get("MCGRA_REPO", ""), "/root/repo"):
        if root and os.path.exists(os.path.join(root, "bench.py")):
            sys.path.insert(0, root)
            break
    import bench as B
    n, f, c, hid, nl, measure, wp = B.WORKLOADS[workload]
    inp = B.make_inputs(n, f, c, hid, nl, seed)
    device = torch.device("cpu")
    victim = GCN(nfeat=f, nclass=c, nhid=hid, nlayer=nl, dropout=0.5, weight_decay=5e-4, device=device)
    with torch.no_grad():
        for l in range(nl):
            victim.gc[l].weight.copy_(torch.tensor(inp["W"][l])); victim.gc[l].bias.copy_(torch.tensor(inp["b"][l]))
        victim.linear1.weight.copy_(torch.tensor(inp["Wlin"])); victim.linear1.bias.copy_(torch.tensor(inp["blin"]))
    lab = inp["labels"]
    np.save("saved_data/cora.npy", (lab[:, None] == lab[None, :]).astype(np.float32))
    adj = torch.tensor(inp["adj"]); feats = torch.tensor(inp["features"]); labels = torch.LongTensor(lab)
    npk = n * (n - 1) // 2
    rng = np.random.RandomState(99)
    pk = np.unique(np.concatenate([np.arange(0, npk, max(1, npk // 4096)), rng.randint(0, npk, 4096)])).astype(np.int64)
    samp = rng.randint(0, n, size=(8192, 2))
    lr, sc0 = B.workload_lr(workload, n), B.start_scale(workload, n)
    out = dict(workload=workload, seed=seed, epochs=epochs, packed_pos=pk, sample_pos=samp, lr=lr, weight_sup=1.0,
               weight_param=np.array(wp, dtype=np.float64), measure=measure, start_scale=sc0)
    import time as _t
    runs = [("run", seed, sc0, epochs)] + [(f"one{k}", seed + off, sc0 * mul, 1) for k, (off, mul) in enumerate(single_starts)]
    for name, sd, sc, ep in runs:
        a0 = B.make_a0(n, sd, sc)
        t0 = _t.time()
        res = run_reference_attack(adj, feats, labels, victim, inp["idx_attack"], measure, wp, 1.0, lr, ep,
                                   "cora", (True, True, True), 1e30, a0=a0)
        sa = np.stack(res["steps_a"]); sg = np.stack(res["steps_g"])
        out.update({f"{name}_a0_seed": sd, f"{name}_a0_scale": sc,
                    f"{name}_a": sa[:, pk], f"{name}_g": sg[:, pk],
                    f"{name}_a_sum": sa.astype(np.float64).sum(1), f"{name}_a_sqsum": (sa.astype(np.float64) ** 2).sum(1),
                    # the hook sees adj_changes after optimizer.step() and before projection + clamp (:281-283)
                    f"{name}_a_clip_sum": np.clip(sa, 0, 1).astype(np.float64).sum(1),
                    f"{name}_g_sum": sg.astype(np.float64).sum(1), f"{name}_g_sqsum": (sg.astype(np.float64) ** 2).sum(1),
                    f"{name}_g_absmax": np.abs(sg).max(1),
                    f"{name}_g_negfrac": (sg < 0).mean(1),
                    f"{name}_final_sample": res["final"][samp[:, 0], samp[:, 1]],
                    f"{name}_final_sum": float(res["final"].astype(np.float64).sum()),
                    f"{name}_auc": res["auc"]})
        if name == "run":
            out.update(H_A2_sample=res["H_A2"][:64], Y_A_sample=res["Y_A"][:64])
        print(tag, name, "auc", res["auc"], "steps", ep, "seconds", round(_t.time() - t0, 1), flush=True)
        del res, sa, sg
        np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)


def gen_bench_ref64(tmp, workload="synthetic-10k-hsic", tag="bench10k_hsic", single_starts=((1, 0.5), (2, 2.0))):
    """The first step of each start of gen_bench through the reference's OWN code in float64 (run_reference_attack(f64=True):
    torch default dtype float64, inputs / weights / adj_changes as doubles): `<name>_g64ref` = its gradient at the fixture's
    packed positions, `<name>_g64ref_absmax`.  An exact-arithmetic truth for the N = 10 000 gradient that is independent of
    oracle/ (tests/golden/make_truth64.py is the oracle in float64; the two agree to ~1e-12, see the printout).
    ~45 GB and ~10 min per start on 8 cores."""
    os.chdir(tmp)
    os.makedirs("saved_data", exist_ok=True)
    for root in (os.path.dirname(os.path.dirname(OUT)), os.environ.get("MCGRA_REPO", ""), "/root/repo"):
        if root and os.path.exists(os.path.join(root, "bench.py")):
            sys.path.insert(0, root)
            break
    import bench as B
    import time as _t
    z = np.load(os.path.join(OUT, f"{tag}.npz"))
    z64 = np.load(os.path.join(OUT, f"{tag}_fp64.npz"))
    seed = int(z["seed"])
    n, f, c, hid, nl, measure, wp = B.WORKLOADS[workload]
    inp = B.make_inputs(n, f, c, hid, nl, seed)
    device = torch.device("cpu")
    victim = GCN(nfeat=f, nclass=c, nhid=hid, nlayer=nl, dropout=0.5, weight_decay=5e-4, device=device)
    with torch.no_grad():
        for l in range(nl):
            victim.gc[l].weight.copy_(torch.tensor(inp["W"][l])); victim.gc[l].bias.copy_(torch.tensor(inp["b"][l]))
        victim.linear1.weight.copy_(torch.tensor(inp["Wlin"])); victim.linear1.bias.copy_(torch.tensor(inp["blin"]))
    lab = inp["labels"]
    np.save("saved_data/cora.npy", (lab[:, None] == lab[None, :]).astype(np.float32))
    adj = torch.tensor(inp["adj"]); feats = torch.tensor(inp["features"]); labels = torch.LongTensor(lab)
    pk = z["packed_pos"]
    lr, sc0 = B.workload_lr(workload, n), B.start_scale(workload, n)
    out = dict(workload=workload, packed_pos=pk)
    runs = [("run", seed, sc0)] + [(f"one{k}", seed + off, sc0 * mul) for k, (off, mul) in enumerate(single_starts)]
    for name, sd, sc in runs:
        assert (int(z[f"{name}_a0_seed"]), float(z[f"{name}_a0_scale"])) == (sd, sc)
        t0 = _t.time()
        res = run_reference_attack(adj, feats, labels, victim, inp["idx_attack"], measure, wp, 1.0, lr, 1, "cora",
                                   (True, True, True), 1e30, a0=B.make_a0(n, sd, sc), f64=True)
        g = res["steps_g"][0]
        assert g.dtype == np.float64
        out[f"{name}_g64ref"] = g[pk]
        out[f"{name}_g64ref_absmax"] = float(np.abs(g).max())
        gmax = out[f"{name}_g64ref_absmax"]
        print(tag, name, "reference in float64: gmax", gmax, " vs the float64 oracle:",
              np.abs(g[pk] - z64[f"{name}_g64"]).max() / gmax, " the reference's fp32 run vs it:",
              np.abs(z[f"{name}_g"][0] - g[pk]).max() / gmax, "rms", np.sqrt(np.mean((z[f"{name}_g"][0] - g[pk]) ** 2)) / gmax,
              "seconds", round(_t.time() - t0, 1), flush=True)
        del res, g
        np.savez_compressed(os.path.join(OUT, f"{tag}_ref64.npz"), **out)


def check_ref64(tmp, n=300):
    """run_reference_attack(f64=True) against the float64 oracle on a small synthetic HSIC case (seconds): the two float64
    evaluations of one algorithm must agree to rounding."""
    os.chdir(tmp)
    os.makedirs("saved_data", exist_ok=True)
    root = os.path.dirname(os.path.dirname(OUT))
    sys.path.insert(0, root)
    from oracle import mcgra_oracle as O
    adj, feats, labels, victim = make_synth(n, 24, 4, 16, 2, 5)
    idx_attack = np.random.RandomState(5).permutation(n)
    lab = labels.numpy()
    np.save("saved_data/cora.npy", (lab[:, None] == lab[None, :]).astype(np.float32))
    wp = (0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)
    a0 = init_adj_changes(n, 3, 1.0 / n)
    res = run_reference_attack(adj, feats, labels, victim, idx_attack, "HSIC", wp, 1.0, 1e-4, 1, "cora", (True, True, True), 1e30,
                               a0=a0, f64=True)
    g = res["steps_g"][0]
    O.F32 = np.float64
    f8 = lambda x: np.asarray(x).astype(np.float64)
    wd = weights_of(victim)
    w = O.GCNWeights([f8(wd["W0"]), f8(wd["W1"])], [f8(wd["b0"]), f8(wd["b1"])], f8(wd["Wlin"]), f8(wd["blin"]))
    X = f8(feats.numpy())
    fadj = 1.0 / (1.0 + np.exp(-np.maximum(X @ X.T - np.eye(n), 0)))
    cfg = O.AttackConfig(measure="HSIC", weight_sup=1.0, weight_param=wp, lr=1e-4, num_edges=float("inf"))
    orc = O.PGDAttackOracle(w, X, f8(adj.numpy()), np.zeros((n, n)), fadj, lab, idx_attack, cfg)
    orc.w = w
    orc.set_adj_changes(f8(a0))
    orc.step()
    go = O.pack_tril(orc.last["G_sym"])
    print("reference(float64) vs oracle(float64): max |dg| / gmax =", np.abs(g - go).max() / np.abs(g).max(), "dtype", g.dtype)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="all")
    a = ap.parse_args()
    torch.set_num_threads(8)
    with tempfile.TemporaryDirectory() as tmp:
        if a.only in ("all", "ops"):
            gen_ops(tmp)
        if a.only in ("all", "small"):
            gen_small(tmp)
        if a.only in ("all", "cw"):
            gen_cw(tmp)
        if a.only in ("all", "cora"):
            gen_cora(tmp)
        if a.only in ("ori",):
            gen_small(tmp, only=("s48_hsic_ori", "s48_mse_ori"))
        if a.only in ("ops_kde", "kde"):
            gen_ops_kde(tmp)
        if a.only in ("kde",):
            gen_small(tmp, only=("s48_kde", "s48_kde_init", "s200_kde_init", "s48_kde_eps"))
        if a.only in ("cora_sparse",):
            gen_cora(tmp, only=("cora_hsic_sparse",))
        if a.only in ("all", "mid"):
            gen_mid(tmp)
        if a.only in ("all", "readme") or a.only.startswith("readme:"):      # README lines on brazil / usair / polblogs / AIDS
            gen_readme(tmp, only=a.only.split(":", 1)[1].split(",") if ":" in a.only else None)
        if a.only.split(":")[0] in ("readme_horizon", "readme_horizon100"):      # 35 min / 3 h on 8 cores: not part of "all"
            gen_readme_horizon(tmp, epochs=100 if a.only.split(":")[0].endswith("100") else 20,
                               only=a.only.split(":", 1)[1].split(",") if ":" in a.only else None)
        if a.only in ("citeseer",):         # ~20 min on 8 cores: not part of "all"
            gen_citeseer_gat(tmp)
        if a.only in ("bench10k",):         # ~25 min and ~20 GB on 8 cores: not part of "all"
            gen_bench(tmp)
        if a.only in ("bench10k_ref64",):   # the reference's own code in float64 on the bench's starts: ~45 GB, ~30 min
            gen_bench_ref64(tmp)
        if a.only in ("check_ref64",):
            check_ref64(tmp)
