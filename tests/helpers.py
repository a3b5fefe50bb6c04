"""Shared test helpers: load golden fixtures, build oracle objects from them."""
import glob
import os

import numpy as np

os.environ.setdefault("MCGRA_KEEP_GSYM", "1")     # the parity tests read each step's mirrored gradient ("G_sym")
os.environ.setdefault("MCGRA_AB", "1")            # ... and the engine honours such switches only beside MCGRA_AB=1 (attack.hip: ab_env)

from oracle import mcgra_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def attack_cases():
    return sorted(os.path.basename(p)[len("attack_"):-4] for p in glob.glob(os.path.join(GOLDEN, "attack_*.npz")))


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"attack_{name}.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def weights_from(z):
    L = int(z["nlayer"]) if "nlayer" in z else 2
    Ws = [z[f"Ws{l}"] for l in range(L)] if "Ws0" in z else None
    return O.GCNWeights([z[f"W{l}"] for l in range(L)], [z[f"b{l}"] for l in range(L)], z["Wlin"], z["blin"], Ws,
                        str(z["act"]) if "act" in z else "relu", str(z["head_act"]) if "head_act" in z else "none")


def cfg_from(z):
    return O.AttackConfig(measure=str(z["measure"]), weight_sup=float(z["weight_sup"]),
                          weight_param=tuple(float(x) for x in z["weight_param"]), lr=float(z["lr"]),
                          num_edges=float(z["num_edges"]), eps=float(z["eps"]) if "eps" in z else 0.0,
                          emb_nlayer=int(z["emb_nlayer"]) if "emb_nlayer" in z else 2,
                          fin_layers=tuple(int(x) for x in z["fin_layers"]) if "fin_layers" in z else (1, 2))


def init_adj_changes(n, seed, scale):
    """Same generator as tests/golden/make_golden.py:init_adj_changes."""
    return (np.random.RandomState(int(seed)).rand(n * (n - 1) // 2) * float(scale)).astype(np.float32)


def a0_of(z):
    if "a0_seed" in z:
        return init_adj_changes(z["adj"].shape[0], z["a0_seed"], z["a0_scale"])
    return None


def seeded_noise(seed, t, shape):
    """Step t's noise of a seeded eps != 0 run (same generator as tests/golden/make_golden.py:seeded_noise: numpy's legacy
    RandomState, bit-reproducible across platforms by its compatibility policy)."""
    return np.random.RandomState(int(seed) + int(t)).standard_normal(shape).astype(np.float32)


def noise_of(z, t):
    """Noise the reference drew at step t, or None when eps == 0: the recorded torch.randn_like matrices (`noise`), or -- on
    large graphs -- the seeded stream the reference was handed (`noise_seed`), checked against the digest the generating
    run kept of it (sum, sum of squares, three entries: float64)."""
    if "noise" in z:
        return z["noise"][t]
    if "noise_seed" in z:
        n = len(z["labels"])
        zn = seeded_noise(z["noise_seed"], t, (n, n))
        z8 = zn.astype(np.float64)
        dig = np.array([z8.sum(), (z8 ** 2).sum(), z8[0, 0], z8[-1, -1], z8[n // 2, n // 3]])
        assert np.allclose(dig, z["noise_digest"][t], rtol=1e-12, atol=1e-9), "the seeded noise stream is not the fixture's"
        return zn
    return None


def oracle_from(z):
    n = z["adj"].shape[0]
    ori = z["ori_adj"] if "ori_adj" in z else np.zeros((n, n), np.float32)
    orc = O.PGDAttackOracle(weights_from(z), z["features"], z["adj"], ori,
                            z["feature_adj"], z["labels"], z["idx_attack"], cfg_from(z))
    if a0_of(z) is not None:
        orc.set_adj_changes(a0_of(z))
    return orc


def load_cora(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    z = {k: z[k] for k in z.files}
    nfeat = int(z["nfeat"])
    feats = np.unpackbits(z["features_bits"], axis=1)[:, :nfeat].astype(np.float32)
    n = feats.shape[0]
    adj = np.zeros((n, n), np.float32)
    e = z["adj_edges"]
    adj[e[:, 0], e[:, 1]] = 1
    adj[e[:, 1], e[:, 0]] = 1
    z["features"], z["adj"] = feats, adj
    return z


def readme_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "readme_*.npz"))
                  if not p.endswith("_fp64.npz") and not p.endswith("_graph.npz"))


def main_feature_adj(feats, dataset):
    """main.dot_product_decode (main.py:44-55): sigmoid(relu(X X^T - I)) for cora / citeseer / AIDS, relu(Xn Xn^T - I) on
    row-normalised attributes for the other datasets."""
    X = np.asarray(feats, np.float32)
    if dataset in ("cora", "citeseer", "AIDS"):
        return cora_feature_adj(X)
    nrm = np.maximum(np.sqrt((X.astype(np.float64) ** 2).sum(1, keepdims=True)), 1e-12)
    Xn = (X / nrm).astype(np.float32)
    return np.maximum(Xn @ Xn.T - np.eye(X.shape[0], dtype=np.float32), 0).astype(np.float32)


_README_GRAPHS = {}


def load_readme_graph(dataset):
    """readme_<dataset>_graph.npz of tests/golden/make_golden.py:gen_readme: the graph rebuilt from its edges and diagonal,
    attributes from bits / the identity flag / float32, feature_adj by the dataset's rule, the trained weights (cached)."""
    key = str(dataset).lower()
    if key not in _README_GRAPHS:
        g = np.load(os.path.join(GOLDEN, f"readme_{key}_graph.npz"), allow_pickle=False)
        g = {k: g[k] for k in g.files}
        n = len(g["labels"])
        adj = np.zeros((n, n), np.float32)
        e = g["adj_edges"]
        adj[e[:, 0], e[:, 1]] = 1
        adj[e[:, 1], e[:, 0]] = 1
        adj[np.arange(n), np.arange(n)] = g["adj_diag"].astype(np.float32)
        g["adj"] = adj
        if int(g["features_identity"]):
            g["features"] = np.eye(n, dtype=np.float32)
        elif "features_bits" in g:
            g["features"] = np.unpackbits(g["features_bits"], axis=1)[:, :int(g["nfeat"])].astype(np.float32)
        else:
            g["features"] = g["features_f32"]
        g["feature_adj"] = main_feature_adj(g["features"], str(g["dataset"]))
        _README_GRAPHS[key] = g
    return _README_GRAPHS[key]


def load_readme(name):
    """A README-line fixture in the layout of load_case: its dataset's graph file merged with the line's own results."""
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    z = {k: z[k] for k in z.files}
    return {**load_readme_graph(str(z["dataset"])), **z}


def tril_pos(p):
    """packed position (torch.tril_indices(n, n, -1) order, topology_attack.py:369) -> (row, col)"""
    p = np.asarray(p, dtype=np.int64)
    i = ((1.0 + np.sqrt(1.0 + 8.0 * p.astype(np.float64))) / 2.0).astype(np.int64)
    i = np.where(i * (i - 1) // 2 > p, i - 1, i)
    i = np.where((i + 1) * i // 2 <= p, i + 1, i)
    return i, p - i * (i - 1) // 2


def cora_feature_adj(feats):
    """main.dot_product_decode for cora (main.py:44-48)."""
    Z = feats @ feats.T
    Z = np.maximum(Z - np.eye(Z.shape[0], dtype=np.float32), 0)
    return (1.0 / (1.0 + np.exp(-Z.astype(np.float64)))).astype(np.float32)


def synthetic_case(n, nfeat, widths, nclass, seed, measure="HSIC", weight_param=(0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)):
    rng = np.random.RandomState(seed)
    dims = [nfeat] + list(widths)
    z = {"measure": np.array(measure), "weight_sup": np.array(1.0), "weight_param": np.array(weight_param, np.float64),
         "lr": np.array(0.01), "num_edges": np.array(1e30), "nlayer": np.array(len(widths)),
         "emb_nlayer": np.array(min(2, len(widths))), "a0_seed": np.array(7), "a0_scale": np.array(0.05)}
    for l in range(len(widths)):
        s = 1.0 / np.sqrt(dims[l + 1])
        z[f"W{l}"] = rng.uniform(-s, s, (dims[l], dims[l + 1])).astype(np.float32)
        z[f"b{l}"] = rng.uniform(0, s, dims[l + 1]).astype(np.float32)
    s = 1.0 / np.sqrt(widths[-1])
    z["Wlin"] = rng.uniform(-s, s, (nclass, widths[-1])).astype(np.float32)
    z["blin"] = rng.uniform(-s, s, nclass).astype(np.float32)
    z["features"] = (rng.rand(n, nfeat) < 0.3).astype(np.float32)
    a = (rng.rand(n, n) < 0.08).astype(np.float32)
    a = np.triu(a, 1); z["adj"] = a + a.T
    z["feature_adj"] = cora_feature_adj(z["features"])
    z["labels"] = rng.randint(0, nclass, n)
    z["idx_attack"] = rng.permutation(n)[: n - 5]
    return z


def masked_weights(z):
    """Weights of a golden / synthetic case whose embedding-layer bias kills about half of em (the bias minus the median
    of the relu output at the first step): the decode then masks pairs (S_ij <= 0), which voids the low-rank forms."""
    w = weights_from(z)
    le = cfg_from(z).emb_nlayer - 1
    probe = oracle_from(z)
    probe.step()
    w.b = [b.copy() for b in w.b]
    w.b[le] = (w.b[le] - np.quantile(probe.last["em"], 0.5, axis=0)).astype(np.float32)
    return w


# ---------------------------------------------------------------- GPU-side helpers
def engine_from(pkg, z, device="cuda:0", measure=None, weight_param=None, **kw):
    """AttackEngine (C-ABI handle) set up from a golden attack case."""
    import torch  # noqa: F401
    cfg = cfg_from(z)
    w = weights_from(z)
    dims = [w.W[0].shape[0]] + [x.shape[1] for x in w.W]
    eng = pkg.AttackEngine(z["adj"].shape[0], dims, w.Wlin.shape[0], cfg.emb_nlayer, measure or cfg.measure,
                           cfg.weight_sup, weight_param or cfg.weight_param, cfg.lr, cfg.num_edges,
                           len(z["idx_attack"]), eps=cfg.eps, device=device, act=w.act, head_act=w.head_act,
                           has_self=w.Ws is not None, fin_layers=cfg.fin_layers, **kw)
    eng.set_model(w.W, w.b, w.Wlin, w.blin, w.Ws)
    eng.set_graph(z["features"], z["adj"], z["ori_adj"] if "ori_adj" in z else None, z["feature_adj"], z["labels"], z["idx_attack"])
    if a0_of(z) is not None:
        eng.set_adj_changes(a0_of(z))
    return eng


class _Layer:
    def __init__(self, W, b):
        import torch
        self.weight = torch.tensor(np.asarray(W, np.float32))
        self.bias = torch.tensor(np.asarray(b, np.float32))


class _Lin:
    def __init__(self, W, b):
        import torch
        self.weight = torch.tensor(np.asarray(W, np.float32))
        self.bias = torch.tensor(np.asarray(b, np.float32))


class FakeGCN:
    """Duck-typed stand-in for models/gcn.py GCN / embedding_GCN: the attributes
    PGDAttack reads (gc[l].weight/.bias, linear1, nclass, nfeat, hidden_sizes, nlayer)."""

    def __init__(self, w: O.GCNWeights):
        self.gc = [_Layer(W, b) for W, b in zip(w.W, w.b)]
        self.linear1 = _Lin(w.Wlin, w.blin)
        self.nclass = w.Wlin.shape[0]
        self.nfeat = w.W[0].shape[0]
        self.hidden_sizes = [w.W[0].shape[1]]
        self.nlayer = 2

    def eval(self):
        return self

    def set_layers(self, n):
        self.nlayer = n


class _Att:
    def __init__(self, W):
        import torch
        self.W = torch.tensor(np.asarray(W, np.float32))


class FakeGAT:
    """Duck-typed models/gat.py GAT / embedding_gat: attentions[l][head].W, out_att, nlayer."""

    def __init__(self, w: O.GCNWeights, nhid=16):
        self.attentions = [[_Att(W[:, k:k + nhid]) for k in range(0, W.shape[1], nhid)] for W in w.W]
        self.out_att = _Lin(w.Wlin, w.blin)
        self.nclass, self.nfeat, self.hidden_sizes, self.nlayer = w.Wlin.shape[0], w.W[0].shape[0], [nhid], len(w.W)

    def eval(self):
        return self

    def set_layers(self, n):
        self.nlayer = n


class _SageLayer:
    def __init__(self, Ws, Wn):
        import torch
        self.weight = torch.tensor(np.vstack([Ws, Wn]).astype(np.float32))      # graphsage.py:20, [2*in, out]
        self.bias = None


class FakeSAGE:
    """Duck-typed models/graphsage.py graphsage / embedding_graphsage."""

    def __init__(self, w: O.GCNWeights):
        self.gc = [_SageLayer(a, b) for a, b in zip(w.Ws, w.W)]
        self.linear1 = _Lin(w.Wlin, w.blin)
        self.linear1.bias = None
        self.nclass, self.nfeat, self.hidden_sizes, self.nlayer = w.Wlin.shape[0], w.W[0].shape[0], [w.W[0].shape[1]], 2

    def eval(self):
        return self

    def set_layers(self, n):
        self.nlayer = n
