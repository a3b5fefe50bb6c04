"""Victim classes of main.py --arch gcn | sage | gat (mc-gra_amd/models/): their torch forward must equal the unified
layer form P_l = adj (H W_l) + H Ws_l + b_l the engine and the oracle evaluate, through the weights
PGDAttack._weights extracts.  CPU only (training-side plumbing; the attack itself has no CPU path)."""
import numpy as np
import pytest
import torch

from oracle import mcgra_oracle as O


@pytest.fixture(scope="module")
def pkg():
    import mcgra_loader
    return mcgra_loader.load()


def _graph(n=40, f=12, c=3, seed=0):
    rng = np.random.RandomState(seed)
    X = (rng.rand(n, f) < 0.3).astype(np.float32)
    A = np.triu(rng.rand(n, n) < 0.15, 1)
    A = (A | A.T).astype(np.float32)
    y = rng.randint(0, c, n)
    return torch.tensor(X), torch.tensor(A), torch.tensor(y)


@pytest.mark.parametrize("arch", ["gcn", "sage", "gat"])
def test_victim_forward_is_the_unified_layer_form(pkg, arch):
    from mc_gra_amd.models.gcn import GCN, embedding_GCN
    from mc_gra_amd.models.gat import GAT, embedding_gat
    from mc_gra_amd.models.graphsage import graphsage, embedding_graphsage
    torch.manual_seed(1)
    X, A, y = _graph()
    dev = torch.device("cpu")
    if arch == "gcn":
        v = GCN(nfeat=12, nclass=3, nhid=16, nlayer=2, device=dev)
        e = embedding_GCN(nfeat=12, nhid=16, nlayer=2, device=dev); e.gc = v.gc
    elif arch == "sage":
        v = graphsage(nfeat=12, nclass=3, nhid=16, nlayer=2, device=dev)
        e = embedding_graphsage(nfeat=12, nhid=16, nlayer=2, device=dev); e.gc = v.gc
    else:
        v = GAT(nfeat=12, nclass=3, nhid=16, nlayer=2, dropout=0.5, alpha=0.1, nheads=3, device=dev)
        e = embedding_gat(nfeat=12, nclass=3, nhid=16, nlayer=2, dropout=0.5, alpha=0.1, nheads=3, device=dev)
        e.attentions = v.attentions
    idx = np.arange(20)
    v.fit(X, A, y, idx, np.arange(20, 30), train_iters=5)
    v.eval(); e.eval()
    W, b, Wlin, blin, Ws, act, head_act, emb_full = pkg.PGDAttack._weights(v, e)
    w = O.GCNWeights([x.numpy() for x in W], [x.numpy() for x in b], Wlin.numpy(), blin.numpy(),
                     [x.numpy() for x in Ws] if Ws is not None else None, act, head_act)
    T0 = X.numpy() @ w.W[0]
    S0 = X.numpy() @ w.Ws[0] if Ws is not None else None
    with torch.no_grad():
        out = v(X, A).numpy()
        em = e(X, A).numpy()
    _, H, _ = O.gcn_chain(T0, A.numpy(), w, 2, S0)
    _, logp = O.victim_head(H[-1], w)
    assert np.abs(em - H[-1]).max() < 1e-4 * max(1.0, np.abs(em).max())
    assert np.abs(out - logp).max() < 1e-4
    assert emb_full == (arch == "gat")


def test_main_builds_every_arch_parser(pkg):
    from mc_gra_amd import main as M
    for arch in ("gcn", "sage", "gat"):
        a = M.build_parser().parse_args(["--arch", arch, "--dataset", "citeseer", "--useH_A", "--useY"])
        assert a.arch == arch and a.useH_A and a.useY and not a.useY_A


# ------------------------------------------------------------------ dataset loader (caller side of the path)
DATA_ROOT = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden", "dataset")
REF_DATA = "/root/reference/MC-GRA/dataset"       # AIDS (1.9 MB of text) is read from the reference checkout when it is there


@pytest.mark.parametrize("name,root", [("brazil", DATA_ROOT), ("usair", DATA_ROOT), ("polblogs", DATA_ROOT), ("AIDS", REF_DATA),
                                       ("cora", REF_DATA), ("citeseer", REF_DATA)])
def test_dataset_loader_gives_what_the_reference_loader_gave(pkg, name, root):
    """mc-gra_amd/dataset.py on the reference's own data files (tests/golden/dataset/: the brazil / usair edge lists with
    labels, polblogs.npz; AIDS 1.9 MB, cora.npz 0.7 MB and citeseer.npz 1.4 MB stay in the reference checkout and are
    skipped where it is absent) against what the reference's Dataset + preprocess produced when the README fixtures were
    made (dataset.py:44-70, :200-300, :340-390; main.py:141-162): edges, self loops, attributes, labels, and -- with
    main.py's seeding -- the train / val / test split and idx_attack."""
    import os
    import random
    from tests import helpers as H
    from mc_gra_amd.dataset import Dataset
    from mc_gra_amd.utils import preprocess
    if not os.path.exists(os.path.join(root, name)) and not os.path.exists(os.path.join(root, name + ".npz")):
        pytest.skip(f"{name}: data files not on this machine")
    z = H.load_readme_graph(name)
    np.random.seed(15); random.seed(15); torch.manual_seed(15)          # main.py:141-143
    data = Dataset(root=root, name=name, setting='GCN')
    assert np.array_equal(data.idx_train, z["idx_train"]) and np.array_equal(data.idx_val, z["idx_val"])
    assert np.array_equal(data.idx_test, z["idx_test"])
    idx_attack = np.array(random.sample(range(data.adj.shape[0]), data.adj.shape[0]))       # main.py:155, and again :244
    idx_attack = np.array(random.sample(range(data.adj.shape[0]), data.adj.shape[0]))
    assert np.array_equal(idx_attack, z["idx_attack"])
    adj, features, labels = preprocess(data.adj, data.features, data.labels, preprocess_adj=False, onehot_feature=False)
    assert np.array_equal(adj.numpy(), z["adj"])
    assert np.array_equal(features.numpy(), z["features"])
    assert np.array_equal(labels.numpy(), z["labels"])
    assert np.array_equal(np.asarray(data.init_adj.todense()), np.zeros_like(z["adj"]))      # dataset.init_matrix (:433-437)
