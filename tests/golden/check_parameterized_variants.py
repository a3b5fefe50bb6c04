"""Evidence for DESIGN.md section 0: the reference's gcn_parameterized.PGDAttack / gaussian_parameterized.PGDAttack
(main.py --mode gcn_attack / gaussian) cannot run as shipped -- attack() builds torch.optim.Adam([self.adj_changes])
(gcn_parameterized.py:165, gaussian_parameterized.py:*) but neither class ever defines adj_changes -- so there is no
reference behaviour to pin and no golden can be generated for them.  Needs /root/reference (this container only).

    python tests/golden/check_parameterized_variants.py
"""
import argparse
import os
import sys
import tempfile
import types

import numpy as np

np.int = int
sys.modules['torchmetrics'] = types.SimpleNamespace(AUROC=None)
import matplotlib
matplotlib.use("Agg")
sys.path.insert(0, '/root/reference/MC-GRA')
import torch
import gaussian_parameterized as GA
import gcn_parameterized as GP
from models.gcn import GCN, embedding_GCN

n = 12
os.chdir(tempfile.mkdtemp())
os.makedirs("saved_data")
np.save("saved_data/cora.npy", np.zeros((n, n), np.float32))
vm = GCN(nfeat=5, nclass=3, nhid=4, nlayer=2, dropout=0.5, device='cpu')
emb = embedding_GCN(nfeat=5, nhid=4, nlayer=2, device='cpu')
args = argparse.Namespace(max_eval=100, lr=-2, dataset="cora", eps=0, measure="MSELoss", useH_A=0, useY_A=0, useY=0)
for name, mod, kw in (("gcn_parameterized", GP, dict(features=torch.rand(n, 5))), ("gaussian_parameterized", GA, {})):
    m = mod.PGDAttack(model=vm, embedding=emb, H_A=torch.rand(n, 4), Y_A=torch.rand(n, 3), nnodes=n, device='cpu', **kw)
    try:
        m.attack(args, None, 0.01, 0, 1, (0.01, 0, 0, 0, 0, 10, 10, 0, 10, 10), torch.eye(n), 0, 0, 0, np.arange(n),
                 np.arange(n), np.arange(n), torch.zeros(n, n), torch.rand(n, 5), np.zeros((n, n), np.float32),
                 np.zeros(n, dtype=np.int64), np.arange(n), 1e9, 0, epochs=1)
        print(name, "attack ran")
    except Exception as e:                                   # noqa: BLE001
        print(f"{name}.PGDAttack.attack -> {type(e).__name__}: {e}")
