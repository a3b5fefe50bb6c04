"""Accuracy of the two split arithmetics against fp64 (GPU): scripts/split_modes_probe.py [n ...]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcgra_loader
pkg = mcgra_loader.load()
from mc_gra_amd import engine as E

def operands(n, kind, dev):
    g = torch.Generator(device="cpu").manual_seed(n)
    if kind == "gram-adjn":       # the step's operands: centred Gram x centred normalised adjacency
        F = torch.randn(n, 16, generator=g)
        F = F - F.mean(0, keepdim=True)
        S = F @ F.T
        A = (torch.rand(n, n, generator=g) < 8.0 / n).float()
        A = torch.maximum(A, A.T); A.fill_diagonal_(1.0)
        d = A.sum(1)
        X = A / d.sqrt()[:, None] / d.sqrt()[None, :]
        sub = X.mean(0)
    elif kind == "wide-range":    # 2^20 dynamic range inside both operands
        S = torch.randn(n, n, generator=g) * torch.exp2(torch.randint(-20, 1, (n, n), generator=g).float())
        S = (S + S.T) * 0.5
        X = torch.randn(n, n, generator=g) * torch.exp2(torch.randint(-20, 1, (n, n), generator=g).float())
        sub = None
    else:                          # plain gaussians
        S = torch.randn(n, n, generator=g); S = (S + S.T) * 0.5
        X = torch.randn(n, n, generator=g)
        sub = None
    return S.to(dev), X.to(dev), (sub.to(dev) if sub is not None else None)

dev = "cuda:0"
for n in [int(x) for x in sys.argv[1:]] or [300, 1100, 4096]:
    for kind in ("gram-adjn", "gauss", "wide-range"):
        S, X, sub = operands(n, kind, dev)
        Xd = X.double() - (sub.double()[:, None] if sub is not None else 0.0)
        ref = S.double() @ Xd.T
        den = S.double().abs() @ Xd.abs().T
        f32 = (S @ (X - (sub[:, None] if sub is not None else 0.0)).T).double()
        out = {}
        out["torch fp32"] = float(((f32 - ref).abs() / den).max())
        out["bf16 x 3"] = float(((E.ssymm_split_bf16(S, X, sub).double() - ref).abs() / den).max())
        out["fp16 x 2"] = float(((E.ssymm_split_f16(S, X, sub).double() - ref).abs() / den).max())
        print(f"n={n:6d} {kind:10s} max err / (|S||B|): " + "  ".join(f"{k} {v:.2e}" for k, v in out.items()), flush=True)
