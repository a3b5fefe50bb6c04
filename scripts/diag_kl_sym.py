"""Where does the fused KL step break the bitwise symmetry of M / G_sym?  (round 6 debugging aid)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from oracle import mcgra_oracle as O
import mcgra_loader
pkg = mcgra_loader.load()
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1100
mode = sys.argv[2] if len(sys.argv) > 2 else "free"
z = H.synthetic_case(n, 11, (16, 16), 4, seed=n, measure="KL")
e = H.engine_from(pkg, z)
orc = H.oracle_from(z) if mode == "tf" else None
for t in range(3):
    e.step(want_scalars=True)
    if orc: orc.step()
    for name in ("G_sym", "M"):
        A = e.buffer(name)
        bad = (A != A.T).nonzero().cpu().numpy()
        print(t, name, "asymmetric entries:", len(bad), "nan:", int(torch.isnan(A).sum()))
        if len(bad):
            i, j = bad[:, 0], bad[:, 1]
            print("  same 64-tile:", int(((i // 64) == (j // 64)).sum()), "of", len(bad), " rows", i.min(), i.max(), "cols", j.min(), j.max())
            d = (A - A.T)[i, j].abs().cpu().numpy()
            print("  max |diff|", d.max(), "rel to max", d.max() / float(A.abs().max()))
            print("  first:", bad[:8].tolist())
    e.monitor()
    if orc: e.set_adj_changes(O.pack_tril(orc.M))
