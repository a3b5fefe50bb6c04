"""Diagnostic: engine vs oracle intermediates on the Cora golden, step by step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mcgra_loader
pkg = mcgra_loader.load()
from oracle import mcgra_oracle as O
from tests import helpers as H
name = sys.argv[1] if len(sys.argv) > 1 else "cora_hsic"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
z = H.load_cora(name)
w = O.GCNWeights([z["W0"], z["W1"]], [z["b0"], z["b1"]], z["Wlin"], z["blin"])
X, adj, lab = z["features"], z["adj"], z["labels"]
fadj = H.cora_feature_adj(X)
n = adj.shape[0]
cfg = O.AttackConfig(measure=str(z["measure"]), weight_sup=float(z["weight_sup"]), weight_param=tuple(z["weight_param"]), lr=float(z["lr"]), num_edges=float(z["num_edges"]))
orc = O.PGDAttackOracle(w, X, adj, np.zeros((n, n), np.float32), fadj, lab, z["idx_attack"], cfg)
eng = pkg.AttackEngine(n, [X.shape[1], 16, 16], 7, 2, cfg.measure, cfg.weight_sup, cfg.weight_param, cfg.lr, cfg.num_edges, n)
eng.set_model(w.W, w.b, w.Wlin, w.blin)
eng.set_graph(X, adj, None, fadj, lab, z["idx_attack"])
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
print("T0", rel(eng.buffer("T0").cpu().numpy(), orc.T0), "HA", rel(eng.buffer("HA").cpu().numpy(), orc.HA), "YA", rel(eng.buffer("YA").cpu().numpy(), orc.YA))
if cfg.measure == "HSIC":
    Kf = O._center_gram(fadj @ fadj.T)
    print("KFC", rel(eng.buffer("KFC").cpu().numpy(), Kf))
for t in range(steps):
    Mb = orc.M.copy()
    eng.set_adj_changes(O.pack_tril(Mb))
    sc = eng.step(want_scalars=True)
    orc.step()
    L = orc.last
    for nm, key in [("adj_norm", "adj_norm"), ("A1", "A1"), ("em", "em"), ("G_adjn", "G_adjn"), ("G_A1", "G_A1"), ("G_em", "G_em"), ("G_A", "G_A"), ("G_sym", "G_sym"), ("logp", "logp"), ("sm2", "sm2")]:
        print(t, nm, rel(eng.buffer(nm).cpu().numpy(), L[key]))
    print(t, "loss", sc["loss"], L["loss"], {k: (sc[k], L["terms"].get(k)) for k in ("c1", "c2", "c6", "c7", "c9", "c10")})
    print(t, "M", rel(eng.buffer("M").cpu().numpy(), orc.M), np.abs(eng.buffer("M").cpu().numpy() - orc.M).max())
# ---- post-loop ensemble comparison (same state: engine re-seeded from the oracle before the last step)
lab_adj = (lab[:, None] == lab[None, :]).astype(np.float32)
_, Hs, _ = O.gcn_chain(orc.T0, adj, orc.w, 2); _, YA = O.victim_head(Hs[-1], orc.w)
f_or = orc.finalize("cora", True, True, True, lab_adj, Hs[-1], YA)
f_en = eng.finalize(0, Hs[-1], YA, lab_adj).cpu().numpy()
print("final max abs diff", np.abs(f_en - f_or).max(), "AUC eng", O.metric_pool(adj, f_en, z["idx_attack"]), "AUC orc", O.metric_pool(adj, f_or, z["idx_attack"]))
d = np.abs(f_en - f_or); i, j = np.unravel_index(d.argmax(), d.shape); print("argmax", i, j, f_en[i, j], f_or[i, j])
print("M after finalize", rel(eng.buffer("M").cpu().numpy(), orc.M))
