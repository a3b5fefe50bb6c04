"""fp32 accuracy of the skinny products of a low-rank step evaluated from M directly (adj_norm = R (M + I) R and
Xc = adj_norm - 1 mean^T never materialised) against the evaluation on a stored Xc, both measured against float64.
Inputs: the bench generator at a reduced N, a seeded start, one forward to get Zn."""
import sys
import numpy as np
sys.path[:0] = ['/root/repo']
import bench as B
from oracle import mcgra_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
f, c, hid, nl = 128, 7, 16, 2
inp = B.make_inputs(n, f, c, hid, nl, 0)
a0 = B.make_a0(n, 0, 0.05)      # the dense start of rounds 1-2
M = O.unpack_sym(a0, n).astype(np.float32)
w = O.GCNWeights(inp["W"], inp["b"], inp["Wlin"], inp["blin"])
T0 = inp["features"] @ inp["W"][0]
_, He, _ = O.gcn_chain(T0, M, w, 2)
em = He[-1]
_, S, Zn, nrm = O.dot_product_decode_dense(em)
h = Zn.shape[1]
print("masked pairs:", int(((S <= 0) & ~np.eye(n, dtype=bool)).sum()), " row spread of Zn:", float(np.abs(Zn - Zn.mean(0)).max()))

def run(dt):
    Md = M.astype(dt)
    d = 1 + Md.sum(1); r = d ** dt(-0.5)
    An = (r[:, None] * (Md + np.eye(n, dtype=dt))) * r[None, :]
    mean = (An.sum(0, dtype=np.float64) / n).astype(dt)
    Xc = An - mean[None, :]
    Z = Zn.astype(dt)
    delta = (Z * Z).sum(1)
    zbar = Z.mean(0, dtype=np.float64).astype(dt)
    U = Z - zbar
    V = np.concatenate([U, delta[:, None] * Z, (delta * delta)[:, None]], 1)
    # (a) stored Xc
    Ta = Xc.T @ V
    Wa = Ta[:, :2 * h]
    Qa = Xc @ Wa
    # (a') stored Xc, right-hand sides column-centred
    vbar0 = V.mean(0, dtype=np.float64).astype(dt)
    Td = Xc.T @ (V - vbar0)
    Qd = Xc @ Td[:, :2 * h]
    # (b) from M: Xc^T V = An Vc - mean (1^T Vc), Vc column-centred (Xc^T 1 = 0 exactly)
    vbar = V.mean(0, dtype=np.float64).astype(dt)
    Vc = V - vbar
    Y = Md @ (r[:, None] * Vc)
    Tb = r[:, None] * (Y + r[:, None] * Vc) - mean[:, None] * Vc.sum(0, dtype=np.float64).astype(dt)[None, :]
    Wb = Tb[:, :2 * h]
    # Xc W = An Wc - 1 (mean^T Wc) + n (mean - meanbar) wbar^T, Wc column-centred
    wbar = Wb.mean(0, dtype=np.float64).astype(dt)
    Wc = Wb - wbar
    Y2 = Md @ (r[:, None] * Wc)
    Qb = (r[:, None] * (Y2 + r[:, None] * Wc) - (mean.astype(np.float64) @ Wc.astype(np.float64)).astype(dt)[None, :]
          + dt(n) * (mean - mean.mean(dtype=np.float64).astype(dt))[:, None] * wbar[None, :])
    # (c) from M, plain (no centring of the right-hand sides)
    Y = Md @ (r[:, None] * V)
    Tc = r[:, None] * (Y + r[:, None] * V) - mean[:, None] * V.sum(0, dtype=np.float64).astype(dt)[None, :]
    Wcp = Tc[:, :2 * h]
    Y2 = Md @ (r[:, None] * Wcp)
    Qc = r[:, None] * (Y2 + r[:, None] * Wcp) - (mean.astype(np.float64) @ Wcp.astype(np.float64)).astype(dt)[None, :]
    return dict(Ta=Ta, Tb=Tb, Tc=Tc, Td=Td, Qa=Qa, Qb=Qb, Qc=Qc, Qd=Qd)

r64 = run(np.float64)
r32 = run(np.float32)
for k in ("Ta", "Td", "Tb", "Tc"):
    for cols, nm in ((slice(0, h), "W"), (slice(h, 2 * h), "W2"), (slice(2 * h, 2 * h + 1), "t3")):
        ref = r64["Ta"][:, cols]
        print(k, nm, "max err / max|ref| =", float(np.abs(r32[k][:, cols] - ref).max() / max(np.abs(ref).max(), 1e-300)))
for k in ("Qa", "Qd", "Qb", "Qc"):
    for cols, nm in ((slice(0, h), "Q"), (slice(h, 2 * h), "Q2")):
        ref = r64["Qa"][:, cols]
        print(k, nm, "max err / max|ref| =", float(np.abs(r32[k][:, cols] - ref).max() / np.abs(ref).max()))
print("fp64 identities: |Tb-Ta|", float(np.abs(r64["Tb"] - r64["Ta"]).max() / np.abs(r64["Ta"]).max()),
      " |Qb-Qa|", float(np.abs(r64["Qb"] - r64["Qa"]).max() / np.abs(r64["Qa"]).max()),
      " |Qc-Qa|", float(np.abs(r64["Qc"] - r64["Qa"]).max() / np.abs(r64["Qa"]).max()))
