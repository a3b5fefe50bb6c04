"""CPU oracle for the MC-GRA adjacency-optimisation hot path.

TEST INFRASTRUCTURE ONLY.  This file is a numpy (float32) restatement of the
reference algorithm with a hand-derived backward pass.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / reported baseline: the shipped path in
``mc-gra_amd/`` never imports anything under ``oracle/`` and fails loudly when
the HIP library is missing.

Parity pinning: ``tests/golden/make_golden.py`` imports the reference itself
(``/root/reference/MC-GRA``: utils.py, models/gcn.py, topology_attack.py) in
the build container, runs ``PGDAttack.attack`` on CPU with torch autograd and
stores inputs / per-step ``adj_changes`` / final ``modified_adj`` / AUC as
fixtures; ``tests/test_oracle_golden.py`` checks this file against them.

Every function cites the reference lines it follows (paths relative to
``/root/reference/MC-GRA``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32

# utils.py:1100-1111
ALIGN_PARAMETER_CORA = {
    "c1": 100, "c2": 1000, "c3": 100, "c4": 10, "c5": 10,
    "c6": 10, "c7": 10, "c8": 0.01, "c9": 1, "c10": 1,
}

# "KDE": utils.MutualInformation (utils.py:980-1049) asks for device='cuda:0' in ONE torch.linspace call (utils.py:990-991);
# tests/golden/make_golden.py drops that keyword (no arithmetic changes) to run the reference's CPU path with it
MEASURES = ("HSIC", "MSELoss", "KL", "CKA", "DP", "KDE")


# --------------------------------------------------------------------------
# packed lower-triangle <-> dense (torch.tril_indices(n, n, offset=-1) order)
# --------------------------------------------------------------------------
def tril_indices(n: int) -> Tuple[np.ndarray, np.ndarray]:
    """Row-major strict lower triangle, the order of torch.tril_indices
    (topology_attack.py:372-374)."""
    return np.tril_indices(n, -1)


def unpack_sym(a: np.ndarray, n: int) -> np.ndarray:
    """m[tril] = a ; m = m + m.T   (topology_attack.py:371-375)."""
    m = np.zeros((n, n), dtype=F32)
    r, c = tril_indices(n)
    m[r, c] = a
    return m + m.T


def pack_tril(m: np.ndarray) -> np.ndarray:
    r, c = tril_indices(m.shape[0])
    return np.ascontiguousarray(m[r, c], dtype=F32)


def get_modified_adj(a: np.ndarray, ori_adj: np.ndarray) -> np.ndarray:
    """PGDAttack.get_modified_adj (topology_attack.py:365-379)."""
    n = ori_adj.shape[0]
    comp = np.ones((n, n), dtype=F32) - np.eye(n, dtype=F32)
    return comp * unpack_sym(a, n) + ori_adj.astype(F32)


def adding_noise(modified_adj: np.ndarray, eps: float, noise: Optional[np.ndarray]):
    """PGDAttack.adding_noise (topology_attack.py:474-478).  ``noise`` stands
    for torch.randn_like(modified_adj); returns (clamped, gate) where gate is
    the clamp pass-through mask of torch.clamp's backward (inclusive)."""
    pre = modified_adj
    if noise is not None and eps != 0:
        pre = pre + noise.astype(F32) * F32(eps)
    gate = (pre >= 0) & (pre <= 1)
    return np.clip(pre, 0, 1).astype(F32), gate


# --------------------------------------------------------------------------
# utils.py
# --------------------------------------------------------------------------
def normalize_adj_tensor(adj: np.ndarray):
    """utils.normalize_adj_tensor, dense branch (utils.py:211-230).
    Returns (adj_norm, d, r)."""
    n = adj.shape[0]
    mx = adj.astype(F32) + np.eye(n, dtype=F32)
    d = mx.sum(1, dtype=F32)
    with np.errstate(divide="ignore"):
        r = np.power(d, F32(-0.5)).astype(F32)
    r[np.isinf(r)] = 0.0
    return (r[:, None] * mx * r[None, :]).astype(F32), d, r


def accuracy(output: np.ndarray, labels: np.ndarray) -> float:
    """utils.accuracy (utils.py:286-308)."""
    return float((output.argmax(1) == labels).astype(np.float64).sum() / len(labels))


def _center_gram(K: np.ndarray) -> np.ndarray:
    """CudaCKA.centering: H K H with H = I - 11^T/n (utils.py:1060-1065),
    evaluated as row/column mean subtraction (same matrix, O(n^2))."""
    rm = K.mean(1, keepdims=True, dtype=np.float64)
    cm = K.mean(0, keepdims=True, dtype=np.float64)
    tm = K.mean(dtype=np.float64)
    return (K - rm - cm + tm).astype(F32)


def linear_hsic(X: np.ndarray, Y: np.ndarray) -> F32:
    """CudaCKA.linear_HSIC (utils.py:1085-1089)."""
    Kx = _center_gram(X @ X.T)
    Ky = _center_gram(Y @ Y.T)
    return F32((Kx.astype(np.float64) * Ky).sum())


def linear_hsic_grads(X, Y, need_x=True, need_y=True):
    """value and d/dX, d/dY of linear_HSIC.  d/dKy = H Kxc H = Kxc, so
    d/dY = (Kxc + Kxc^T) Y = 2 Kxc Y (Gram matrices are symmetric)."""
    Kxc = _center_gram(X @ X.T)
    Kyc = _center_gram(Y @ Y.T)
    val = F32((Kxc.astype(np.float64) * Kyc).sum())
    gX = (F32(2) * (Kyc @ X)).astype(F32) if need_x else None
    gY = (F32(2) * (Kxc @ Y)).astype(F32) if need_y else None
    return val, gX, gY


def linear_cka(X, Y) -> F32:
    """CudaCKA.linear_CKA (utils.py:1091-1096)."""
    h = linear_hsic(X, Y)
    return F32(h / (np.sqrt(linear_hsic(X, X)) * np.sqrt(linear_hsic(Y, Y))))


def linear_cka_grads(X, Y, need_x=True, need_y=True):
    Kxc = _center_gram(X @ X.T)
    Kyc = _center_gram(Y @ Y.T)
    hxy = (Kxc.astype(np.float64) * Kyc).sum()
    hxx = (Kxc.astype(np.float64) * Kxc).sum()
    hyy = (Kyc.astype(np.float64) * Kyc).sum()
    den = math.sqrt(hxx) * math.sqrt(hyy)
    if den == 0:
        # 0/0: one operand has identical rows (e.g. em at adj_changes == 0).  The
        # reference evaluates H K H by fp32 matmuls, gets rounding noise instead
        # of exact zeros and returns a finite noise-driven value; that is not
        # reproducible by any other evaluation order.  Defined here as 0 with
        # zero gradient (documented divergence, DESIGN.md "degenerate CKA").
        z = F32(0)
        return z, (np.zeros_like(X) if need_x else None), (np.zeros_like(Y) if need_y else None)
    val = F32(hxy / den)
    gX = gY = None
    if need_x:  # d hxy/dX = 2 Kyc X ; d hxx/dX = 4 Kxc X
        gX = ((2.0 / den) * (Kyc @ X) - (hxy / den) * (0.5 / hxx) * 4.0 * (Kxc @ X)).astype(F32)
    if need_y:
        gY = ((2.0 / den) * (Kxc @ Y) - (hxy / den) * (0.5 / hyy) * 4.0 * (Kyc @ Y)).astype(F32)
    return val, gX, gY


def hsic_distmat(X):
    """hsic.distmat (hsic.py:20-27)."""
    X = X.astype(F32)
    r = (X * X).sum(1, dtype=F32)
    return (r[:, None] - F32(2) * (X @ X.T) + r[None, :]).astype(F32)


def hsic_kernelmat(X, sigma):
    """hsic.kernelmat with a given sigma (hsic.py:30-47): exp(-D / (2 sigma^2)) @ H."""
    K = np.exp(-hsic_distmat(X) / F32(2.0 * sigma * sigma)).astype(F32)
    return (K - K.mean(1, keepdims=True, dtype=np.float64)).astype(F32)          # K @ (I - 11^T/m)


def hsic_regular(x, y, sigma) -> F32:
    """hsic.hsic_regular (hsic.py:117-124)."""
    return F32((hsic_kernelmat(x, sigma).astype(np.float64) * hsic_kernelmat(y, sigma).T).mean())


def hsic_normalized(x, y, sigma) -> F32:
    """hsic.hsic_normalized (hsic.py:127-135)."""
    return F32(hsic_regular(x, y, sigma) / (np.sqrt(hsic_regular(x, x, sigma)) * np.sqrt(hsic_regular(y, y, sigma))))


def hsic_sigma_estimation(X, Y) -> float:
    """hsic.sigma_estimation (hsic.py:5-17)."""
    D = hsic_distmat(np.concatenate([X, Y]))
    tri = D[np.tril_indices(D.shape[0], -1)]
    med = np.median(tri)
    if med <= 0:
        med = np.mean(tri)
    if med < 1e-2:
        med = 1e-2
    return float(med)


def hsic_regular_auto(x, y) -> F32:
    """hsic.hsic_regular with sigma=None: kernelmat estimates one sigma per operand (hsic.py:39-41)."""
    kx, ky = hsic_kernelmat(x, hsic_sigma_estimation(x, x)), hsic_kernelmat(y, hsic_sigma_estimation(y, y))
    return F32((kx.astype(np.float64) * ky.T).mean())


def hsic_normalized_auto(x, y) -> F32:
    sx, sy = hsic_sigma_estimation(x, x), hsic_sigma_estimation(y, y)
    reg = lambda a, sa, b, sb: (hsic_kernelmat(a, sa).astype(np.float64) * hsic_kernelmat(b, sb).T).mean()
    return F32(reg(x, sx, y, sy) / (np.sqrt(reg(x, sx, x, sx)) * np.sqrt(reg(y, sy, y, sy))))


def hsic_distcorr(X, sigma=1.0) -> F32:
    """hsic.distcorr (hsic.py:50-53)."""
    return F32(np.exp(-hsic_distmat(X).astype(np.float64) / (2.0 * sigma * sigma)).mean())


def hsic_mmd(x, y, sigma=None) -> F32:
    """hsic.mmd (hsic.py:68-89)."""
    if sigma:
        sx = sy = sxy = sigma
    else:
        sx, sy, sxy = hsic_sigma_estimation(x, x), hsic_sigma_estimation(y, y), hsic_sigma_estimation(x, y)
    kx = np.exp(-hsic_distmat(x).astype(np.float64) / (2.0 * sx * sx))
    ky = np.exp(-hsic_distmat(y).astype(np.float64) / (2.0 * sy * sy))
    dxy = hsic_distmat(np.concatenate([x, y]))[: x.shape[0], x.shape[0]:]
    kxy = np.exp(-dxy.astype(np.float64) / (1.0 * sxy * sxy))
    return F32(kx.mean() + ky.mean() - 2 * kxy.mean())


def hsic_mmd_pxpy_pxy(x, y, sigma=None) -> F32:
    """hsic.mmd_pxpy_pxy (hsic.py:92-114)."""
    if sigma:
        sx = sy = sigma
    else:
        sx, sy = hsic_sigma_estimation(x, x), hsic_sigma_estimation(y, y)
    kx = np.exp(-hsic_distmat(x).astype(np.float64) / (2.0 * sx * sx))
    ky = np.exp(-hsic_distmat(y).astype(np.float64) / (2.0 * sy * sy))
    return F32((kx * ky).mean() - 2 * (kx.mean(0) * ky.mean(0)).mean() + kx.mean() * ky.mean())


def hsic_normalized_cca(x, y, sigma=None, dtype=None):
    """hsic.hsic_normalized_cca (hsic.py:138-151): sum(Rx o Ry^T), R = Kc inv(Kc + 1e-5 m I).  dtype=np.float64 gives
    the exact value the fp32 reference approximates (its two inverses are ill-conditioned)."""
    T = dtype or F32
    if sigma:
        sx = sy = sigma
    else:
        sx, sy = hsic_sigma_estimation(x, x), hsic_sigma_estimation(y, y)
    m = x.shape[0]

    def R(X, s_):
        X = X.astype(T)
        r = (X * X).sum(1)
        K = np.exp(-(r[:, None] - T(2) * (X @ X.T) + r[None, :]) / T(2.0 * s_ * s_)).astype(T)
        Kc = (K @ (np.eye(m, dtype=T) - np.ones((m, m), T) / T(m))).astype(T)
        return (Kc @ np.linalg.inv(Kc + T(1e-5 * m) * np.eye(m, dtype=T)).astype(T)).astype(T)

    return T((R(x, sx).astype(np.float64) * R(y, sy).T).sum())


def mse_loss(X, Y) -> F32:
    """torch.nn.MSELoss()(X, Y) (topology_attack.py:194-195)."""
    return F32(np.mean((X.astype(np.float64) - Y) ** 2))


def mse_grads(X, Y, need_x=True, need_y=True):
    diff = (X - Y).astype(F32)
    val = F32(np.mean(diff.astype(np.float64) ** 2))
    s = F32(2.0 / diff.size)
    return val, (s * diff if need_x else None), (-s * diff if need_y else None)


def _softmax(z):
    z = z - z.max(1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(1, keepdims=True)).astype(F32)


def _log_softmax(z):
    z = z - z.max(1, keepdims=True)
    return (z - np.log(np.exp(z).sum(1, keepdims=True))).astype(F32)


def calc_kl(X, Y) -> F32:
    """PGDAttack.calc_kl (topology_attack.py:483-487): implicit dim=1 for 2-D."""
    xs = _softmax(X)
    yl = _log_softmax(Y)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.where(xs > 0, xs * (np.log(xs) - yl), 0.0)
    return F32(t.sum(dtype=np.float64) / X.shape[0])


def kl_grads(X, Y, need_x=True, need_y=True):
    xs = _softmax(X)
    yl = _log_softmax(Y)
    b = F32(X.shape[0])
    with np.errstate(divide="ignore", invalid="ignore"):
        lx = np.where(xs > 0, np.log(xs), 0.0).astype(F32)
    val = F32(np.where(xs > 0, xs * (lx - yl), 0.0).sum(dtype=np.float64) / X.shape[0])
    gX = gY = None
    if need_y:  # d/dY = (softmax(Y) - softmax(X)) / batch
        gY = ((np.exp(yl) - xs) / b).astype(F32)
    if need_x:  # d/dxs = (log xs + 1 - yl)/b, then softmax backward
        gxs = (lx + F32(1) - yl) / b
        gX = (xs * (gxs - (gxs * xs).sum(1, keepdims=True))).astype(F32)
    return val, gX, gY


def dot_product(X, Y) -> F32:
    """PGDAttack.dot_product (topology_attack.py:480-481): Frobenius norm of Y^T X."""
    P = Y.T @ X
    return F32(np.sqrt((P.astype(np.float64) ** 2).sum()))


def dp_grads(X, Y, need_x=True, need_y=True):
    P = (Y.T @ X).astype(F32)
    val = F32(np.sqrt((P.astype(np.float64) ** 2).sum()))
    if val == 0:
        z = np.zeros_like(X)
        return val, (z if need_x else None), (z.copy() if need_y else None)
    gX = ((Y @ P) / val).astype(F32) if need_x else None
    gY = ((X @ P.T) / val).astype(F32) if need_y else None
    return val, gX, gY


KDE_SIGMA = 2 * 0.4 ** 2      # MutualInformation(sigma=0.4): self.sigma = 2 * sigma ** 2 (utils.py:985)
KDE_EPS = 1e-10               # self.epsilon (utils.py:988)


def kde_bins(num_bins: int) -> np.ndarray:
    """torch.linspace(0, num_bins, num_bins).float() (utils.py:990-991) as torch evaluates it in float32: step =
    num_bins / (num_bins - 1), b_j = step * j in the lower half and fma(-step, num_bins - 1 - j, num_bins) in the upper."""
    f4 = np.float32        # the bins are `.float()` whatever dtype the operands have (also in a float64 evaluation)
    if num_bins == 1:
        return np.zeros(1, f4)
    step = np.float64(f4(f4(num_bins) / f4(num_bins - 1)))
    j = np.arange(num_bins)
    lo = (step * j).astype(f4)
    hi = (np.float64(num_bins) - step * (num_bins - 1 - j)).astype(f4)      # one rounding: torch's kernel uses a fused multiply-add
    return np.where(j < num_bins // 2, lo, hi).astype(f4)


def kde_kernel_values(V: np.ndarray) -> np.ndarray:
    """marginalPdf's kernel_values (utils.py:995-996) for a 2-D input [m, c] with num_bins == c, as every call site has it
    (topology_attack.py:199-201: num_bins = feature_adj.shape[0] on N x N operands; :244-246, :261-263: the operand's
    width): `values - bins.unsqueeze(0).unsqueeze(0)` broadcasts the bins over the LAST axis, so entry (i, j) is
    compared with bin j only: k_ij = exp(-0.5 ((V_ij - b_j) / sigma)^2)."""
    b = kde_bins(V.shape[1])
    res = (V.astype(F32) - b[None, :]).astype(F32)
    t = (res / F32(KDE_SIGMA)).astype(F32)
    with np.errstate(under="ignore"):
        return np.exp(F32(-0.5) * t * t).astype(F32)


def kde_mi_grads(X, Y, need_x=True, need_y=True):
    """MutualInformation(sigma=0.4, num_bins=X.shape[1], normalize=True)(X, Y)[0] (utils.py:1014-1049) and its gradients.
    pdf = mean_i k / (sum + eps) (utils.py:998-1000); joint = k1^T k2 / (sum + eps) (:1004-1010); entropies with log2(p + eps)
    (:1033-1036); 2 (H1 + H2 - H12) / (H1 + H2) (:1038-1041).  Columns whose kernel values are all exactly zero in float32
    (bins far from every value: all but the first ~6 on N x N operands with values in [0, 1]) add nothing to any sum and are
    skipped in the joint -- the same numbers, not an approximation."""
    k1, k2 = kde_kernel_values(X), kde_kernel_values(Y)
    m = X.shape[0]
    ln2 = math.log(2.0)

    def marginal(k):
        q = k.mean(0, dtype=np.float64)
        nrm = q.sum() + KDE_EPS
        p = q / nrm
        H = -(p * np.log2(p + KDE_EPS)).sum()
        return p, nrm, H

    p1, n1, H1 = marginal(k1)
    p2, n2, H2 = marginal(k2)
    c1, c2 = np.flatnonzero(k1.any(0)), np.flatnonzero(k2.any(0))
    J = k1[:, c1].astype(np.float64).T @ k2[:, c2].astype(np.float64)
    nJ = J.sum() + KDE_EPS
    P = J / nJ
    H12 = -(P * np.log2(P + KDE_EPS)).sum()
    S = H1 + H2
    val = F32(2.0 * (S - H12) / S)
    # out = 2 - 2 H12 / S
    g_H12, g_H = -2.0 / S, 2.0 * H12 / (S * S)
    dH = lambda p: -(np.log2(p + KDE_EPS) + p / ((p + KDE_EPS) * ln2))
    gP = g_H12 * dH(P)
    gJ = (gP - (gP * P).sum()) / nJ

    def back(k, V, p, nrm, cols, gk_joint):
        gp = g_H * dH(p)
        gq = (gp - (gp * p).sum()) / nrm
        gk = np.repeat((gq / m)[None, :], m, 0)
        gk[:, cols] += gk_joint
        b = kde_bins(V.shape[1]).astype(np.float64)
        return (gk * k * (-(V.astype(np.float64) - b[None, :]) / (KDE_SIGMA * KDE_SIGMA))).astype(F32)

    gX = back(k1, X, p1, n1, c1, k2[:, c2].astype(np.float64) @ gJ.T) if need_x else None
    gY = back(k2, Y, p2, n2, c2, k1[:, c1].astype(np.float64) @ gJ) if need_y else None
    return val, gX, gY


_CALC = {
    "HSIC": linear_hsic_grads, "MSELoss": mse_grads, "KL": kl_grads,
    "CKA": linear_cka_grads, "DP": dp_grads, "KDE": kde_mi_grads,
}


def info_entropy(prob: np.ndarray) -> F32:
    """Info_entropy (topology_attack.py:44-47)."""
    q = np.clip(prob, F32(1e-4), F32(1 - 1e-4)).astype(F32)
    return F32(-np.mean((q * np.log2(q)).astype(np.float64)))


def info_entropy_grad(prob: np.ndarray):
    lo, hi = F32(1e-4), F32(1 - 1e-4)
    q = np.clip(prob, lo, hi).astype(F32)
    val = F32(-np.mean((q * np.log2(q)).astype(np.float64)))
    mask = (prob >= lo) & (prob <= hi)
    g = -(np.log2(q) + F32(1.0 / math.log(2.0))) / F32(prob.size)
    return val, np.where(mask, g, 0).astype(F32)


# --------------------------------------------------------------------------
# models/gcn.py
# --------------------------------------------------------------------------
@dataclass
class GCNWeights:
    """Weights of a victim in the unified layer form
        P_l = adj @ (H_{l-1} W_l) + H_{l-1} Ws_l + b_l ,  H_l = act(P_l)
    * models/gcn.py GCN: gc[l].weight (in,out), gc[l].bias, relu, no Ws; head linear1.
    * models/gat.py GAT: the attention product is overwritten by ``torch.matmul(adj, h)`` (gat.py:44-45), so a
      layer is elu(adj @ (x [W_1|...|W_heads])) without bias; head elu(out_att(x)) (gat.py:206).
    * models/graphsage.py: [x | adj@x] @ weight (graphsage.py:43-45) = x W_top + adj @ (x W_bot): Ws = W_top."""
    W: List[np.ndarray]
    b: List[np.ndarray]
    Wlin: np.ndarray
    blin: np.ndarray
    Ws: Optional[List[np.ndarray]] = None
    act: str = "relu"
    head_act: str = "none"

    def f32(self):
        return GCNWeights([w.astype(F32) for w in self.W], [x.astype(F32) for x in self.b],
                          self.Wlin.astype(F32), self.blin.astype(F32),
                          None if self.Ws is None else [w.astype(F32) for w in self.Ws], self.act, self.head_act)


def _act(p, kind):
    if kind == "relu":
        return np.maximum(p, 0).astype(F32)
    return np.where(p > 0, p, np.expm1(np.minimum(p, 0))).astype(F32)        # F.elu


def _act_grad(p, kind):
    if kind == "relu":
        return (p > 0).astype(F32)
    return np.where(p > 0, 1.0, np.exp(np.minimum(p, 0))).astype(F32)


def gcn_chain(T0: np.ndarray, adj: np.ndarray, w: GCNWeights, nlayer: int, S0: Optional[np.ndarray] = None):
    """x = act(adj @ (x @ W_l) + x @ Ws_l + b_l) for l < nlayer
    (GraphConvolution.forward models/gcn.py:35-46; embedding_GCN.forward :71-76;
    GCN.forward :164-172 in eval mode, dropout off; gat.py:36-50; graphsage.py:37-50).
    T0 = X @ W_0 and S0 = X @ Ws_0 are passed in because they do not depend on
    the adjacency.  Returns lists P (pre-act), H (post-act), T (H_l @ W_{l+1};
    T[0] = T0)."""
    P, H, T = [], [], [T0]
    S = S0
    for l in range(nlayer):
        p = (adj @ T[l] + w.b[l][None, :]).astype(F32)
        if S is not None:
            p = (p + S).astype(F32)
        h = _act(p, w.act)
        P.append(p)
        H.append(h)
        if l + 1 < nlayer:
            T.append((h @ w.W[l + 1]).astype(F32))
            S = (h @ w.Ws[l + 1]).astype(F32) if w.Ws is not None else None
    return P, H, T


def gcn_chain_backward(gH_last: np.ndarray, adj: np.ndarray, P, T, w: GCNWeights, nlayer: int):
    """Backward of gcn_chain w.r.t. adj.  Returns g_adj (dense)."""
    g_adj = np.zeros_like(adj, dtype=F32)
    gH = gH_last
    for l in range(nlayer - 1, -1, -1):
        gP = (gH * _act_grad(P[l], w.act)).astype(F32)
        g_adj += gP @ T[l].T
        if l > 0:
            gT = adj.T @ gP
            gH = (gT @ w.W[l].T).astype(F32)
            if w.Ws is not None:
                gH = (gH + gP @ w.Ws[l].T).astype(F32)
    return g_adj


def victim_head(H_last: np.ndarray, w: GCNWeights):
    """linear1 + log_softmax (models/gcn.py:173-174); GAT: elu(out_att(x)) first (gat.py:206-207).
    Returns (Z fed to the softmax, log_softmax(Z))."""
    Zl = (H_last @ w.Wlin.T + w.blin[None, :]).astype(F32)
    Z = _act(Zl, "elu") if w.head_act == "elu" else Zl
    return Z, _log_softmax(Z)


def victim_head_backward(G_Z: np.ndarray, H_last: np.ndarray, w: GCNWeights):
    """d/dH_last given d/dZ (Z = the softmax input)."""
    if w.head_act == "elu":
        Zl = (H_last @ w.Wlin.T + w.blin[None, :]).astype(F32)
        G_Z = (G_Z * _act_grad(Zl, "elu")).astype(F32)
    return (G_Z @ w.Wlin).astype(F32)


def dot_product_decode_dense(Z: np.ndarray):
    """PGDAttack.dot_product_decode without the tril gather
    (topology_attack.py:414-419): relu(normalize(Z) normalize(Z)^T).
    Returns (relu(S), S, Zn, nrm)."""
    nrm = np.sqrt((Z.astype(F32) ** 2).sum(1, dtype=F32)).astype(F32)
    den = np.maximum(nrm, F32(1e-12))
    Zn = (Z / den[:, None]).astype(F32)
    S = (Zn @ Zn.T).astype(F32)
    return np.maximum(S, 0), S, Zn, nrm


def sym_from_lower(R: np.ndarray) -> np.ndarray:
    """m[tril] = R[tril]; m + m.T  (topology_attack.py:387-391): the strict
    lower triangle mirrored, zero diagonal."""
    L = np.tril(R, -1)
    return (L + L.T).astype(F32)


def dot_product_decode2(Z: np.ndarray, dataset: str, useH_A=False, useY_A=False, useY=False):
    """PGDAttack.dot_product_decode2 (topology_attack.py:421-467)."""
    n = Z.shape[0]
    eye = np.eye(n, dtype=F32)

    def l2n(v, p=2):
        nr = (np.abs(v).astype(np.float64) ** p).sum(1) ** (1.0 / p)
        return (v / np.maximum(nr, 1e-12)[:, None]).astype(F32)

    sig = lambda x: (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)
    Z = Z.astype(F32)
    if dataset in ("cora", "AIDS"):
        return sig(np.maximum(Z @ Z.T - eye, 0))
    if dataset == "citeseer":
        Zn = l2n(Z)
        return sig(np.maximum(Zn @ Zn.T - eye, 0))
    if dataset == "brazil":
        return np.maximum(Z @ Z.T - eye, 0).astype(F32)
    if dataset in ("polblogs", "usair"):
        if dataset == "polblogs" and useH_A and useY_A and useY:
            ZZ = l2n(Z @ Z.T)
        elif dataset == "usair" and useY and not useH_A and not useY_A:
            Zn = l2n(Z, 3); ZZ = Zn @ Zn.T
        elif dataset == "usair" and not useY and useH_A and useY_A:
            Zn = l2n(Z, 2); ZZ = Zn @ Zn.T
        elif dataset == "usair" and useY and useH_A and not useY_A:
            Zn = l2n(Z, 5); ZZ = Zn @ Zn.T
        else:
            ZZ = l2n(Z @ Z.T)
        return np.maximum(ZZ - eye, 0).astype(F32)
    raise ValueError(dataset)


# --------------------------------------------------------------------------
# the attack loop
# --------------------------------------------------------------------------
@dataclass
class AdamState:
    """torch.optim.Adam defaults (betas 0.9/0.999, eps 1e-8), single-tensor
    form, as constructed at topology_attack.py:121."""
    lr: float
    m: np.ndarray
    v: np.ndarray
    t: int = 0
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8

    def step(self, p: np.ndarray, g: np.ndarray) -> np.ndarray:
        self.t += 1
        g = g.astype(F32)
        self.m += F32(1 - self.b1) * (g - self.m)           # exp_avg.lerp_(grad, 1-beta1)
        self.v *= F32(self.b2)
        self.v += F32(1 - self.b2) * g * g                   # addcmul_
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        step_size = self.lr / bc1
        denom = (np.sqrt(self.v) / F32(math.sqrt(bc2)) + F32(self.eps)).astype(F32)
        return (p - F32(step_size) * (self.m / denom)).astype(F32)


def bisection(a_vec: np.ndarray, lo: float, hi: float, num_edges: float, epsilon: float):
    """PGDAttack.bisection (topology_attack.py:397-412)."""
    def func(x):
        return float(np.clip(a_vec - F32(x), 0, 1).sum(dtype=F32)) - num_edges
    a, b = F32(lo), F32(hi)
    miu = a
    while (b - a) >= epsilon:
        miu = F32((a + b) / 2)
        if func(miu) == 0.0:
            break
        if func(miu) * func(a) < 0:
            b = miu
        else:
            a = miu
    return F32(miu)


def projection(a_vec: np.ndarray, num_edges: float) -> np.ndarray:
    """PGDAttack.projection (topology_attack.py:338-347)."""
    if np.clip(a_vec, 0, 1).sum(dtype=F32) > num_edges:
        left = (a_vec - 1).min()
        right = a_vec.max()
        miu = bisection(a_vec, left, right, num_edges, 1e-5)
        return np.clip(a_vec - miu, 0, 1).astype(F32)
    return np.clip(a_vec, 0, 1).astype(F32)


@dataclass
class AttackConfig:
    measure: str = "HSIC"
    weight_sup: float = 1.0
    # (w1, w2, _, _, _, w6, w7, w8, w9, w10)  topology_attack.py:151
    weight_param: Sequence[float] = (0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    lr: float = 0.01
    num_edges: float = float("inf")
    eps: float = 0.0
    emb_nlayer: int = 2     # embedding.nlayer at loop entry (main.py:240 leaves 2); GAT's embedding runs all layers
    fin_layers: Sequence[int] = (1, 2)   # depths of H_A1 / H_A2 in the post-loop ensemble (:304-307)


class PGDAttackOracle:
    """Restatement of topology_attack.PGDAttack.attack (topology_attack.py:95-324).

    State is the dense symmetric M (= unpack_sym(adj_changes)); Adam moments are
    kept dense-symmetric too, which is the same optimiser on the packed vector
    because the packed gradient g[p(i,j)] = G_M[i,j] + G_M[j,i] is mirrored.
    """

    def __init__(self, w: GCNWeights, features, adj_true, ori_adj, feature_adj,
                 labels, idx_attack, cfg: AttackConfig):
        assert cfg.measure in MEASURES, cfg.measure
        self.w = w.f32()
        self.cfg = cfg
        self.n = n = adj_true.shape[0]
        self.L = len(self.w.W)
        self.X = np.asarray(features, dtype=F32)
        self.adj_true = np.asarray(adj_true, dtype=F32)
        self.ori = np.asarray(ori_adj, dtype=F32)
        self.fadj = np.asarray(feature_adj, dtype=F32)
        self.labels = np.asarray(labels, dtype=np.int64)
        self.idx = np.asarray(idx_attack, dtype=np.int64)
        self.comp = (np.ones((n, n), dtype=F32) - np.eye(n, dtype=F32))
        self.M = np.zeros((n, n), dtype=F32)                     # Parameter zeros (:77-78)
        self.adam = AdamState(cfg.lr, np.zeros((n, n), F32), np.zeros((n, n), F32))
        self.T0 = (self.X @ self.w.W[0]).astype(F32)             # X @ W_1, adjacency independent
        self.S0 = (self.X @ self.w.Ws[0]).astype(F32) if self.w.Ws is not None else None
        # priors from the true graph, constant over the loop (:177-182; unnormalised adj!)
        Le = cfg.emb_nlayer
        _, Hh, _ = gcn_chain(self.T0, self.adj_true, self.w, Le, self.S0)
        self.HA = Hh[-1]                                          # H_A_cur (:243)
        _, Hv, _ = gcn_chain(self.T0, self.adj_true, self.w, self.L, self.S0)
        _, self.YA = victim_head(Hv[-1], self.w)                  # Y_A (:182), log-probs
        self.fadj_nonconst = bool(self.fadj.max() != self.fadj.min())   # (:212)
        self.adj_norm_last = None
        self.last: Dict[str, object] = {}

    # -- accessors in the reference's packed form ---------------------------
    @property
    def adj_changes(self) -> np.ndarray:
        return pack_tril(self.M)

    def set_adj_changes(self, a: np.ndarray):
        self.M = unpack_sym(np.asarray(a, F32), self.n)

    # -- one iteration of the loop at topology_attack.py:161-298 -------------
    def step(self, noise: Optional[np.ndarray] = None) -> Dict[str, float]:
        cfg, w, n, idx = self.cfg, self.w, self.n, self.idx
        w1, w2, _, _, _, w6, w7, w8, w9, w10 = [float(x) for x in cfg.weight_param]
        sign = -1.0 if cfg.measure == "HSIC" else 1.0            # (:217-220 etc.)
        calc = _CALC[cfg.measure]
        AP = ALIGN_PARAMETER_CORA

        # forward ---------------------------------------------------------
        mod = self.comp * self.M + self.ori                      # get_modified_adj (:164)
        A, gate = adding_noise(mod, cfg.eps, noise)              # (:165)
        adj_norm, d, r = normalize_adj_tensor(A)                 # (:166)
        self.adj_norm_last = adj_norm
        Pv, Hv, Tv = gcn_chain(self.T0, adj_norm, w, self.L, self.S0)     # victim(features, adj_norm) (:167)
        Z, logp = victim_head(Hv[-1], w)
        na = len(idx)
        nll = F32(-logp[idx, self.labels[idx]].mean(dtype=np.float64))      # _loss CE (:326-328)
        a_vec = pack_tril(self.M)
        norm_a = F32(np.sqrt((a_vec.astype(np.float64) ** 2).sum()))
        origin_loss = F32(nll + norm_a * F32(0.001))            # (:172-173)
        loss = float(cfg.weight_sup) * float(origin_loss)        # (:175)

        B = (A - self.ori).astype(F32)                           # modified_adj - ori_adj (:185)
        Le = cfg.emb_nlayer
        Pe, He, Te = gcn_chain(self.T0, B, w, Le, self.S0)                # embedding (:185)
        em = He[-1]
        R_, S, Zn, nrm = dot_product_decode_dense(em)            # (:187)
        A1 = self.comp * sym_from_lower(R_) + self.ori           # get_modified_adj_after (:188)

        G_adjn = np.zeros((n, n), F32)     # d loss / d adj_norm
        G_A1 = np.zeros((n, n), F32)       # d loss / d modified_adj1
        G_em = np.zeros_like(em)           # d loss / d em (both uses, :185 and :241)
        terms: Dict[str, float] = {}

        if w1 != 0 and self.fadj_nonconst:                       # (:212-220)
            k = sign * w1 * 1000 * AP["c1"]
            v, _, gy = calc(self.fadj, adj_norm, need_x=False, need_y=True)
            terms["c1"] = float(w1 * float(v) * 1000 * AP["c1"]); loss += k * float(v)
            G_adjn += F32(k) * gy
        if w2 != 0:                                              # (:221-229)
            k = sign * w2 * 100 * AP["c2"]
            v, gx, gy = calc(adj_norm, A1)
            terms["c2"] = float(w2 * float(v) * 100 * AP["c2"]); loss += k * float(v)
            G_adjn += F32(k) * gx
            G_A1 += F32(k) * gy
        if w6 != 0:                                              # (:230-232)
            k = w6 * 100 * AP["c6"]
            v, g = info_entropy_grad(adj_norm)
            terms["c6"] = float(k * float(v)); loss += k * float(v)
            G_adjn += F32(k) * g
        if w7 != 0:                                              # (:233-236)
            k = w7 * AP["c7"]
            v, g = info_entropy_grad(A1)
            terms["c7"] = float(k * float(v)); loss += k * float(v)
            G_A1 += F32(k) * g
        if w9 != 0:                                              # (:237-258); em_cur == em, H_A_cur == HA
            k = sign * w9 * AP["c9"]
            v, _, gy = calc(self.HA[idx], em[idx], need_x=False, need_y=True)
            terms["c9"] = float(w9 * float(v) * AP["c9"]); loss += k * float(v)
            np.add.at(G_em, idx, F32(k) * gy)
        # output2 = victim(features, modified_adj) (:259)
        Po, Ho, To = gcn_chain(self.T0, A, w, self.L, self.S0)
        Z2, _ = victim_head(Ho[-1], w)
        sm2 = _softmax(Z2)
        G_Ho = None
        if w10 != 0:                                             # (:260-272)
            k = sign * w10 * AP["c10"]
            v, _, gy = calc(self.YA[idx], sm2[idx], need_x=False, need_y=True)
            terms["c10"] = float(w10 * float(v) * AP["c10"]); loss += k * float(v)
            G_sm = np.zeros_like(sm2)
            np.add.at(G_sm, idx, F32(k) * gy)
            G_Z2 = (sm2 * (G_sm - (G_sm * sm2).sum(1, keepdims=True))).astype(F32)
            G_Ho = victim_head_backward(G_Z2, Ho[-1], w)

        # backward --------------------------------------------------------
        # nll -> victim(adj_norm)
        G_Z = np.zeros_like(Z)
        sm = np.exp(logp)
        cnt = np.zeros(n, F32); np.add.at(cnt, idx, 1)
        onehot = np.zeros_like(Z); onehot[np.arange(n), self.labels] = 1
        G_Z = (F32(cfg.weight_sup) * (sm - onehot) * (cnt / F32(na))[:, None]).astype(F32)
        G_adjn += gcn_chain_backward(victim_head_backward(G_Z, Hv[-1], w), adj_norm, Pv, Tv, w, self.L)

        # decode backward: A1 = comp * sym_from_lower(relu(S)) + ori
        Gc = self.comp * G_A1
        GL = np.tril(Gc + Gc.T, -1) * (np.tril(S, -1) > 0)       # grad w.r.t. S[i,j], i>j
        G_Zn = ((GL + GL.T) @ Zn).astype(F32)                    # S = Zn Zn^T
        den = np.maximum(nrm, F32(1e-12))
        big = nrm >= F32(1e-12)
        proj = (Zn * G_Zn).sum(1, keepdims=True)
        G_em += np.where(big[:, None], (G_Zn - Zn * proj) / den[:, None], G_Zn / den[:, None]).astype(F32)

        G_A = np.zeros((n, n), F32)
        G_A += gcn_chain_backward(G_em, B, Pe, Te, w, Le)        # B = A - ori
        if G_Ho is not None:
            G_A += gcn_chain_backward(G_Ho, A, Po, To, w, self.L)

        # normalisation backward: adj_norm = r_i (A+I)_ij r_j, r = d^-1/2
        mx = A + np.eye(n, dtype=F32)
        Gm = G_adjn * mx
        gr = (Gm * r[None, :]).sum(1, dtype=F32) + (Gm * r[:, None]).sum(0, dtype=F32)
        with np.errstate(divide="ignore", invalid="ignore"):
            gd = np.where(d > 0, F32(-0.5) * gr * np.power(d, F32(-1.5)), 0).astype(F32)
        G_A += G_adjn * r[:, None] * r[None, :] + gd[:, None]

        G_M = self.comp * (G_A * gate)
        G_sym = (G_M + G_M.T).astype(F32)          # packed grad mirrored to both halves
        if norm_a > 0:                              # torch.norm backward, 0 at the origin
            G_sym += F32(cfg.weight_sup * 0.001) * self.comp * self.M / norm_a

        # optimizer.step (:279), projection + clamp (:281-283)
        newM = self.adam.step(self.M, G_sym)
        a_new = projection(pack_tril(newM), cfg.num_edges)
        a_new = np.clip(a_new, 0, 1).astype(F32)
        self.last = dict(A=A, adj_norm=adj_norm, d=d, r=r, logp=logp, em=em, S=S, A1=A1,
                         G_adjn=G_adjn, G_A1=G_A1, G_em=G_em, G_A=G_A, G_sym=G_sym,
                         sm2=sm2, loss=loss, nll=float(nll), terms=terms)
        self.M = unpack_sym(a_new, n)

        # monitoring forward on the updated adjacency (:290-296)
        mod2 = self.comp * self.M + self.ori
        adj_norm2, _, _ = normalize_adj_tensor(mod2)
        _, Hm, _ = gcn_chain(self.T0, adj_norm2, w, self.L, self.S0)
        _, out2 = victim_head(Hm[-1], w)
        return dict(loss=loss, origin_loss=float(origin_loss), sparsity=float(mod2.mean()),
                    out_monitor=out2, **terms)

    # -- after the loop (topology_attack.py:300-324) -------------------------
    def finalize(self, dataset: str, useH_A: bool, useY_A: bool, useY: bool,
                 label_adj: Optional[np.ndarray], H_A: np.ndarray, Y_A: np.ndarray) -> np.ndarray:
        w, n = self.w, self.n
        if self.adj_norm_last is None:               # epochs == 0: adj_norm from :142
            self.adj_norm_last, _, _ = normalize_adj_tensor(self.comp * self.M + self.ori)
        _, He, _ = gcn_chain(self.T0, self.adj_norm_last, w, self.cfg.emb_nlayer, self.S0)   # (:300)
        R_, _, _, _ = dot_product_decode_dense(He[-1])
        self.M = sym_from_lower(R_)                                                  # (:301)
        mod = (self.comp * self.M + self.ori).astype(F32)                            # (:302)
        _, H1, _ = gcn_chain(self.T0, mod, w, self.cfg.fin_layers[0], self.S0)       # (:304-305)
        _, H2, _ = gcn_chain(self.T0, mod, w, self.cfg.fin_layers[1], self.S0)       # (:306-307)
        _, Hv, _ = gcn_chain(self.T0, mod, w, self.L, self.S0)
        _, Y2 = victim_head(Hv[-1], w)                                               # (:308)
        dd = lambda z: dot_product_decode2(z, dataset, useH_A, useY_A, useY)
        cur = mod + dd(H1[-1]) + dd(H2[-1]) + self.fadj + dd(Y2)                     # (:311-314)
        if useH_A:
            cur = cur + dd(np.asarray(H_A, F32))
        if useY_A:
            cur = cur + dd(np.asarray(Y_A, F32))
        if useY:
            cur = cur + np.asarray(label_adj, F32)
        return cur.astype(F32)


# --------------------------------------------------------------------------
# metric (main.py:66-75): sklearn roc_curve + auc on all pairs of idx
# --------------------------------------------------------------------------
def auc_score(real: np.ndarray, pred: np.ndarray) -> float:
    """Area under the ROC curve with tie handling identical to
    sklearn.metrics.roc_curve + auc (trapezoid over distinct thresholds):
    equals the Mann-Whitney U statistic with ties counted 1/2."""
    real = np.asarray(real).reshape(-1)
    pred = np.asarray(pred, dtype=np.float64).reshape(-1)
    order = np.argsort(pred, kind="mergesort")
    ps = pred[order]
    ranks = np.empty(len(ps), dtype=np.float64)
    i = 0
    # average ranks over ties
    bounds = np.flatnonzero(np.r_[True, ps[1:] != ps[:-1], True])
    for s, e in zip(bounds[:-1], bounds[1:]):
        ranks[s:e] = 0.5 * (s + e - 1) + 1.0
    rk = np.empty_like(ranks)
    rk[order] = ranks
    pos = real > 0
    npos = int(pos.sum()); nneg = len(real) - npos
    if npos == 0 or nneg == 0:
        return float("nan")
    return float((rk[pos].sum() - npos * (npos + 1) / 2.0) / (npos * nneg))


def metric_pool(ori_adj: np.ndarray, inference_adj: np.ndarray, idx: np.ndarray) -> float:
    """main.metric_pool (main.py:66-75); index_delete is applied after
    roc_curve there, so it does not affect the AUC."""
    real = ori_adj[idx, :][:, idx].reshape(-1)
    pred = inference_adj[idx, :][:, idx].reshape(-1)
    return auc_score(real, pred)
