#!/usr/bin/env python3
"""Error of the split N x N x N product with its residual planes x1, y1 rounded to fewer significant bits (CPU, numpy):
fewer set bits in the operands of the two correction products lower the matrix cores' power and -- on a power-limited
chip -- raise the clock (measured: DESIGN.md section 8).  8 bits put the product in the error class of an fp32 GEMM, 11
(what is built) 5 x below it.      python scripts/residual_bits_sim.py"""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
import bench as B
def f16(x): return np.asarray(x, np.float32).astype(np.float16).astype(np.float64)
def rbits(x, bits):
    # round fp16-representable values to `bits` significant bits (RNE)
    x = np.asarray(x, np.float64)
    m, e = np.frexp(x)
    return np.ldexp(np.round(m * 2.0**bits) / 2.0**bits, e)
def run(n, state):
    wl = "synthetic-10k-hsic"
    _, f, c, hid, nl, _, _ = B.WORKLOADS[wl][:7]
    X = B.make_inputs(n, f, c, hid, nl, 0)["features"].astype(np.float64)
    Kf = (1.0 / (1.0 + np.exp(-np.maximum(X @ X.T - np.eye(n), 0)))).astype(np.float32).astype(np.float64)
    H = np.eye(n) - 1.0 / n
    A = (H @ Kf @ H).astype(np.float32).astype(np.float64)
    if state == "bench":
        M = np.zeros((n, n)); M[np.tril_indices(n, -1)] = B.make_a0(n, 0, B.start_scale(wl, n)); M = M + M.T
    else:
        rng = np.random.default_rng(1)
        M = np.tril((rng.random((n, n)) < 3.0 / n) * rng.random((n, n)) + 1e-4 * rng.random((n, n)), -1); M = M + M.T
    r = (M.sum(1) + 1) ** -0.5
    Bm = (r[:, None] * (M + np.eye(n)) * r[None, :]).astype(np.float32).astype(np.float64)
    truth = A @ Bm; gmax = np.abs(truth).max()
    sa = 2.0 ** (15 - int(np.frexp(np.abs(A).max())[1])); sb = 2.0 ** (15 - int(np.frexp(np.abs(Bm).max())[1]))
    A0 = f16(A * sa); A1 = f16(A * sa - A0); B0 = f16(Bm * sb); B1 = f16(Bm * sb - B0)
    for bits in (11, 9, 8, 7, 6, 5):
        A1t, B1t = rbits(A1, bits), rbits(B1, bits)
        v = (A0 @ B0 + A0 @ B1t + A1t @ B0) / (sa * sb)
        e = np.abs(v - truth)
        print(f"n={n} {state:6s} residual planes at {bits:2d} bits: max {e.max()/gmax:.2e} rms {np.sqrt((e**2).mean())/gmax:.2e}")
    f32 = (A.astype(np.float32) @ Bm.astype(np.float32)).astype(np.float64)
    e = np.abs(f32 - truth); print(f"n={n} {state:6s} fp32 GEMM: max {e.max()/gmax:.2e} rms {np.sqrt((e**2).mean())/gmax:.2e}")
for st in ("bench", "sparse"): run(2048, st)
