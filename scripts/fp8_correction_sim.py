#!/usr/bin/env python3
"""Would fp8 matrix cores do for the two correction products of the split N x N x N product?  (CPU, numpy.)

split2_m16_kernel evaluates P1 = A B (A = H Kf H, B = adj_norm) as x0 y0 + x0 y1 + x1 y0 on fp16 planes: three fp16 MFMA
products.  The block-scaled fp8 MFMA of gfx950 (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 operands) runs at twice the
fp16 rate, so rounding the operands of the two 2^-11 corrections to e4m3 would cut the matrix time to 2/3.  This script
rounds them (OCP e4m3fn, round to nearest even, exact products and sums otherwise) on the bench's own operands and on a
sparse-graph state, and prints the error of each arithmetic against float64, in units of the largest |P1|.

    python scripts/fp8_correction_sim.py [n ...]            (default 1024 3000; ~1 min)

Result (profiles/r03_fp8_correction_sim.txt): 2e-5 of the largest magnitude against 2e-7 for the three fp16 products and
5e-7 ... 2e-6 for an fp32 GEMM -- the diagonal of adj_norm and the edges of a sparse state are single terms that carry a
dot product, so the 2^-4 rounding of an e4m3 operand does not average out.  Not fp32-level: not built.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                     # noqa: E402


def e4m3(x):
    """round to nearest even onto OCP e4m3fn: 3 mantissa bits, smallest normal 2^-6, subnormal step 2^-9, largest 448"""
    x = np.asarray(x, np.float64)
    a = np.abs(x)
    e = np.maximum(np.floor(np.log2(np.maximum(a, 1e-300))), -6)
    step = 2.0 ** (e - 3)
    return np.sign(x) * np.minimum(np.round(a / step) * step, 448.0)


def f16(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float64)


def run(n, seed, state):
    wl = "synthetic-10k-hsic"
    _, f, c, hid, nl, _, _ = B.WORKLOADS[wl][:7]
    X = B.make_inputs(n, f, c, hid, nl, seed)["features"].astype(np.float64)
    Kf = (1.0 / (1.0 + np.exp(-np.maximum(X @ X.T - np.eye(n), 0)))).astype(np.float32).astype(np.float64)
    H = np.eye(n) - 1.0 / n
    A = (H @ Kf @ H).astype(np.float32).astype(np.float64)
    if state == "bench":
        M = np.zeros((n, n))
        M[np.tril_indices(n, -1)] = B.make_a0(n, seed, B.start_scale(wl, n))
        M = M + M.T
    else:                                   # three edges per node near 1 on a floor of 1e-4
        rng = np.random.default_rng(1)
        M = np.tril((rng.random((n, n)) < 3.0 / n) * rng.random((n, n)) + 1e-4 * rng.random((n, n)), -1)
        M = M + M.T
    r = (M.sum(1) + 1) ** -0.5
    Bm = (r[:, None] * (M + np.eye(n)) * r[None, :]).astype(np.float32).astype(np.float64)
    truth = A @ Bm
    gmax = np.abs(truth).max()
    sa = 2.0 ** (15 - int(np.frexp(np.abs(A).max())[1]))
    sb = 2.0 ** (15 - int(np.frexp(np.abs(Bm).max())[1]))
    A0 = f16(A * sa); A1 = f16(A * sa - A0)
    B0 = f16(Bm * sb); B1 = f16(Bm * sb - B0)

    def q8(x, s):
        return e4m3(x * s) / s
    arith = {
        "3 x fp16 (built)": (A0 @ B0 + A0 @ B1 + A1 @ B0) / (sa * sb),
        "fp16 + 2 x e4m3": (A0 @ B0 + q8(A0, 2.0 ** -7) @ q8(B1, 2.0 ** 4) + q8(A1, 2.0 ** 4) @ q8(B0, 2.0 ** -7)) / (sa * sb),
        "1 x fp16": (A0 @ B0) / (sa * sb),
        "fp32 GEMM": (A.astype(np.float32) @ Bm.astype(np.float32)).astype(np.float64),
    }
    for name, v in arith.items():
        e = np.abs(v - truth)
        print(f"n={n:5d} {state:6s} {name:18s} max {e.max() / gmax:.2e}   rms {np.sqrt((e ** 2).mean()) / gmax:.2e}", flush=True)


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [1024, 3000]:
        for state in ("bench", "sparse"):
            run(n, 0, state)
