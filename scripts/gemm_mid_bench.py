"""GPU box: the engine's fp32 GEMM on the shapes of a wide-embedding victim's chain (N x N times N x w, 64 < w <= 128) and its
rank-w update, against torch.matmul (hipBLASLt) as a reference point.  python3 scripts/gemm_mid_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mcgra_loader
pkg = mcgra_loader.load()
from mc_gra_amd import engine as E

def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for n in (3312, 10000):
    for w in (80, 128):
        A = torch.rand(n, n, device="cuda"); B = torch.randn(n, w, device="cuda"); C = torch.empty(n, w, device="cuda")
        G = torch.randn(n, w, device="cuda"); O = torch.zeros(n, n, device="cuda")
        r = {}
        r["NN"] = t(lambda: E.sgemm(A, B, out=C)); r["NN_torch"] = t(lambda: torch.matmul(A, B, out=C))
        err = float((E.sgemm(A, B) - A.double().matmul(B.double()).float()).abs().max())
        r["TN"] = t(lambda: E.sgemm(A, B, ta=True, out=C)); r["TN_torch"] = t(lambda: torch.matmul(A.t(), B, out=C))
        r["NT_b1"] = t(lambda: E.sgemm(G, B, tb=True, beta=1.0, out=O)); r["NT_torch"] = t(lambda: O.addmm_(G, B.t()))
        O.zero_(); E.sgemm(G, B, tb=True, beta=1.0, out=O)
        err2 = float((O - G.double().matmul(B.double().t()).float()).abs().max())
        print(f"n {n} w {w} " + " ".join(f"{k} {v:7.1f}" for k, v in r.items()) + f"  us   maxerr NN {err:.2e} NT {err2:.2e}", flush=True)
