"""Skinny product A[n x n] @ T[n x c] timing vs the split-K block target (mcgra_set_gemm_variant bits 20..27, x256)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mcgra_loader
pkg = mcgra_loader.load()
from mc_gra_amd import engine as E
from mc_gra_amd._lib import lib
for n in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["10000", "2708"])]:
    A = torch.rand(n, n, device="cuda") - 0.5
    for c, ta in [(16, False), (32, False), (16, True), (36, True)]:
        T = torch.rand(n, c, device="cuda") - 0.5
        out = torch.empty(n, c, device="cuda")
        res = []
        for tgt in (2, 4, 6, 8, 12):
            lib.mcgra_set_gemm_variant(2 | (tgt << 20))
            E.sgemm(A, T, ta=ta, out=out); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                E.sgemm(A, T, ta=ta, out=out)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            res.append(f"{tgt*256}: {dt*1e6:6.1f} us ({n*n*4/dt/1e12:4.2f} TB/s)")
        print(f"n={n} c={c} ta={int(ta)}  " + "  ".join(res), flush=True)
