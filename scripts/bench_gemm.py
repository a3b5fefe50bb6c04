"""GEMM micro-benchmark: interleaved A/B of kernel variants in one process (fp32 MFMA)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mcgra_loader
pkg = mcgra_loader.load()
from mc_gra_amd import engine as E
from mc_gra_amd._lib import lib
sizes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4096", "8192", "10000"])]
variants = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "2"])]
torch.manual_seed(0)
for n in sizes:
    A = torch.rand(n, n, device="cuda") * 2 - 1
    B = torch.rand(n, n, device="cuda") * 2 - 1
    C = torch.empty(n, n, device="cuda")
    for (ta, tb) in [(False, False), (False, True), (True, False)]:
        res = {v: [] for v in variants}
        for rnd in range(4):
            for v in variants:
                lib.mcgra_set_gemm_variant(v)
                E.sgemm(A, B, ta=ta, tb=tb, out=C); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    E.sgemm(A, B, ta=ta, tb=tb, out=C)
                torch.cuda.synchronize()
                res[v].append((time.perf_counter() - t0) / 3)
        print(f"n={n} ta={int(ta)} tb={int(tb)} " + "  ".join(f"v{v}: {2*n**3/min(r)/1e12:6.1f} TF (med {2*n**3/sorted(r)[len(r)//2]/1e12:6.1f})" for v, r in res.items()), flush=True)
