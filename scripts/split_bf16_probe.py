"""Feasibility probe for DESIGN.md section 3 'next': P1 = Kf Xc as a 3-plane bf16 split (6 products, i + j <= 2) run as
ONE plain bf16 GEMM with the planes concatenated along K (K' = 6 K), fp32 accumulation, through the library GEMM that
torch calls.  Reports time and the error against fp64 next to the fp32 MFMA kernel of this repo."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mcgra_loader
pkg = mcgra_loader.load()
from mc_gra_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
torch.manual_seed(0)
dev = "cuda"
F = torch.rand(n, 64, device=dev) - 0.5
A = (F @ F.T).contiguous()                       # symmetric, like the centred Gram
B = (torch.rand(n, n, device=dev) - 0.5) * 1e-2


def planes(X):
    p0 = X.to(torch.bfloat16)
    r = X - p0.float()
    p1 = r.to(torch.bfloat16)
    r = r - p1.float()
    p2 = r.to(torch.bfloat16)
    return p0, p1, p2


a0, a1, a2 = planes(A)
b0, b1, b2 = planes(B.T.contiguous())            # [n][k] layout for the NT product
Acat = torch.cat([a0, a0, a0, a1, a1, a2], dim=1).contiguous()      # [n, 6n]
Bcat = torch.cat([b0, b1, b2, b0, b1, b0], dim=1).contiguous()      # [n, 6n]
del a0, a1, a2, b0, b1, b2
torch.cuda.synchronize()


def run():
    try:
        return torch.mm(Acat, Bcat.T, out_dtype=torch.float32)
    except TypeError:
        return None


C = run()
if C is None:
    print("torch.mm(out_dtype=float32) not available in this torch; bf16-output product only")
    C = (Acat @ Bcat.T).float()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    C = run() if C is not None else None
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"n={n}: split-bf16 K'=6K library GEMM {dt*1e3:.2f} ms = {12*n**3/dt/1e12:.0f} TFLOP/s bf16 issued, {2*n**3/dt/1e12:.0f} fp32-equivalent")

S = E.ssyrk_lower(F)
out = torch.empty(n, n, device=dev)
E.ssymm_lower(S, B, out=out); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    E.ssymm_lower(S, B, out=out)
torch.cuda.synchronize()
dt2 = (time.perf_counter() - t0) / 3
print(f"n={n}: fp32 MFMA SYMM (this repo) {dt2*1e3:.2f} ms = {2*n**3/dt2/1e12:.0f} TFLOP/s")
rows = slice(0, 512)
ref = A[rows].double() @ B.double()
scale = (A[rows].abs().double() @ B.abs().double())
print("max |err| / (|A||B|):  split-bf16 %.2e   fp32 MFMA %.2e" % (float(((C[rows].double() - ref).abs() / scale).max()),
                                                                   float(((out[rows].double() - ref).abs() / scale).max())))
