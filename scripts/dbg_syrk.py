import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mcgra_loader
pkg = mcgra_loader.load()
from mc_gra_amd import engine as E
for n, k in [(300, 77), (300, 64), (256, 96), (384, 96), (300, 96)]:
    rng = np.random.RandomState(1)
    A = rng.randn(n, k).astype(np.float32)
    ref = A.astype(np.float64) @ A.astype(np.float64).T
    out = E.ssyrk_lower(torch.tensor(A, device="cuda"), out=torch.full((n, n), float("nan"), device="cuda")).cpu().numpy()
    i, j = np.indices((n, n)); valid = j < (i // 128 + 1) * 128
    bad = valid & ~(np.abs(out - ref) <= 1e-3)
    print(n, k, "bad", bad.sum(), "nan in valid", np.isnan(out[valid]).sum())
    if bad.sum():
        bi, bj = np.nonzero(bad)
        print("  rows", bi.min(), bi.max(), "cols", bj.min(), bj.max(), "tiles", sorted(set(zip((bi // 128).tolist(), (bj // 128).tolist()))))
        print("  sample", [(int(a), int(b), float(out[a, b]), float(ref[a, b])) for a, b in list(zip(bi, bj))[:4]])
