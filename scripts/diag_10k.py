"""Per-step gradient / state errors of the HIP engine against tests/golden/bench10k_hsic.npz and the float64 truth
(tests/golden/bench10k_hsic_ref64.npz: the reference's own code in float64; bench10k_hsic_fp64.npz: the numpy oracle in
float64).  GPU box only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MCGRA_KEEP_GSYM", "1"); os.environ.setdefault("MCGRA_AB", "1")
import torch
import mcgra_loader
pkg = mcgra_loader.load()
import bench
from tests.test_gpu_fullsize import _tril_pos
dev = torch.device("cuda:0")
WL = "synthetic-10k-hsic"
z = np.load(os.path.join(ROOT, "tests/golden/bench10k_hsic.npz"))
p64 = os.path.join(ROOT, "tests/golden/bench10k_hsic_ref64.npz")
z64 = np.load(p64) if os.path.exists(p64) else None
zo64 = np.load(os.path.join(ROOT, "tests/golden/bench10k_hsic_fp64.npz"))
n = bench.WORKLOADS[WL][0]
pi, pj = _tril_pos(z["packed_pos"])
ti, tj = torch.as_tensor(pi, device=dev), torch.as_tensor(pj, device=dev)
for name in ["run"] + sorted({k[:4] for k in z.files if k.startswith("one")}):
    sd, sc = int(z[f"{name}_a0_seed"]), float(z[f"{name}_a0_scale"])
    eng, inp, adj_dev = bench.build_engine(pkg, torch, dev, WL, int(z["seed"]))
    if (sd, sc) != (int(z["seed"]), bench.start_scale(WL, n)):
        eng.set_adj_changes(torch.as_tensor(bench.make_a0(n, sd, sc), device=dev))
    G, A = z[f"{name}_g"], z[f"{name}_a"]
    for t in range(G.shape[0]):
        sc_ = eng.step(want_scalars=True); eng.monitor()
        Gs = eng.buffer("G_sym")
        g = Gs[ti, tj].cpu().numpy()
        gmax = float(z[f"{name}_g_absmax"][t])
        e = np.abs(g - G[t])
        line = f"{name} step {t}: gmax {gmax:.3e} hip-vs-ref max {e.max()/gmax:.3e} rms {np.sqrt((e**2).mean())/gmax:.3e} hipmax {float(Gs.abs().max()):.3e}"
        if z64 is not None and t == 0 and f"{name}_g64ref" in z64.files:
            g64 = z64[f"{name}_g64ref"]
            rms = lambda d: np.sqrt(np.mean(np.square(d.astype(np.float64)))) / gmax
            line += (f" | hip-vs-ref64 max {np.abs(g - g64).max()/gmax:.3e} rms {rms(g - g64):.3e}  ref32-vs-ref64 max {np.abs(G[t] - g64).max()/gmax:.3e}"
                     f" rms {rms(G[t] - g64):.3e}  oracle64-vs-ref64 {np.abs(zo64[name + '_g64'] - g64).max()/gmax:.3e}")
        a = eng.buffer("M")[ti, tj].cpu().numpy()
        moved = np.abs(a - np.clip(A[t], 0, 1)) > 0.05 * float(z["lr"])
        line += f" | moved {moved.mean():.4f} loss {sc_['loss']:.6e} c1 {sc_['c1']:.4e} c2 {sc_['c2']:.4e}"
        print(line, flush=True)
    lab = torch.as_tensor(inp["labels"], device=dev)
    final = eng.finalize(0, eng.buffer("HA"), eng.buffer("YA"), (lab[:, None] == lab[None, :]).float())
    print(name, "auc hip", bench.gpu_auc(adj_dev, final, torch), "ref", float(z[f"{name}_auc"]), flush=True)
    del eng, final, Gs
    torch.cuda.empty_cache()
