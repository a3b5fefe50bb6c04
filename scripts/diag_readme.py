"""GPU box: the reference's README lines (tests/golden/readme_*.npz: cora, citeseer, polblogs, usair, brazil, AIDS) through the
engine: first-step gradient against the float64 truth and the reference, AUC against the reference's (and the reference's
own distance from the float64 run), which step implementation ran, time per step.  -> profiles/r03_readme_lines.txt"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mcgra_loader
pkg = mcgra_loader.load()
from tests import helpers as H
from oracle import mcgra_oracle as O
from mc_gra_amd.topology_attack import _decode_mode

TRUTH = np.load(os.path.join(H.GOLDEN, "readme_fp64.npz"))
print(f"{'fixture':34s} {'n':>5s} {'README':>6s} {'measure':8s} {'lr':>8s} {'g0 vs fp64':>10s} {'ref vs fp64':>11s} {'AUC ref':>9s} {'AUC diff':>9s} {'ref-fp64':>9s} {'fused':>5s} {'ms/step':>8s}")
for name in H.readme_cases():
    z = H.load_readme(name)
    n = len(z["labels"])
    wp = [float(x) for x in z["weight_param"]]
    if not (z["feature_adj"].max() != z["feature_adj"].min()):
        wp[0] = 0.0
    eng = H.engine_from(pkg, z, weight_param=tuple(wp))
    pi, pj = H.tril_pos(z["packed_pos"])
    ti, tj = torch.as_tensor(pi, device="cuda:0"), torch.as_tensor(pj, device="cuda:0")
    g64, gmax = TRUTH[f"{name}_g64"].astype(np.float64), float(TRUTH[f"{name}_gmax"])
    for t in range(int(z["epochs"])):
        nz = H.noise_of(z, t)
        eng.step(noise=None if nz is None else torch.as_tensor(nz, device="cuda:0"))
        if t == 0:
            g = eng.buffer("G_sym")[ti, tj].cpu().numpy().astype(np.float64)
            e0 = np.abs(g - g64).max() / gmax
            r0 = np.abs(z["step_g"][0] - g64).max() / gmax
    use = [bool(u) for u in z["use"]]
    lab = z["labels"]
    args = argparse.Namespace(dataset=str(z["dataset"]), useH_A=use[0], useY_A=use[1], useY=use[2])
    final = eng.finalize(_decode_mode(args), z["H_A2"] if use[0] else None, z["Y_A"] if use[1] else None,
                         (lab[:, None] == lab[None, :]).astype(np.float32) if use[2] else None).cpu().numpy()
    auc = O.metric_pool(z["adj"], final, z["idx_attack"])
    fused = eng.fused_steps()
    # time: 20 more steps without noise bookkeeping (eps lines: fresh device noise)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20):
        eng.step(noise=torch.randn(n, n, device="cuda:0") if float(z["eps"]) != 0 else None)
    torch.cuda.synchronize(); ms = (time.time() - t0) / 20 * 1e3
    print(f"{name:34s} {n:5d} {int(z['readme_line']):6d} {str(z['measure']):8s} {float(z['lr']):8.1e} {e0:10.1e} {r0:11.1e} {float(z['auc']):9.6f} {abs(auc - float(z['auc'])):9.1e} {abs(float(TRUTH[name + '_auc64']) - float(z['auc'])):9.1e} {fused:5d} {ms:8.3f}")
    eng.close()
