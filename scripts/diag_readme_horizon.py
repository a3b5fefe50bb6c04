"""GPU box: the README lines (eps == 0) through PGDAttack.attack for the horizon of tests/golden/horizon<epochs>_readme.npz (20 or
100 epochs) against the reference's own run of that length and its float64 run -> profiles/r05_readme_horizon<epochs>.txt
    python3 scripts/diag_readme_horizon.py [20|100]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mcgra_loader
pkg = mcgra_loader.load()
from tests import helpers as H
from tests import test_gpu_readme as T
from oracle import mcgra_oracle as O

Z = np.load(os.path.join(H.GOLDEN, f"horizon{int(sys.argv[1]) if len(sys.argv) > 1 else 20}_readme.npz"))
ep = int(Z["epochs"])
print(f"{'fixture':34s} {'n':>5s} {'measure':8s} {'AUC reference':>13s} {'engine - ref':>12s} {'ref64 - ref':>12s} {'path':>14s}   ({ep} epochs)")
worst = 0.0
for name in H.readme_cases():
    if f"{name}_auc" not in Z.files:
        continue
    z = H.load_readme(name)
    if f"{name}_noise_seed" in Z.files:      # (eps != 0 lines run in tests/test_gpu_readme.py, where the class's noise draw is patched)
        continue
    final = T._class_run(pkg, z, ep)
    auc = O.metric_pool(z["adj"], final, z["idx_attack"])
    ref, r64 = float(Z[f"{name}_auc"]), float(Z[f"{name}_auc64"])
    worst = max(worst, abs(auc - ref) / max(1e-4, abs(r64 - ref)))
    print(f"{name:34s} {len(z['labels']):5d} {str(z['measure']):8s} {ref:13.6f} {auc - ref:12.1e} {r64 - ref:12.1e}", flush=True)
print(f"largest |engine - ref| / max(1e-4, |ref64 - ref|): {worst:.2f}")
