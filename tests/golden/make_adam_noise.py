#!/usr/bin/env python3
"""How far can two correct fp32 implementations of the README Cora run (MSELoss, lr 0.01) drift apart?

Adam's update m / sqrt(v) is scale free per entry, so an entry whose gradient sits at the fp32 rounding level moves by
+-lr on the SIGN of that noise; over many epochs trajectories of implementations that differ only in summation order
separate.  This script measures it (build container only; imports the reference):
  * the reference itself (torch CPU) at 10 / 20 / 40 / 100 epochs,
  * the numpy oracle in float32 and in float64 (same algorithm, F32 := float64) at the same horizons.
Output: profiles/r02_adam_noise_experiment.json and tests/golden/cora_mse_checkpoints.npz (the reference's AUC at the
intermediate horizons, consumed by tests/test_gpu_parity.py::test_cora_mse_checkpoints).

    python tests/golden/make_adam_noise.py        (~15 min on 8 cores)
"""
import json
import os
import sys
import time
from copy import deepcopy

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from tests import helpers as H           # noqa: E402
from oracle import mcgra_oracle as O     # noqa: E402

HORIZONS = (10, 20, 40, 100)


def oracle_curve(z, dtype):
    O.F32 = dtype
    w = O.GCNWeights([z["W0"].astype(dtype), z["W1"].astype(dtype)], [z["b0"].astype(dtype), z["b1"].astype(dtype)],
                     z["Wlin"].astype(dtype), z["blin"].astype(dtype))
    X, adj, lab = z["features"].astype(dtype), z["adj"].astype(dtype), z["labels"]
    n = adj.shape[0]
    cfg = O.AttackConfig(measure=str(z["measure"]), weight_sup=float(z["weight_sup"]),
                         weight_param=tuple(float(x) for x in z["weight_param"]), lr=float(z["lr"]),
                         num_edges=float(z["num_edges"]))
    orc = O.PGDAttackOracle(w, X, adj, np.zeros((n, n), dtype), H.cora_feature_adj(z["features"]).astype(dtype), lab,
                            z["idx_attack"], cfg)
    orc.w = w
    label_adj = (lab[:, None] == lab[None, :]).astype(dtype)
    _, Hs, _ = O.gcn_chain(orc.T0, adj, w, 2)
    _, Hv, _ = O.gcn_chain(orc.T0, adj, w, 2)
    _, YA = O.victim_head(Hv[-1], w)
    out = {}
    for t in range(1, max(HORIZONS) + 1):
        orc.step()
        if t in HORIZONS:
            keep = orc.M.copy()
            final = orc.finalize("cora", True, True, True, label_adj, Hs[-1], YA)
            orc.M = keep
            out[t] = O.metric_pool(z["adj"], final, z["idx_attack"])
            print(dtype.__name__, t, out[t], flush=True)
    O.F32 = np.float32
    return out


def reference_curve(z):
    import make_golden as MG            # imports the reference (build container only)
    import tempfile
    torch = MG.torch
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        os.makedirs("saved_data", exist_ok=True)
        lab = z["labels"]
        np.save("saved_data/cora.npy", (lab[:, None] == lab[None, :]).astype(np.float32))
        device = torch.device("cpu")
        victim = MG.GCN(nfeat=z["features"].shape[1], nclass=int(lab.max()) + 1, nhid=16, nlayer=2, dropout=0.5,
                        weight_decay=5e-4, device=device)
        with torch.no_grad():
            for l in range(2):
                victim.gc[l].weight.copy_(torch.tensor(z[f"W{l}"])); victim.gc[l].bias.copy_(torch.tensor(z[f"b{l}"]))
            victim.linear1.weight.copy_(torch.tensor(z["Wlin"])); victim.linear1.bias.copy_(torch.tensor(z["blin"]))
        adj, feats, labels = torch.tensor(z["adj"]), torch.tensor(z["features"]), torch.LongTensor(lab)
        for ep in HORIZONS:
            res = MG.run_reference_attack(adj, feats, labels, victim, z["idx_attack"], str(z["measure"]),
                                          tuple(float(x) for x in z["weight_param"]), float(z["weight_sup"]), float(z["lr"]),
                                          ep, "cora", (True, True, True), float(z["num_edges"]), capture_steps=False)
            out[ep] = res["auc"]
            print("reference", ep, res["auc"], flush=True)
    return out


def main():
    z = H.load_cora("cora_mse_readme")
    t0 = time.time()
    ref = reference_curve(z)
    o32 = oracle_curve(z, np.float32)
    o64 = oracle_curve(z, np.float64)
    rows = [{"epochs": t, "reference_fp32": ref[t], "oracle_fp32": o32[t], "oracle_fp64": o64[t],
             "abs_ref_minus_oracle32": abs(ref[t] - o32[t]), "abs_ref_minus_oracle64": abs(ref[t] - o64[t]),
             "abs_oracle32_minus_oracle64": abs(o32[t] - o64[t])} for t in HORIZONS]
    out = {"what": "recovered-adjacency AUC of the README Cora MSELoss run at several horizons: the reference (torch CPU fp32), "
                   "the numpy oracle in fp32 and the same oracle in fp64; three evaluations of ONE algorithm on identical inputs",
           "fixture_auc_100": float(z["auc"]), "rows": rows, "seconds": round(time.time() - t0, 1)}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r02_adam_noise_experiment.json"), "w") as f:
        json.dump(out, f, indent=1)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cora_mse_checkpoints.npz"),
                        epochs=np.array(HORIZONS), auc_reference=np.array([ref[t] for t in HORIZONS]),
                        auc_oracle_fp32=np.array([o32[t] for t in HORIZONS]), auc_oracle_fp64=np.array([o64[t] for t in HORIZONS]))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
