#!/usr/bin/env python3
"""How much of the adjacency gradient do the N x N loss terms carry?  (CPU, numpy oracle.)

For a bench workload (bench.WORKLOADS, bench.make_inputs, bench.make_a0) it walks the oracle through `--steps` attack
steps with the workload's own weights and, AT THE STATE OF EVERY STEP, evaluates the gradient again with single loss
terms switched on (weight_sup = 0 and every other weight 0):
    c1  = w1 * calc(feature_adj, adj_norm)         topology_attack.py:212-220   (the N x N x N product P1 on the GPU)
    c2  = w2 * calc(adj_norm, modified_adj1)       :221-229                      (low-rank factors / decode backward)
    c6, c7 = the two Info_entropy terms            :230-236
    c9, c10 = the small-operand terms              :237-272
and prints |G_term|_max / |G_full|_max.  A term whose share is below 2^-24 ~ 6e-8 is rounded away in the fp32 sum of
the gradient: a parity test on the full gradient then cannot see the kernels that compute it.

    python scripts/nxn_share.py --workload synthetic-10k-hsic --nodes 2048 --steps 4 [--scale S] [--lr LR] [--json out]

`--nodes` shrinks the workload's N (same generator); the default is the workload's own N (N = 10 000: ~2 min per
gradient evaluation on 8 cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                     # noqa: E402
from oracle import mcgra_oracle as O  # noqa: E402

TERMS = {"c1": 0, "c2": 1, "c6": 5, "c7": 6, "c9": 8, "c10": 9}


def build(workload, ns, seed, wp, wsup, lr):
    n0, f, c, hid, nl, measure, _ = B.WORKLOADS[workload][:7]
    inp = B.make_inputs(ns, f, c, hid, nl, seed)
    X = inp["features"]
    fadj = (1.0 / (1.0 + np.exp(-np.maximum(X @ X.T - np.eye(ns, dtype=np.float32), 0)))).astype(np.float32)
    w = O.GCNWeights(inp["W"], inp["b"], inp["Wlin"], inp["blin"])
    cfg = O.AttackConfig(measure=measure, weight_sup=wsup, weight_param=wp, lr=lr, num_edges=float("inf"))
    return O.PGDAttackOracle(w, X, inp["adj"], np.zeros((ns, ns), np.float32), fadj, inp["labels"], inp["idx_attack"], cfg), inp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synthetic-10k-hsic")
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--scale", type=float, default=None, help="start scale of adj_changes (default: the workload's, bench.start_scale)")
    ap.add_argument("--lr", type=float, default=None)
    ap.add_argument("--terms", default="c1,c2,c6,c7,c9,c10")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    n0 = B.WORKLOADS[a.workload][0]
    wp = B.WORKLOADS[a.workload][6]
    ns = a.nodes or n0
    scale = a.scale if a.scale is not None else B.start_scale(a.workload, ns)
    lr = a.lr if a.lr is not None else B.workload_lr(a.workload)
    full, _ = build(a.workload, ns, a.seed, wp, 1.0, lr)
    full.set_adj_changes(B.make_a0(ns, a.seed, scale))
    cfg_full = full.cfg
    singles = {}
    for t in a.terms.split(","):
        w1 = [0.0] * 10
        w1[TERMS[t]] = wp[TERMS[t]]
        if w1[TERMS[t]] != 0:
            singles[t] = O.AttackConfig(measure=cfg_full.measure, weight_sup=0.0, weight_param=tuple(w1), lr=lr, num_edges=float("inf"))
    rows = []
    for s in range(a.steps):
        M = full.M.copy()
        t0 = time.time()
        share = {}
        # single-term gradients at this state: the same oracle object with the other weights zeroed (its optimiser state
        # is put back afterwards, so the run itself is the full-weight run)
        keep = (full.adam.m.copy(), full.adam.v.copy(), full.adam.t)
        gmax_t = {}
        for t, cfg1 in singles.items():
            full.cfg = cfg1
            full.step()
            gmax_t[t] = float(np.abs(full.last["G_sym"]).max())
            full.M = M.copy()
            full.adam.m[:] = keep[0]; full.adam.v[:] = keep[1]; full.adam.t = keep[2]
        full.cfg = cfg_full
        del keep
        sc = full.step()
        G = full.last["G_sym"]
        gmax = float(np.abs(G).max())
        row = {"step": s, "gmax": gmax, "loss": float(sc["loss"]), "rowsum_mean": float(M.sum(1).mean()),
               "neg_frac": float((G < 0).mean()), "masked_pairs": int((full.last["S"][np.tril_indices(ns, -1)] <= 0).sum()),
               "terms": {k: float(v) for k, v in full.last["terms"].items()},
               "share": {t: v / gmax for t, v in gmax_t.items()}}
        row["seconds"] = round(time.time() - t0, 1)
        rows.append(row)
        print(json.dumps(row), flush=True)
    out = {"workload": a.workload, "nodes": ns, "seed": a.seed, "start_scale": scale, "lr": lr, "weight_param": list(wp),
           "what": "|G_term|_max / |G_full|_max of the mirrored packed gradient (numpy oracle), per step of the run", "steps": rows}
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
