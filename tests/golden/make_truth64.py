#!/usr/bin/env python3
"""float64 evaluation of the reference ALGORITHM (the numpy oracle with F32 := float64) on the bench's own workload:
the step-0 gradient at the packed positions tests/golden/bench10k_hsic.npz samples.  It says how far the reference's
own float32 run is from the exact gradient at N = 10 000 (its Gram evaluation of linear_HSIC sums 10^8 products of
O(1) Gram entries in fp32), which bounds what "equal to the reference" can mean per entry at that size.

    python tests/golden/make_truth64.py          (~10 min and ~30 GB per start on 8 cores)
    python tests/golden/make_truth64.py mid      (the n = 1200 fixture, seconds)
    python tests/golden/make_truth64.py readme   (the README-line fixtures, ~10 min; `readme+`: only the ones the file lacks)

Writes tests/golden/bench10k_hsic_fp64.npz: for each start of the fixture (`run`, `one0`, ...) `<name>_g64` = the
mirrored packed gradient of its first step at `packed_pos`, plus its largest magnitude over the whole vector."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B                     # noqa: E402
from oracle import mcgra_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main(workload="synthetic-10k-hsic", tag="bench10k_hsic"):
    z = np.load(os.path.join(OUT, f"{tag}.npz"))
    seed = int(z["seed"])
    n, f, c, hid, nl, measure, wp = B.WORKLOADS[workload]
    O.F32 = np.float64
    inp = B.make_inputs(n, f, c, hid, nl, seed)
    X = inp["features"].astype(np.float64)
    fadj = 1.0 / (1.0 + np.exp(-np.maximum(X @ X.T - np.eye(n), 0)))
    w = O.GCNWeights([x.astype(np.float64) for x in inp["W"]], [x.astype(np.float64) for x in inp["b"]],
                     inp["Wlin"].astype(np.float64), inp["blin"].astype(np.float64))
    cfg = O.AttackConfig(measure=measure, weight_sup=1.0, weight_param=wp, lr=float(z["lr"]), num_edges=float("inf"))
    pk = z["packed_pos"]
    i = ((1.0 + np.sqrt(1.0 + 8.0 * pk.astype(np.float64))) / 2.0).astype(np.int64)
    i = np.where(i * (i - 1) // 2 > pk, i - 1, i)
    i = np.where((i + 1) * i // 2 <= pk, i + 1, i)
    j = pk - i * (i - 1) // 2
    out = dict(workload=workload, packed_pos=pk)
    for name in ["run"] + sorted({k[:4] for k in z.files if k.startswith("one")}):
        sd, sc = int(z[f"{name}_a0_seed"]), float(z[f"{name}_a0_scale"])
        t0 = time.time()
        orc = O.PGDAttackOracle(w, X, inp["adj"].astype(np.float64), np.zeros((n, n)), fadj, inp["labels"],
                                inp["idx_attack"], cfg)
        orc.w = w                                    # keep the float64 weights (f32() would be a no-op cast anyway)
        orc.set_adj_changes(B.make_a0(n, sd, sc).astype(np.float64))
        orc.step()
        G = orc.last["G_sym"]
        out[f"{name}_g64"] = G[i, j].astype(np.float64)
        out[f"{name}_g64_absmax"] = float(np.abs(G).max())
        ref = z[f"{name}_g"][0].astype(np.float64)
        e = np.abs(ref - out[f"{name}_g64"]).max() / out[f"{name}_g64_absmax"]
        print(name, "reference fp32 vs float64 oracle: max err / gmax =", e, "seconds", round(time.time() - t0, 1), flush=True)
        del orc, G
        np.savez_compressed(os.path.join(OUT, f"{tag}_fp64.npz"), **out)


def mid(tag="mid_s1200_hsic_sparse"):
    """The same for the n = 1200 reference fixture (make_golden.py --only mid): first-step gradient of the float64 oracle
    at the fixture's packed positions -> tests/golden/<tag>_fp64.npz (seconds)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests import helpers as H
    z = H.load_cora(tag)
    n = z["adj"].shape[0]
    O.F32 = np.float64
    f8 = lambda x: np.asarray(x).astype(np.float64)
    w0 = H.weights_from(z)
    w = O.GCNWeights([f8(x) for x in w0.W], [f8(x) for x in w0.b], f8(w0.Wlin), f8(w0.blin))
    cfg = O.AttackConfig(measure=str(z["measure"]), weight_sup=float(z["weight_sup"]),
                         weight_param=tuple(float(x) for x in z["weight_param"]), lr=float(z["lr"]), num_edges=float(z["num_edges"]))
    X = f8(z["features"])
    fadj = 1.0 / (1.0 + np.exp(-np.maximum(X @ X.T - np.eye(n), 0)))
    orc = O.PGDAttackOracle(w, X, f8(z["adj"]), np.zeros((n, n)), fadj, z["labels"], z["idx_attack"], cfg)
    orc.w = w
    orc.set_adj_changes(f8(H.init_adj_changes(n, z["a0_seed"], z["a0_scale"])))
    orc.step()
    pi, pj = H.tril_pos(z["packed_pos"])
    G = orc.last["G_sym"]
    g64 = G[pi, pj].astype(np.float64)
    gmax = float(np.abs(G).max())
    print(tag, "reference fp32 vs float64 oracle: max err / gmax =", np.abs(z["step_g"][0] - g64).max() / gmax, flush=True)
    np.savez_compressed(os.path.join(OUT, f"{tag}_fp64.npz"), packed_pos=z["packed_pos"], step0_g64=g64, step0_g64_absmax=gmax)


def readme(only_missing=False):
    """The same for the README-line fixtures (make_golden.py --only readme): first-step gradient of the float64 oracle at each
    fixture's packed positions and the AUC of the float64 run to its end -> tests/golden/readme_fp64.npz (`<fixture>_g64` as
    float32, `<fixture>_gmax`, `<fixture>_auc64`); ~10 min."""
    from tests import helpers as H
    O.F32 = np.float64
    f8 = lambda x: np.asarray(x).astype(np.float64)
    out = {}
    path = os.path.join(OUT, "readme_fp64.npz")
    if only_missing and os.path.exists(path):      # `readme+`: keep what the file holds, add the fixtures it does not know yet
        old = np.load(path)
        out = {k: old[k] for k in old.files}
    for name in H.readme_cases():
        if f"{name}_auc64" in out:
            continue
        z = H.load_readme(name)
        n = len(z["labels"])
        w0 = H.weights_from(z)
        w = O.GCNWeights([f8(x) for x in w0.W], [f8(x) for x in w0.b], f8(w0.Wlin), f8(w0.blin))
        X = f8(z["features"])
        if str(z["dataset"]) in ("cora", "citeseer", "AIDS"):                      # main.dot_product_decode (main.py:44-55)
            fadj = 1.0 / (1.0 + np.exp(-np.maximum(X @ X.T - np.eye(n), 0)))
        else:
            Xn = X / np.maximum(np.sqrt((X ** 2).sum(1, keepdims=True)), 1e-12)
            fadj = np.maximum(Xn @ Xn.T - np.eye(n), 0)
        orc = O.PGDAttackOracle(w, X, f8(z["adj"]), np.zeros((n, n)), fadj, z["labels"], z["idx_attack"], H.cfg_from(z))
        orc.w = w
        if H.a0_of(z) is not None:
            orc.set_adj_changes(f8(H.a0_of(z)))
        pi, pj = H.tril_pos(z["packed_pos"])
        for t in range(int(z["epochs"])):
            nz = H.noise_of(z, t)
            orc.step(noise=f8(nz)) if nz is not None else orc.step()
            if t == 0:
                G = orc.last["G_sym"]
                g64, gmax = G[pi, pj].astype(np.float64), float(np.abs(G).max())
                out[f"{name}_g64"] = g64.astype(np.float32)
                out[f"{name}_gmax"] = gmax
        # the run to its end in float64: how far the reference's own AUC is from the exact dynamics on this line (Adam turns
        # rounding noise on near-zero gradients into +-lr moves, and the ensemble's entries sit within 1e-6 of each other on
        # some lines: AIDS line 174 differs by 2.7e-3)
        use = [bool(u) for u in z["use"]]
        lab = z["labels"]
        final = orc.finalize(str(z["dataset"]), use[0], use[1], use[2], f8(lab[:, None] == lab[None, :]), f8(z["H_A2"]), f8(z["Y_A"]))
        out[f"{name}_auc64"] = O.metric_pool(z["adj"], np.asarray(final, np.float64), z["idx_attack"])
        print(name, "reference fp32 vs float64 oracle: first gradient, max err / gmax =", np.abs(z["step_g"][0] - g64).max() / gmax,
              " AUC", float(z["auc"]), "vs", out[f"{name}_auc64"], flush=True)
    np.savez_compressed(os.path.join(OUT, "readme_fp64.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "mid":
        mid()
    elif len(sys.argv) > 1 and sys.argv[1] in ("readme", "readme+"):
        readme(only_missing=sys.argv[1] == "readme+")
    else:
        main()
