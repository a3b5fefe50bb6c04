"""BASELINE.json configs[2] on the GPU: Citeseer (N = 3312) through the reference's Dataset, the dense GAT victim
(models/gat.py, 5 heads x 16, trained by the reference's GAT.fit) with embedding_gat sharing its attention layers,
priors H_A + Y, the citeseer branch of dot_product_decode2 (:427-431).  Fixtures: tests/golden/citeseer_gat_*.npz
(tests/golden/make_golden.py --only citeseer ran the reference's PGDAttack.attack on CPU).

"bf16" in that config line has no counterpart in the reference (its CPU path is fp32 torch): on this path 16-bit
matrix-core arithmetic appears only as the 2-plane fp16 split of N x N x N products at fp32-level error (DESIGN.md
section 3) -- the ELU embeddings of width 80 take the Gram evaluation of linear_HSIC, whose four products per step run
on that kernel (n = 3312 >= 1024); KL is elementwise fp32."""
import argparse
import json
import os
import time

import numpy as np
import pytest

from oracle import mcgra_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
CASES = ["citeseer_gat_kl", "citeseer_gat_hsic"]


@pytest.fixture(scope="module")
def pkg():
    import mcgra_loader
    p = mcgra_loader.load()
    p._lib.require_device()
    return p


def _weights(z):
    return O.GCNWeights([z["W0"], z["W1"]], [z["b0"], z["b1"]], z["Wlin"], z["blin"], None, str(z["act"]), str(z["head_act"]))


@pytest.mark.parametrize("name", CASES)
def test_citeseer_gat_engine_matches_reference(pkg, name):
    """Engine level: per-step mirrored gradient on 8k sampled entries (step 0 starts from the reference's own state and
    is held to 3e-4 of the gradient's largest magnitude; later steps free-run), post-loop ensemble with decode_mode 1
    and priors H_A + Y, AUC within 1e-4."""
    import torch
    z = H.load_cora(name)
    w = _weights(z)
    n = z["adj"].shape[0]
    assert n == 3312 and str(z["dataset"]) == "citeseer" and list(z["use"]) == [1, 0, 1]
    dims = [w.W[0].shape[0]] + [x.shape[1] for x in w.W]
    eng = pkg.AttackEngine(n, dims, w.Wlin.shape[0], int(z["emb_nlayer"]), str(z["measure"]), float(z["weight_sup"]),
                           tuple(float(x) for x in z["weight_param"]), float(z["lr"]), float(z["num_edges"]),
                           len(z["idx_attack"]), act="elu", head_act="elu", fin_layers=tuple(int(x) for x in z["fin_layers"]))
    eng.set_model(w.W, w.b, w.Wlin, w.blin)
    eng.set_graph(z["features"], z["adj"], None, H.cora_feature_adj(z["features"]), z["labels"], z["idx_attack"])
    if "a0_seed" in z:
        eng.set_adj_changes(H.init_adj_changes(n, z["a0_seed"], z["a0_scale"]))
    # Y_A the engine computed from the true graph against the reference's (main.py:236).  The H_A2 of the fixture is
    # NOT comparable: main.py:235-241 evaluates embedding_gat before anything puts it in eval mode, so the reference's
    # H_A prior carries dropout noise (gat.py:171); it only enters the post-loop ensemble (:315-316), where the test
    # hands the fixture's own H_A2 to finalize, as the reference run did.  Inside the loop the reference recomputes
    # H_A_cur in eval mode (:124, :243), which is what the engine's "HA" buffer holds.
    assert np.abs(eng.buffer("YA").cpu().numpy() - z["Y_A"]).max() <= 5e-5
    pi, pj = H.tril_pos(z["packed_pos"])
    ti, tj = torch.as_tensor(pi, device="cuda:0"), torch.as_tensor(pj, device="cuda:0")
    lr = float(z["lr"])
    for t in range(int(z["epochs"])):
        eng.step(); eng.monitor()
        Gs = eng.buffer("G_sym")
        g = Gs[ti, tj].cpu().numpy()
        gmax = float(z["step_g_absmax"][t])
        err = np.abs(g - z["step_g"][t]).max() / gmax
        assert err <= (3e-4 if t == 0 else 3e-3), (name, t, err)
        assert abs(float(Gs.abs().max()) - gmax) <= 1e-3 * gmax
        a = eng.buffer("M")[ti, tj].cpu().numpy()
        moved = np.abs(a - np.clip(z["step_a"][t], 0, 1)) > 0.05 * lr
        assert moved.mean() <= 0.01, (name, t, moved.mean())
    lab = z["labels"]
    label_adj = (lab[:, None] == lab[None, :]).astype(np.float32)
    final = eng.finalize(1, z["H_A2"], None, label_adj).cpu().numpy()          # citeseer branch; useH_A, useY
    auc = O.metric_pool(z["adj"], final, z["idx_attack"])
    assert abs(auc - float(z["auc"])) <= 1e-4, (auc, float(z["auc"]))
    sp = z["sample_pos"]
    assert np.mean(np.abs(final[sp[:, 0], sp[:, 1]] - z["final_sample"]) > 2e-2) < 0.01
    assert abs(final.astype(np.float64).sum() - float(z["final_sum"])) <= 2e-4 * abs(float(z["final_sum"]))


def test_citeseer_gat_through_the_pgdattack_class(pkg):
    """Class level, as main.py --dataset citeseer --arch gat --useH_A --useY drives it (README citeseer line, KL)."""
    import torch
    z = H.load_cora("citeseer_gat_kl")
    w = _weights(z)
    victim, emb = H.FakeGAT(w), H.FakeGAT(w)
    n = z["adj"].shape[0]
    args = argparse.Namespace(max_eval=100, lr=0, dataset="citeseer", eps=0, measure=str(z["measure"]), useH_A=True,
                              useY_A=False, useY=True, w1=0, w2=0, w6=0, w7=0, w8=0, w9=0, w10=0)
    model = pkg.PGDAttack(model=victim, embedding=emb, H_A=torch.tensor(z["H_A2"]), Y_A=torch.tensor(z["Y_A"]), nnodes=n,
                          loss_type="CE", device="cuda:0")
    lab = z["labels"]
    model.attack(args, None, float(z["lr"]), 0, float(z["weight_sup"]), tuple(z["weight_param"]),
                 H.cora_feature_adj(z["features"]), 0, 0, 0, None, None, z["idx_test"], z["adj"], z["features"],
                 np.zeros_like(z["adj"]), lab, z["idx_attack"], float(z["num_edges"]), 0, epochs=int(z["epochs"]),
                 label_adj=(lab[:, None] == lab[None, :]).astype(np.float32))
    final = model.modified_adj.cpu().numpy()
    assert abs(O.metric_pool(z["adj"], final, z["idx_attack"]) - float(z["auc"])) <= 1e-4


def test_citeseer_gat_hsic_step_time(pkg):
    """Step time of config[2]'s shape (GAT victim, Gram evaluation on the split kernel).  Written to
    $MCGRA_REPORT_DIR/citeseer_gat_hsic_step.json when that is set (profiles/ keeps a copy); the bound only catches a
    fall back to the fp32 products (measured 3.5x slower at this size)."""
    import torch
    z = H.load_cora("citeseer_gat_hsic")
    w = _weights(z)
    n = z["adj"].shape[0]
    dims = [w.W[0].shape[0]] + [x.shape[1] for x in w.W]
    eng = pkg.AttackEngine(n, dims, w.Wlin.shape[0], int(z["emb_nlayer"]), "HSIC", float(z["weight_sup"]),
                           tuple(float(x) for x in z["weight_param"]), float(z["lr"]), float(z["num_edges"]),
                           len(z["idx_attack"]), act="elu", head_act="elu", fin_layers=tuple(int(x) for x in z["fin_layers"]))
    eng.set_model(w.W, w.b, w.Wlin, w.blin)
    eng.set_graph(z["features"], z["adj"], None, H.cora_feature_adj(z["features"]), z["labels"], z["idx_attack"])
    for _ in range(3):
        eng.step(); eng.monitor()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 20
    for _ in range(steps):
        eng.step(); eng.monitor()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    assert eng.gram_split_steps() == steps + 3
    out = {"workload": "citeseer (N = 3312), GAT victim 5 x 16 ELU, HSIC, priors H_A + Y", "ms_per_step": ms,
           "steps_per_s": 1e3 / ms, "products": "Gram evaluation: 4 split products per step (2-plane fp16)",
           "monitor_forward": True}
    d = os.environ.get("MCGRA_REPORT_DIR")
    if d and os.path.isdir(d):
        json.dump(out, open(os.path.join(d, "citeseer_gat_hsic_step.json"), "w"))
    print(out)
    assert ms < 6.0, ms


def test_bench_gat_shaped_workload_takes_the_gram_evaluation(pkg):
    """bench.py's `citeseer-shape-gat-hsic` (configs[2]'s shape on synthetic data: N = 3312, an 80-wide ELU chain): no low-rank
    forms apply, every step is a general step whose four Gram products run on the split kernel -- the number bench.py reports
    for it under `other_workloads` is that path's."""
    import torch
    import bench
    eng, inp, adj = bench.build_engine(pkg, torch, torch.device("cuda:0"), "citeseer-shape-gat-hsic", 0)
    for _ in range(3):
        eng.step(); eng.monitor()
    assert eng.fused_steps() == 0 and eng.path_stats() == {"lowrank_steps": 0, "general_steps": 3} and eng.gram_split_steps() == 3
    lab = torch.as_tensor(inp["labels"], device="cuda:0")
    final = eng.finalize(0, eng.buffer("HA"), eng.buffer("YA"), (lab[:, None] == lab[None, :]).float())
    assert 0.5 < bench.gpu_auc(adj, final, torch) < 1.0
