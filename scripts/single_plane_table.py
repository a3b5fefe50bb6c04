"""The single-plane product mode (MCGRA_SPLIT_BF16=1) against the default fp32-level split, measured (round 6).

For `synthetic-10k-hsic` (fixtures tests/golden/bench10k_hsic.npz + bench10k_hsic_ref64.npz: the reference's own fp32 run and the
reference's own code in float64, at 8k sampled entries) and for Cora HSIC (tests/golden/cora_hsic_sparse.npz through bench's
Cora-shaped workload is NOT the fixture's graph, so Cora is measured on the fixture's own inputs): per product mode
  * first-gradient error against float64 (max and rms, of the gradient's largest magnitude),
  * sign flips against float64 (Adam's first step is lr * sign(g): the reference's own fp32 gradient flips 0.2 % at N = 10 000),
  * recovered-adjacency AUC after 4 and after 20 steps and its difference from the default mode's,
  * ms per product launch (HIP events on the product's stream, in situ) and ms per step.
Writes one text table to stdout (committed as profiles/r06_single_plane_table.txt).  Test infrastructure: uses the fixtures only."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MCGRA_KEEP_GSYM", "1")
os.environ.setdefault("MCGRA_AB", "1")
import mcgra_loader  # noqa: E402

pkg = mcgra_loader.load()
import torch  # noqa: E402
import bench  # noqa: E402
from tests import helpers as H  # noqa: E402
from oracle import mcgra_oracle as O  # noqa: E402  (checker only: metric_pool for the Cora AUC)

dev = torch.device("cuda:0")
MODES = (("3", "fp16 x 2 planes, 3 products (default)"), ("1", "fp16, 1 plane product (MCGRA_SPLIT_BF16=1)"))


def with_mode(mode, fn):
    old = os.environ.get("MCGRA_SPLIT_BF16")
    os.environ["MCGRA_SPLIT_BF16"] = mode
    try:
        return fn()
    finally:
        if old is None:
            del os.environ["MCGRA_SPLIT_BF16"]
        else:
            os.environ["MCGRA_SPLIT_BF16"] = old


def timed(eng, steps):
    eng.profile(True); eng.gemm_stats(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(); eng.monitor()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = eng.gemm_stats(reset=True); eng.profile(False)
    return 1e3 * dt / steps, (st["ms"] / st["launches"]) if st["launches"] else float("nan")


def bench10k():
    WL = "synthetic-10k-hsic"
    z = np.load(os.path.join(ROOT, "tests", "golden", "bench10k_hsic.npz"))
    z64 = np.load(os.path.join(ROOT, "tests", "golden", "bench10k_hsic_ref64.npz"))
    pi, pj = H.tril_pos(z["packed_pos"])
    ti, tj = torch.as_tensor(pi, device=dev), torch.as_tensor(pj, device=dev)
    g64, gref, gmax = z64["run_g64ref"].astype(np.float64), z["run_g"][0].astype(np.float64), float(z["run_g_absmax"][0])
    rows, aucs = [], {}
    print(f"== {WL}: N = 10 000, first step from the bench's own start; truth = the reference's own code in float64 (8k sampled entries)")
    print(f"   the reference's own fp32 gradient: max {np.abs(gref - g64).max() / gmax:.2e}  rms {np.sqrt(np.mean((gref - g64) ** 2)) / gmax:.2e}  "
          f"sign flips {100 * float((np.sign(gref) != np.sign(g64)).mean()):.3f} %")
    for mode, label in MODES:
        eng, inp, adj_dev = with_mode(mode, lambda: bench.build_engine(pkg, torch, dev, WL, int(z["seed"])))
        assert eng.product_mode() == int(mode)
        lab = torch.as_tensor(inp["labels"], device=dev)
        la = (lab[:, None] == lab[None, :]).float()
        eng.step(); eng.monitor()
        g = eng.buffer("G_sym")[ti, tj].cpu().numpy().astype(np.float64)
        emax, erms = np.abs(g - g64).max() / gmax, np.sqrt(np.mean((g - g64) ** 2)) / gmax
        flips = 100 * float((np.sign(g) != np.sign(g64)).mean())
        res = {}
        for upto in (4, 20):
            while eng.fused_steps() < upto:
                eng.step(); eng.monitor()
            res[upto] = bench.gpu_auc(adj_dev, eng.finalize(0, eng.buffer("HA"), eng.buffer("YA"), la), torch)
        ms_step, ms_prod = timed(eng, 40)
        aucs[mode] = res
        rows.append((label, emax, erms, flips, res[4], res[20], ms_prod, ms_step))
        del eng
        torch.cuda.empty_cache()
    print(f"   {'product':<46} {'max err':>9} {'rms err':>9} {'flips %':>8} {'AUC@4':>10} {'AUC@20':>10} {'ms/product':>11} {'ms/step':>8}")
    for r in rows:
        print(f"   {r[0]:<46} {r[1]:9.2e} {r[2]:9.2e} {r[3]:8.3f} {r[4]:10.6f} {r[5]:10.6f} {r[6]:11.3f} {r[7]:8.3f}")
    print(f"   AUC delta (single plane - default): {aucs['1'][4] - aucs['3'][4]:+.2e} at 4 steps, {aucs['1'][20] - aucs['3'][20]:+.2e} at 20 steps; "
          f"the reference's AUC at 4 steps: {float(z['run_auc']):.6f}")


def cora():
    z = H.load_cora("cora_hsic_sparse")
    zz = dict(z, nlayer=np.array(2), emb_nlayer=np.array(2), feature_adj=H.cora_feature_adj(z["features"]))
    rows, aucs = [], {}
    orc = H.oracle_from(zz)      # float64 would be the truth; the fp32 oracle stands in for the reference here (pinned to it by the fixture)
    orc.step()
    g_or = orc.last["G_sym"].astype(np.float64)
    gmax = np.abs(g_or).max()
    print(f"\n== Cora (n = {z['adj'].shape[0]}), HSIC from the sparse start, the fixture's reference-trained victim; first gradient against the fp32 oracle")
    lab = z["labels"]
    la = (lab[:, None] == lab[None, :]).astype(np.float32)
    for mode, label in MODES:
        eng = with_mode(mode, lambda: H.engine_from(pkg, zz))
        assert eng.product_mode() == int(mode)
        eng.step(); eng.monitor()
        g = eng.buffer("G_sym").cpu().numpy().astype(np.float64)
        emax, erms = np.abs(g - g_or).max() / gmax, np.sqrt(np.mean((g - g_or) ** 2)) / gmax
        off = ~np.eye(g.shape[0], dtype=bool)
        flips = 100 * float((np.sign(g[off]) != np.sign(g_or[off])).mean())
        res = {}
        for upto in (4, 20):
            while eng.fused_steps() < upto:
                eng.step(); eng.monitor()
            final = eng.finalize(0, eng.buffer("HA"), eng.buffer("YA"), la).cpu().numpy()
            res[upto] = O.metric_pool(z["adj"], final, z["idx_attack"])
        ms_step, ms_prod = timed(eng, 100)
        aucs[mode] = res
        rows.append((label, emax, erms, flips, res[4], res[20], ms_prod, ms_step))
        eng.close()
    print(f"   {'product':<46} {'max err':>9} {'rms err':>9} {'flips %':>8} {'AUC@4':>10} {'AUC@20':>10} {'ms/product':>11} {'ms/step':>8}")
    for r in rows:
        print(f"   {r[0]:<46} {r[1]:9.2e} {r[2]:9.2e} {r[3]:8.3f} {r[4]:10.6f} {r[5]:10.6f} {r[6]:11.3f} {r[7]:8.3f}")
    print(f"   AUC delta (single plane - default): {aucs['1'][4] - aucs['3'][4]:+.2e} at 4 steps, {aucs['1'][20] - aucs['3'][20]:+.2e} at 20 steps")


def citeseer_gat():
    """BASELINE.json configs[2]: Citeseer (n = 3312), the reference-trained GAT victim, HSIC -- the general step, whose Gram
    evaluation's FOUR N x N x N products per step run on the same kernel; against the reference's own run (fixture)."""
    z = H.load_cora("citeseer_gat_hsic")
    w = O.GCNWeights([z["W0"], z["W1"]], [z["b0"], z["b1"]], z["Wlin"], z["blin"], None, str(z["act"]), str(z["head_act"]))
    n = z["adj"].shape[0]
    dims = [w.W[0].shape[0]] + [x.shape[1] for x in w.W]
    pi, pj = H.tril_pos(z["packed_pos"])
    ti, tj = torch.as_tensor(pi, device=dev), torch.as_tensor(pj, device=dev)
    lab = z["labels"]
    la = (lab[:, None] == lab[None, :]).astype(np.float32)
    fadj = H.cora_feature_adj(z["features"])
    print(f"\n== Citeseer + GAT (BASELINE.json configs[2]; n = {n}, 5 x 16 = 80 wide, ELU), HSIC: the Gram evaluation, four products per step; "
          f"against the reference's own fp32 run (its AUC: {float(z['auc']):.6f})")
    rows = []
    for mode, label in MODES:
        def mk():
            e = pkg.AttackEngine(n, dims, w.Wlin.shape[0], int(z["emb_nlayer"]), "HSIC", float(z["weight_sup"]),
                                 tuple(float(x) for x in z["weight_param"]), float(z["lr"]), float(z["num_edges"]), len(z["idx_attack"]),
                                 act="elu", head_act="elu", fin_layers=tuple(int(x) for x in z["fin_layers"]))
            e.set_model(w.W, w.b, w.Wlin, w.blin)
            e.set_graph(z["features"], z["adj"], None, fadj, lab, z["idx_attack"])
            if "a0_seed" in z:
                e.set_adj_changes(H.init_adj_changes(n, z["a0_seed"], z["a0_scale"]))
            return e
        eng = with_mode(mode, mk)
        assert (eng.product_mode() == 1) == (mode == "1"), eng.product_mode()      # (a victim without a low-rank form reports 0 / 1: its products are the Gram evaluation's)
        errs = []
        for t in range(int(z["epochs"])):
            eng.step(); eng.monitor()
            g = eng.buffer("G_sym")[ti, tj].cpu().numpy().astype(np.float64)
            errs.append(np.abs(g - z["step_g"][t]).max() / float(z["step_g_absmax"][t]))
        final = eng.finalize(1, z["H_A2"], None, la).cpu().numpy()
        auc = O.metric_pool(z["adj"], final, z["idx_attack"])
        ms_step, ms_prod = timed(eng, 60)
        rows.append((label, errs[0], max(errs), auc, auc - float(z["auc"]), ms_prod, ms_step))
        eng.close()
    print(f"   {'products':<46} {'step-0 err':>10} {'worst step':>10} {'AUC':>10} {'- reference':>12} {'ms/product':>11} {'ms/step':>8}")
    for r in rows:
        print(f"   {r[0]:<46} {r[1]:10.2e} {r[2]:10.2e} {r[3]:10.6f} {r[4]:+12.2e} {r[5]:11.3f} {r[6]:8.3f}")


if __name__ == "__main__":
    bench10k()
    cora()
    citeseer_gat()
