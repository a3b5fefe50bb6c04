"""The README.md command lines of the reference -- every one its CPU path can run, on cora (n = 2708), citeseer (3312),
polblogs (1490, no attributes), usair (1190, identity attributes), brazil (131, self loops) and AIDS (1429, four real-valued
attributes) -- against fixtures of the reference's own runs (tests/golden/make_golden.py:gen_readme: its Dataset,
preprocess, GCN.fit, six steps of PGDAttack.attack, the dataset's branch of dot_product_decode2, AUC).  Measures MSELoss /
KL / DP / CKA / HSIC, every prior combination of those lines, eps != 0 (recorded noise), lr from 1e-3 to 1.  Run with -m gpu."""
import argparse
import os

import numpy as np
import pytest

from oracle import mcgra_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu

CASES = H.readme_cases()
TRUTH = np.load(os.path.join(H.GOLDEN, "readme_fp64.npz"))


@pytest.fixture(scope="module")
def pkg():
    import mcgra_loader
    p = mcgra_loader.load()
    p._lib.require_device()
    return p


def _args(z):
    use = [bool(u) for u in z["use"]]
    return argparse.Namespace(max_eval=100, lr=0, dataset=str(z["dataset"]), eps=float(z["eps"]), measure=str(z["measure"]),
                              useH_A=use[0], useY_A=use[1], useY=use[2])


def _final_checks(z, final, name):
    sp = z["sample_pos"]
    ref = z["final_sample"].astype(np.float64)
    got = final[sp[:, 0], sp[:, 1]].astype(np.float64)
    scale = np.abs(ref).max()
    # entries whose gradient sits at the fp32 noise level move by +-lr on noise (Adam): a small fraction, off by a few lr
    assert np.mean(np.abs(got - ref) > 2e-2 * max(1.0, scale)) < 0.01
    assert abs(final.astype(np.float64).sum() - float(z["final_sum"])) <= 1e-4 * abs(float(z["final_sum"]))
    auc = O.metric_pool(z["adj"], final, z["idx_attack"])
    # north_star's bar, 1e-4 -- on 38 of the 40 lines.  On AIDS lines 158 and 174 (MSELoss with Y_A: entries of the ensemble
    # sit within 1e-6 of each other) the reference's own AUC is 3.0e-3 / 2.7e-3 away from the float64 run of the same
    # algorithm (tests/golden/make_truth64.py readme): there the bar is the reference's own distance from it
    bar = max(1e-4, abs(float(z["auc"]) - float(TRUTH[f"{name}_auc64"])))
    assert abs(auc - float(z["auc"])) <= bar, (auc, float(z["auc"]), bar)


@pytest.mark.parametrize("name", CASES)
def test_readme_line_engine_against_reference(pkg, name):
    """Engine level (C ABI): the first gradient from the reference's own start against the float64 evaluation of the same
    algorithm and against the reference, both within the reference's own distance from the exact gradient + 3e-4; then
    free-running steps with the reference's recorded noise, the post-loop ensemble in the dataset's decode branch, AUC within
    north_star's 1e-4."""
    import torch
    from mc_gra_amd.topology_attack import _decode_mode
    z = H.load_readme(name)
    n = len(z["labels"])
    wp = [float(x) for x in z["weight_param"]]
    if not (z["feature_adj"].max() != z["feature_adj"].min()):      # the host layer's rule (topology_attack.py:212)
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
            ref = z["step_g"][0].astype(np.float64)
            err_true, ref_true, err = np.abs(g - g64).max() / gmax, np.abs(ref - g64).max() / gmax, np.abs(g - ref).max() / gmax
            # (usair line 96: c9 = 3.9e8 beside a gradient of 79 -- every fp32 evaluation sits 4.5e-4 from the exact one)
            assert err_true <= ref_true + 3e-4 and err <= ref_true + 3e-4, (name, err_true, err, ref_true)
    use = [bool(u) for u in z["use"]]
    lab = z["labels"]
    label_adj = (lab[:, None] == lab[None, :]).astype(np.float32)
    final = eng.finalize(_decode_mode(_args(z)), z["H_A2"] if use[0] else None, z["Y_A"] if use[1] else None,
                         label_adj if use[2] else None).cpu().numpy()
    _final_checks(z, final, name)
    eng.close()


def _class_run(pkg, z, epochs, noise_seed=None, monkeypatch=None):
    """PGDAttack.attack on a README line as main.py drives it; returns modified_adj (numpy).  noise_seed (an eps != 0 line): the
    class's device-side torch.randn draw of adding_noise (topology_attack.py:474-478) is handed the seeded stream the reference's
    run was handed (tests/helpers.py:seeded_noise), one matrix per step."""
    import torch
    from mc_gra_amd import engine as E
    if noise_seed is not None:
        from mc_gra_amd import topology_attack as TA
        n_, draws, real_randn = len(z["labels"]), [], torch.randn

        def seeded_randn(*shape, **kw):
            if tuple(shape) == (n_, n_):
                draws.append(len(draws))
                return torch.as_tensor(H.seeded_noise(noise_seed, draws[-1], (n_, n_)), device=kw.get("device", "cpu"))
            return real_randn(*shape, **kw)

        monkeypatch.setattr(TA.torch, "randn", seeded_randn)
    w = H.weights_from(z)
    victim, emb = H.FakeGCN(w), H.FakeGCN(w)
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda:0")
    Y_A, H_A2 = E.gcn_forward(dev(z["features"]), dev(z["adj"]), [dev(x) for x in w.W], [dev(x) for x in w.b], dev(w.Wlin),
                              dev(w.blin), emb_nlayer=2)
    assert np.abs(Y_A.cpu().numpy() - z["Y_A"]).max() <= 1e-4 * max(1.0, np.abs(z["Y_A"]).max())
    model = pkg.PGDAttack(model=victim, embedding=emb, H_A=H_A2, Y_A=Y_A, nnodes=len(z["labels"]), loss_type="CE", device="cuda:0")
    if H.a0_of(z) is not None:
        model.adj_changes = H.a0_of(z)
    lab = z["labels"]
    model.attack(_args(z), None, float(z["lr"]), 0, float(z["weight_sup"]), tuple(z["weight_param"]), z["feature_adj"], 0, 0, 0,
                 None, None, z["idx_test"], z["adj"], z["features"], np.zeros_like(z["adj"]), lab, z["idx_attack"],
                 float(z["num_edges"]), 0, epochs=epochs, label_adj=(lab[:, None] == lab[None, :]).astype(np.float32))
    _check_path(z, model.history["path"], epochs)
    return model.modified_adj.cpu().numpy()


def _check_path(z, path, epochs):
    """calc = MSELoss (16 README lines) and calc = calc_kl (13 lines; round 6) run the fused elementwise step of attack_fused.hip on
    every graph the tail's 64 x 64 tile pairs cover (n >= 256: all datasets but brazil, n = 131) when eps == 0: every step of such a
    line is a fused one, none falls back."""
    if str(z["measure"]) in ("MSELoss", "KL") and float(z["eps"]) == 0 and len(z["labels"]) >= 256:
        assert path["fused_steps"] == epochs and path["general_steps"] == 0, (str(z["measure"]), path)


@pytest.mark.parametrize("name", [c for c in CASES if "_eps" not in c])
def test_readme_line_through_the_class(pkg, name):
    """The same lines through PGDAttack.attack as main.py drives it (host layer: the constant-feature_adj rule of :212, the
    dataset -> decode branch mapping, label_adj); the eps != 0 lines: test_readme_eps_line_through_the_class."""
    z = H.load_readme(name)
    _final_checks(z, _class_run(pkg, z, int(z["epochs"])), name)


HORIZONS = {ep: np.load(os.path.join(H.GOLDEN, f"horizon{ep}_readme.npz")) for ep in (20, 100)
            if os.path.exists(os.path.join(H.GOLDEN, f"horizon{ep}_readme.npz"))}


@pytest.mark.parametrize("epochs,name", [(ep, c) for ep, hz in sorted(HORIZONS.items()) for c in CASES if f"{c}_auc" in hz.files])
def test_readme_line_at_a_longer_horizon(pkg, epochs, name, monkeypatch):
    """The same lines for 20 and for 100 epochs -- the README's own horizon (main.py:80); the per-step fixtures above run six --
    against the reference's own run of that length from the same trained victim (tests/golden/make_golden.py:
    gen_readme_horizon; all 42 lines: the four eps != 0 ones on seeded noise handed to the class's draw).  The reference was also run in float64 there -- its own code, same inputs: |AUC - AUC64| is what its
    arithmetic leaves of "the" AUC at that horizon (Adam turns rounding noise on near-zero gradients into +-lr moves).  Bar:
    north_star's 1e-4 around the reference's run, or -- where its two evaluations are further apart -- around the interval
    they span.  Measured (profiles/r05_readme_horizon*.txt): at 20 epochs 36 of 38 lines within 3.4e-5 (AIDS 158 / 174,
    MSELoss with Y_A: 1.4e-3 / 2.7e-4, the reference's two runs 2.8e-2 / 1.8e-2 apart); at 100 epochs 35 lines within 6.4e-5,
    polblogs 85 (CKA) 3.6e-4 where the reference's two runs are 3.9e-4 apart (3e-5 from the float64 one), the two AIDS lines
    2.7e-3 / 1.0e-4 where they are 2.9e-2 / 1.8e-2 apart."""
    hz = HORIZONS[epochs]
    assert int(hz["epochs"]) == epochs
    z = H.load_readme(name)
    seed = int(hz[f"{name}_noise_seed"]) if f"{name}_noise_seed" in hz.files else None      # (the eps != 0 lines: seeded noise)
    final = _class_run(pkg, z, epochs, seed, monkeypatch)
    auc = O.metric_pool(z["adj"], final, z["idx_attack"])
    ref, ref64 = float(hz[f"{name}_auc"]), float(hz[f"{name}_auc64"])
    # within 1e-4 of the reference's run; where its two evaluations are D > 1e-4 apart the AUC is not determined more finely than
    # that: within D + 1e-4 of the reference's run (round 5 allowed 2 D; every line it measured sat inside 1 D), or within 1e-4 of
    # the interval the two span.  One line keeps round 5's 2 D: brazil line 149 -- lr = 1 with eps != 0, seeded noise -- where a
    # third evaluation of the noise-amplifying iteration measured 1.9e-3 with D = 1.2e-3 (the rule of test_cora_readme_100_epochs)
    D = abs(ref - ref64)
    bar = max(1e-4, 2 * D) if name == "readme_brazil_kl_all_eps" else D + 1e-4
    assert abs(auc - ref) <= bar or min(ref, ref64) - 1e-4 <= auc <= max(ref, ref64) + 1e-4, (name, epochs, auc, ref, ref64, D)
    fs = float(hz[f"{name}_final_sum"])      # (the ensemble's sum: 1e-3 where the AUC is determined to 1e-4, 5 % on the noise-amplified lines)
    assert abs(final.astype(np.float64).sum() - fs) <= (1e-3 if D <= 1e-4 else 5e-2) * abs(fs)


@pytest.mark.parametrize("dataset,line,fixture", [
    ("brazil", "--w1=0.001 --w2=0.1 --w6=100 --w7=1000 --w9=0.01 --weight_sup=0 --lr=-1 --useH_A --measure=KL", "readme_brazil_kl_h"),
    ("usair", "--w2=100 --w6=100 --w7=10000 --w9=100 --weight_sup=0 --lr=-3 --useH_A --measure=MSELoss", "readme_usair_mse_h"),
    ("polblogs", "--w1=0.1 --w2=0.01 --w6=100 --w7=10000 --w9=0.001 --lr=-1 --useH_A --useY --measure=HSIC", "readme_polblogs_hsic_hY")])
def test_main_entry_runs_readme_lines_on_the_committed_datasets(pkg, dataset, line, fixture, tmp_path, monkeypatch):
    """`python main.py --dataset=... <README line>` end to end on the reference's own data files (tests/golden/dataset): the
    loader, victim training (this run's own weights: torch on the GPU), priors, the attack and the AUC.  The victim differs
    from the reference's, so the AUC is held to the level of the reference's run of the same line (its 6-step fixture)."""
    from mc_gra_amd import main as M
    root = os.path.join(H.GOLDEN, "dataset")
    monkeypatch.chdir(tmp_path)
    args = M.build_parser().parse_args(["--dataset", dataset, "--dataset_root", root, "--epochs", "6"] + line.split())
    res = M.run(args)
    ref = float(H.load_readme(fixture)["auc"])
    assert abs(res["auc_attack"] - ref) < 0.03, (res, ref)
    assert os.path.exists(tmp_path / "results" / "result.txt")


@pytest.mark.parametrize("name", [c for c in CASES if "_eps" in c])
def test_readme_eps_line_through_the_class(pkg, name, monkeypatch):
    """The `eps != 0` README lines (brazil 149; usair 104 / 112 and cora 120 -- a negative eps -- at n > 1000) through
    PGDAttack.attack: the class draws adding_noise's torch.randn on the device (topology_attack.py:474-478); here that draw is
    handed the noise the reference's run was handed (recorded matrices, or the seeded stream of the large graphs), as
    make_golden.py did to the reference's torch.randn_like -- the host layer then has to reproduce the reference's AUC."""
    import torch
    from mc_gra_amd import engine as E
    from mc_gra_amd import topology_attack as TA
    z = H.load_readme(name)
    n = len(z["labels"])
    w = H.weights_from(z)
    victim, emb = H.FakeGCN(w), H.FakeGCN(w)
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda:0")
    Y_A, H_A2 = E.gcn_forward(dev(z["features"]), dev(z["adj"]), [dev(x) for x in w.W], [dev(x) for x in w.b], dev(w.Wlin),
                              dev(w.blin), emb_nlayer=2)
    model = pkg.PGDAttack(model=victim, embedding=emb, H_A=H_A2, Y_A=Y_A, nnodes=n, loss_type="CE", device="cuda:0")
    draws = []
    real_randn = torch.randn

    def seeded_randn(*shape, **kw):
        if tuple(shape) == (n, n):
            t = len(draws)
            draws.append(t)
            return torch.as_tensor(H.noise_of(z, t), device=kw.get("device", "cpu"))
        return real_randn(*shape, **kw)

    monkeypatch.setattr(TA.torch, "randn", seeded_randn)
    lab = z["labels"]
    model.attack(_args(z), None, float(z["lr"]), 0, float(z["weight_sup"]), tuple(z["weight_param"]), z["feature_adj"], 0, 0, 0,
                 None, None, z["idx_test"], z["adj"], z["features"], np.zeros_like(z["adj"]), lab, z["idx_attack"],
                 float(z["num_edges"]), 0, epochs=int(z["epochs"]), label_adj=(lab[:, None] == lab[None, :]).astype(np.float32))
    assert len(draws) == int(z["epochs"]), "one draw of adding_noise per step"
    _final_checks(z, model.modified_adj.cpu().numpy(), name)
