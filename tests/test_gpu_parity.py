"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the
numpy oracle and the reference-generated golden fixtures.  Run with -m gpu."""
import os

import numpy as np
import pytest

from oracle import mcgra_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu

OPS = np.load(os.path.join(H.GOLDEN, "ops.npz"))


@pytest.fixture(scope="module")
def pkg():
    import mcgra_loader
    p = mcgra_loader.load()
    p._lib.require_device()
    return p


@pytest.fixture(scope="module")
def torch_():
    import torch
    assert torch.cuda.is_available()
    return torch


def dev(t, x):
    return t.as_tensor(np.ascontiguousarray(x), device="cuda:0")


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (130, 67, 45), (300, 257, 129), (1000, 16, 1000),
                                   (16, 16, 3000), (2708, 7, 2708), (515, 515, 515), (64, 200, 1)])
def test_sgemm_matches_fp64(pkg, torch_, ta, tb, m, n, k):
    rng = np.random.RandomState(m * 7 + n * 3 + k + 2 * ta + tb)
    A = rng.randn(*((k, m) if ta else (m, k))).astype(np.float32)
    B = rng.randn(*((n, k) if tb else (k, n))).astype(np.float32)
    from mc_gra_amd import engine as E
    C = E.sgemm(dev(torch_, A), dev(torch_, B), ta=bool(ta), tb=bool(tb)).cpu().numpy()
    ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
    # fp32 fmaf chain: error <= ~1e-7 * sum|a b| (guide: 0.75-1.5e-7 at K <= 1024)
    bound = 4e-7 * (np.abs(A.T if ta else A).astype(np.float64) @ np.abs(B.T if tb else B).astype(np.float64))
    assert np.all(np.abs(C - ref) <= bound + 1e-30)


@pytest.mark.parametrize("m,n,k,beta", [(256, 256, 16, 0.0), (1000, 777, 33, 1.0), (300, 2708, 64, -0.5),
                                        (515, 301, 7, 2.0), (2708, 2708, 32, 1.0)])
def test_rank_k_update_path(pkg, torch_, m, n, k, beta):
    """C = beta C + alpha A B^T with K <= 64 goes through the HBM-bound rank-k kernel (rankk_f32.hip);
    odd K / odd N exercise its scalar staging and tail stores; operands with a leading dimension > K too."""
    from mc_gra_amd import engine as E
    rng = np.random.RandomState(m + n + k)
    Abig = rng.randn(m, k + 5).astype(np.float32)
    A = Abig[:, :k]
    B = rng.randn(n, k).astype(np.float32)
    C0 = rng.randn(m, n).astype(np.float32)
    got = E.sgemm(dev(torch_, Abig)[:, :k], dev(torch_, B), tb=True, alpha=1.5, beta=beta,
                  out=dev(torch_, C0).clone()).cpu().numpy()
    ref = 1.5 * (A.astype(np.float64) @ B.astype(np.float64).T) + beta * C0
    bound = 4e-7 * (1.5 * np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T + abs(beta) * np.abs(C0)) + 1e-30
    assert np.all(np.abs(got - ref) <= bound)


def test_sgemm_identity_asymmetric_and_beta(pkg, torch_):
    from mc_gra_amd import engine as E
    n = 200
    B = (np.arange(n * n, dtype=np.float32).reshape(n, n) % 97) - 48      # asymmetric, exact in fp32
    I = np.eye(n, dtype=np.float32)
    out = E.sgemm(dev(torch_, I), dev(torch_, B)).cpu().numpy()
    assert np.array_equal(out, B)
    out = E.sgemm(dev(torch_, B), dev(torch_, I), tb=True).cpu().numpy()
    assert np.array_equal(out, B)
    C0 = np.ones((n, n), np.float32)
    out = E.sgemm(dev(torch_, I), dev(torch_, B), alpha=2.0, beta=3.0, out=dev(torch_, C0).clone()).cpu().numpy()
    assert np.array_equal(out, 2 * B + 3)
    # non-contiguous (ld > cols) operands and unaligned base pointers (scalar-load path)
    big = np.random.RandomState(0).randn(300, 301).astype(np.float32)
    tb_ = dev(torch_, big)
    Av, Bv = tb_[1:201, 1:130], tb_[3:132, 2:99]
    ref = big[1:201, 1:130].astype(np.float64) @ big[3:132, 2:99].astype(np.float64)
    got = E.sgemm(Av, Bv).cpu().numpy()
    assert rel(got, ref) < 1e-5


@pytest.mark.parametrize("n,k", [(128, 64), (300, 77), (515, 515), (1000, 40), (2708, 300)])
def test_syrk_symm_lower_tile_storage(pkg, torch_, n, k):
    """SYRK writes only tiles on/below the diagonal; SYMM reads only those and must equal S @ B."""
    from mc_gra_amd import engine as E
    rng = np.random.RandomState(n + k)
    A = rng.randn(n, k).astype(np.float32)
    B = rng.randn(n, 200).astype(np.float32)
    ref = A.astype(np.float64) @ A.astype(np.float64).T
    poison = np.full((n, n), np.nan, np.float32)
    out = E.ssyrk_lower(dev(torch_, A), out=dev(torch_, poison).clone()).cpu().numpy()
    i, j = np.indices((n, n))
    valid = j < (i // 128 + 1) * 128
    assert np.all(np.isnan(out[~valid]))                      # untouched outside lower tile storage
    # diagonal entries are sums of squares (no cancellation): sequential fp32 accumulation error ~ sqrt(K) * 2^-24
    bound = max(4e-7, 2e-7 * np.sqrt(k)) * (np.abs(A).astype(np.float64) @ np.abs(A).astype(np.float64).T)
    assert np.all(np.abs(out[valid] - ref[valid]) <= bound[valid] + 1e-30)
    # bitwise symmetric where both (i,j) and (j,i) are stored (diagonal tiles): same k order, commutative products
    dmask = valid & valid.T
    assert np.array_equal(out[dmask], out.T[dmask])
    # SYMM on the NaN-poisoned upper part: must never read it
    got = E.ssymm_lower(dev(torch_, out), dev(torch_, B)).cpu().numpy()
    S = np.where(valid, out, out.T).astype(np.float64)
    refp = S @ B.astype(np.float64)
    assert np.all(np.isfinite(got))
    assert np.abs(got - refp).max() <= 1e-5 * np.abs(refp).max()


# ------------------------------------------------------------------ standalone ops
def test_ops_against_reference_goldens(pkg, torch_):
    from mc_gra_amd import engine as E
    t = torch_
    out = E.normalize_adj_tensor(dev(t, OPS["norm_in"])).cpu().numpy()
    assert rel(out, OPS["norm_out"]) < 1e-6
    assert abs(float(E.info_entropy(dev(t, OPS["ie_in"]))) - float(OPS["ie_val"])) < 1e-6
    got = E.dot_product_decode(dev(t, OPS["dd2_Z"])).cpu().numpy()
    assert rel(got, OPS["dd_out"]) < 2e-6
    for tag in ("a", "b"):
        X, Y = OPS[f"hsic_{tag}_X"], OPS[f"hsic_{tag}_Y"]
        v = float(E.linear_hsic(dev(t, X), dev(t, Y)))
        assert abs(v - float(OPS[f"hsic_{tag}_val"])) <= 3e-5 * abs(float(OPS[f"hsic_{tag}_val"]))
    v = float(E.mse(dev(t, OPS["mse_X"]), dev(t, OPS["mse_Y"])))
    assert abs(v - float(OPS["mse_val"])) <= 1e-6 * abs(float(OPS["mse_val"]))
    got = E.get_modified_adj(dev(t, OPS["gma_a"]), dev(t, OPS["gma_ori"]), 30).cpu().numpy()
    assert np.array_equal(got, OPS["gma_out"])            # data movement: bit exact


@pytest.mark.parametrize("sg", [1.0, 5.0])
def test_gaussian_hsic_ops(pkg, torch_, sg):
    """hsic.py hsic_regular / hsic_normalized against the reference's own values."""
    from mc_gra_amd import engine as E
    x, y = dev(torch_, OPS["ghsic_x"]), dev(torch_, OPS["ghsic_y"])
    ref_r, ref_n = float(OPS[f"ghsic_reg_{sg}"]), float(OPS[f"ghsic_norm_{sg}"])
    assert abs(float(E.hsic_regular(x, y, sg)) - ref_r) <= 5e-5 * abs(ref_r) + 1e-9
    assert abs(float(E.hsic_normalized(x, y, sg)) - ref_n) <= 3e-4 * abs(ref_n)
    with pytest.raises(NotImplementedError):
        E.hsic_regular(x, y, 0.0)


@pytest.mark.parametrize("m,dx,dy", [(257, 9, 5), (64, 3, 3), (1, 4, 2)])
def test_hsic_normalized_cca_against_float64(pkg, torch_, m, dx, dy):
    """hsic_normalized_cca at sizes beyond the reference fixture (m not a multiple of anything, m = 1): the device path
    (fp64 kernel matrices, Gauss-Jordan inverses with partial pivoting) against the oracle's float64 evaluation."""
    from mc_gra_amd import hsic as HS
    rng = np.random.RandomState(m)
    x = rng.randn(m, dx).astype(np.float32)
    y = (x[:, :min(dx, dy)] @ rng.randn(min(dx, dy), dy) * 0.5 + rng.randn(m, dy) * 0.7).astype(np.float32)
    for sg in (0.8, 3.0):
        want = float(O.hsic_normalized_cca(x, y, sg, dtype=np.float64))
        got = float(HS.hsic_normalized_cca(dev(torch_, x), dev(torch_, y), sigma=sg))
        assert abs(got - want) <= 2e-6 * abs(want) + 1e-7, (m, sg, got, want)


def test_hsic_py_mirror_against_reference(pkg, torch_):
    """mc-gra_amd/hsic.py (the reference's hsic.py surface on the device) against the reference's own values:
    sigma=None forms (median heuristic: distance matrix on the device, median on the host as in the reference),
    distmat, distcorr, mmd, mmd_pxpy_pxy, hsic_normalized_cca."""
    from mc_gra_amd import hsic as HS
    x, y, z = dev(torch_, OPS["ghsic_x"]), dev(torch_, OPS["ghsic_y"]), dev(torch_, OPS["ghsic_z"])
    assert abs(HS.sigma_estimation(x, x) - float(OPS["ghsic_sigma_xx"])) <= 2e-5 * float(OPS["ghsic_sigma_xx"])
    assert abs(HS.sigma_estimation(y, z) - float(OPS["ghsic_sigma_yz"])) <= 2e-5 * float(OPS["ghsic_sigma_yz"])
    assert rel(HS.distmat(x).cpu().numpy(), OPS["ghsic_distmat"]) < 2e-6
    for sg in (1.0, 5.0):
        assert abs(float(HS.hsic_regular(x, y, sigma=sg)) - float(OPS[f"ghsic_reg_{sg}"])) <= 3e-5 * abs(float(OPS[f"ghsic_reg_{sg}"]))
    assert abs(float(HS.hsic_regular(x, y)) - float(OPS["ghsic_reg_auto"])) <= 2e-4 * abs(float(OPS["ghsic_reg_auto"]))
    assert abs(float(HS.hsic_normalized(x, y)) - float(OPS["ghsic_norm_auto"])) <= 3e-4 * abs(float(OPS["ghsic_norm_auto"]))
    assert abs(float(HS.distcorr(x, 2.0)) - float(OPS["ghsic_distcorr_2.0"])) <= 2e-6
    for sg in (None, 1.5):
        assert abs(float(HS.mmd(y, z, sigma=sg)) - float(OPS[f"ghsic_mmd_{sg}"])) <= 3e-5 * abs(float(OPS[f"ghsic_mmd_{sg}"]))
        assert abs(float(HS.mmd_pxpy_pxy(x, y, sigma=sg)) - float(OPS[f"ghsic_mmdp_{sg}"])) <= 3e-4 * abs(float(OPS[f"ghsic_mmdp_{sg}"])) + 1e-8
    # hsic_normalized_cca: two ill-conditioned inverses.  The fixture holds the reference's fp32 value and the same
    # formula in float64; the device path (fp64 throughout) must sit on the float64 value and be no further from the
    # reference than the reference is from the float64 value.
    for sg in (1.0, 5.0, None):
        ref, exact = float(OPS[f"ghsic_cca_{sg}"]), float(OPS[f"ghsic_cca64_{sg}"])
        got = float(HS.hsic_normalized_cca(x, y, sigma=sg))
        assert abs(got - exact) <= (2e-6 if sg else 2e-4) * abs(exact), (sg, got, exact)      # (None: sigma from an fp32 median)
        assert abs(got - ref) <= abs(ref - exact) + (2e-6 if sg else 2e-4) * abs(exact), (sg, got, ref, exact)


def test_ops_edge_cases(pkg, torch_):
    from mc_gra_amd import engine as E
    t = torch_
    # isolated nodes / all-zero adjacency: d = 1, adj_norm = I
    z = np.zeros((9, 9), np.float32)
    assert np.array_equal(E.normalize_adj_tensor(dev(t, z)).cpu().numpy(), np.eye(9, dtype=np.float32))
    # ragged n (not a multiple of 4) and n = 2
    for n in (2, 5, 131):
        A = np.random.RandomState(n).rand(n, n).astype(np.float32)
        ref, _, _ = O.normalize_adj_tensor(A)
        assert rel(E.normalize_adj_tensor(dev(t, A)).cpu().numpy(), ref) < 1e-6
    # zero embedding rows: F.normalize clamps the norm at 1e-12 -> decode gives 0, not NaN
    Z = np.zeros((6, 4), np.float32); Z[0] = 1
    got = E.dot_product_decode(dev(t, Z)).cpu().numpy()
    assert np.all(np.isfinite(got)) and np.all(got == 0)
    # entropy clamp boundaries
    P = np.array([[0, 1e-4], [1 - 1e-4, 1.5]], np.float32)
    assert abs(float(E.info_entropy(dev(t, P))) - float(O.info_entropy(P))) < 1e-7


def test_gcn_forward_matches_oracle(pkg, torch_):
    from mc_gra_amd import engine as E
    t = torch_
    z = H.load_case("s80_hsic_l3")
    w = H.weights_from(z)
    T0 = (z["features"] @ w.W[0]).astype(np.float32)
    _, Hs, _ = O.gcn_chain(T0, z["adj"], w.f32(), 3)
    _, logp = O.victim_head(Hs[-1], w.f32())
    out, emb = E.gcn_forward(dev(t, z["features"]), dev(t, z["adj"]), [dev(t, x) for x in w.W],
                             [dev(t, x) for x in w.b], dev(t, w.Wlin), dev(t, w.blin), emb_nlayer=2)
    assert rel(out.cpu().numpy(), logp) < 2e-5
    assert rel(emb.cpu().numpy(), Hs[1]) < 2e-5
    assert rel(out.cpu().numpy(), z["Y_A"]) < 2e-5          # the reference's own Y_A
    assert rel(emb.cpu().numpy(), z["H_A2"]) < 2e-5


# ------------------------------------------------------------------ attack engine
# entries (of 44 850) the separate-kernel path moves the other way on a noise-level gradient (measured: one at step 0, a second at step 1,
# by 2 lr: the fixture holds entries whose reference gradient is 1e-16 ... 1e-9 of the gradient's largest magnitude, and
# Adam's first step is lr * sign(g))
STRICT_OUTLIERS = {"s300_hsic_eps": 2}
ENGINE_CASES = [c for c in H.attack_cases() if str(H.load_case(c)["measure"]) in ("HSIC", "MSELoss", "KL", "DP", "CKA", "KDE")]


@pytest.mark.parametrize("name", ENGINE_CASES)
def test_step_gradients_match_reference(pkg, torch_, name):
    """Teacher-forced: every step starts from the reference's adj_changes; the packed
    gradient mirrored by the engine (G_sym) must equal the reference's autograd gradient."""
    z = H.load_case(name)
    eng = H.engine_from(pkg, z)
    orc = H.oracle_from(z)
    free_run = float(z["num_edges"]) < 1e9
    for t in range(int(z["epochs"])):
        if t > 0 and not free_run:
            a = np.clip(z["steps_a"][t - 1], 0, 1)
            eng.set_adj_changes(a)
            orc.set_adj_changes(a)
        nz = H.noise_of(z, t)
        sc = eng.step(want_scalars=True, noise=None if nz is None else dev(torch_, nz))
        orc.step(nz)
        g = O.pack_tril(eng.buffer("G_sym").cpu().numpy())
        g_ref = z["steps_g"][t]
        scale = np.abs(g_ref).max()
        tol = 6e-4 if "cka" in name else 3e-4
        assert np.abs(g - g_ref).max() <= tol * scale, (name, t, np.abs(g - g_ref).max(), scale)
        if t == 0 and "step0_g64" in z:      # measure KDE: also against the reference's OWN code run in float64 (make_golden.py)
            assert np.abs(g - z["step0_g64"]).max() <= 3e-5 * np.abs(z["step0_g64"]).max(), (name, np.abs(g - z["step0_g64"]).max())
        # intermediates against the oracle
        last = orc.last
        # the bisection case cannot be teacher-forced (state is not recoverable from the hook), so engine
        # and oracle free-run from their own Adam-amplified states: compare loosely there
        k = 500.0 if free_run else 1.0
        assert rel(eng.buffer("adj_norm").cpu().numpy(), last["adj_norm"]) < 2e-6 * k
        assert rel(eng.buffer("A1").cpu().numpy(), last["A1"]) < 2e-5 * k
        assert rel(eng.buffer("em").cpu().numpy(), last["em"]) < 2e-5 * k
        assert abs(sc["loss"] - last["loss"]) <= 2e-4 * abs(last["loss"]) + 1e-5, (sc, last["loss"], last["terms"])
    eng.close()


@pytest.mark.parametrize("name", ENGINE_CASES)
def test_free_run_final_matches_reference(pkg, torch_, name):
    """Free-running loop + post-loop ensemble: final modified_adj and its AUC."""
    z = H.load_case(name)
    eng = H.engine_from(pkg, z)
    # entries whose first gradient has the other SIGN in the reference's fp32 run than in the reference's own float64 run
    # (s200_kde_init: 3 of 19 900, |g| = 5e-7 of the largest): Adam's first step is lr * sign(g), the engine follows the exact sign
    ref_flip = (np.sign(z["steps_g"][0]) != np.sign(z["step0_g64"])) if "step0_g64" in z else False
    for t in range(int(z["epochs"])):
        nz = H.noise_of(z, t)
        eng.step(noise=None if nz is None else dev(torch_, nz))
        a = eng.get_adj_changes().cpu().numpy()
        off = (np.abs(a - np.clip(z["steps_a"][t], 0, 1)) >= 0.05 * float(z["lr"]) + 1e-6) & ~ref_flip
        # (n >= 256: a few of the 10^4..10^5 entries carry gradients at rounding-noise level, whose sign decides a whole
        # +-lr Adam move in the first steps -- DESIGN.md section 5; the small cases match entry for entry)
        # (... and where the reference's own first step went the other way on some entries, their 2 lr reach their neighbours'
        # gradients from the second step on: s200_kde_init, 3 more entries at step 1, 10 at step 2 -- in the float64 oracle alike)
        strict = z["adj"].shape[0] < 256 and not np.any(ref_flip)
        assert off.mean() <= (0.0 if strict else 1e-3) or float(z["num_edges"]) < 1e9, (t, off.sum())
        # ... and an entry that does differ differs by Adam moves of opposite sign, not by anything larger
        if float(z["num_edges"]) >= 1e9:
            worst = float(np.abs(a - np.clip(z["steps_a"][t], 0, 1)).max())
            assert worst <= 2.0 * (t + 1) * float(z["lr"]) + 1e-6, (t, worst)
    lab = z["labels"]
    label_adj = (lab[:, None] == lab[None, :]).astype(np.float32)
    final = eng.finalize(0, z["H_A2"], z["Y_A"], label_adj).cpu().numpy()
    assert np.abs(final - z["final"]).max() < 1e-3 * max(1.0, np.abs(z["final"]).max())
    auc = O.metric_pool(z["adj"], final, z["idx_attack"])
    assert abs(auc - float(z["auc"])) < 1e-4
    eng.close()


@pytest.mark.parametrize("name", [c for c in ENGINE_CASES if H.load_case(c)["adj"].shape[0] >= 256])
def test_free_run_strict_on_the_separate_kernels(pkg, torch_, name, monkeypatch):
    """The same fixtures through the general path with its tail as separate kernels (MCGRA_NO_FUSED_LR=1,
    MCGRA_NO_FUSED_TAIL=1): the per-step state is held to the small cases' bar, entry for entry, so a regression of the
    general path is caught exactly even where the default kernels are allowed their Adam-noise outliers."""
    monkeypatch.setenv("MCGRA_NO_FUSED_LR", "1")
    monkeypatch.setenv("MCGRA_NO_FUSED_TAIL", "1")
    z = H.load_case(name)
    eng = H.engine_from(pkg, z)
    lr = float(z["lr"])
    for t in range(int(z["epochs"])):
        nz = H.noise_of(z, t)
        eng.step(noise=None if nz is None else dev(torch_, nz))
        a = eng.get_adj_changes().cpu().numpy()
        err = np.abs(a - np.clip(z["steps_a"][t], 0, 1))
        assert int((err >= 0.05 * lr + 1e-6).sum()) <= STRICT_OUTLIERS.get(name, 0), (name, t, int((err >= 0.05 * lr + 1e-6).sum()), err.max())
        assert err.max() <= 2.0 * (t + 1) * lr + 1e-6
    assert eng.fused_steps() == 0
    eng.close()


# dot_product_decode2 branch (topology_attack.py:421-467) -> decode_mode, with the (dataset, useH_A, useY_A, useY)
# the reference was run with for each golden in ops.npz
DD2_GOLDENS = [("cora", (1, 1, 1), 0), ("AIDS", (1, 0, 0), 0), ("citeseer", (1, 1, 1), 1), ("brazil", (1, 1, 1), 2),
               ("polblogs", (1, 1, 1), 3), ("polblogs", (1, 0, 1), 3), ("usair", (0, 0, 1), 5), ("usair", (1, 1, 0), 4),
               ("usair", (1, 0, 1), 6), ("usair", (1, 1, 1), 3)]


@pytest.mark.parametrize("ds,use,mode", DD2_GOLDENS)
def test_dot_product_decode2_reference_goldens(pkg, torch_, ds, use, mode):
    """All ten dot_product_decode2 goldens of ops.npz (the reference's own outputs) through the HIP kernels the
    post-loop ensemble uses, and through the host layer's dataset -> mode mapping."""
    import argparse
    from mc_gra_amd import engine as E
    from mc_gra_amd.topology_attack import _decode_mode
    args = argparse.Namespace(dataset=ds, useH_A=bool(use[0]), useY_A=bool(use[1]), useY=bool(use[2]))
    assert _decode_mode(args) == mode
    got = E.dot_product_decode2(dev(torch_, OPS["dd2_Z"]), mode).cpu().numpy()
    ref = OPS[f"dd2_{ds}_{use[0]}{use[1]}{use[2]}"]
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()


FIN_MODES = {0: ("cora", (1, 1, 1)), 1: ("citeseer", (1, 1, 1)), 2: ("brazil", (1, 1, 1)), 3: ("polblogs", (1, 1, 1)),
             4: ("usair", (1, 1, 0)), 5: ("usair", (0, 0, 1)), 6: ("usair", (1, 0, 1))}


@pytest.mark.parametrize("mode", sorted(FIN_MODES))
@pytest.mark.parametrize("case", ["s48_mse", "s80_hsic_l3"])
def test_finalize_every_decode_mode_matches_oracle(pkg, torch_, case, mode):
    """mcgra_attack_finalize (post-loop ensemble, topology_attack.py:300-324) with every dot_product_decode2 branch:
    HIP against the oracle (itself pinned to the ten reference goldens by tests/test_oracle_golden.py), priors
    switched as the branch's own flags say."""
    ds, use = FIN_MODES[mode]
    z = H.load_case(case)
    eng = H.engine_from(pkg, z)
    orc = H.oracle_from(z)
    for t in range(2):
        eng.step(); orc.step()
        orc.set_adj_changes(eng.get_adj_changes().cpu().numpy())      # same state going into the ensemble
    lab = z["labels"]
    label_adj = (lab[:, None] == lab[None, :]).astype(np.float32)
    got = eng.finalize(mode, z["H_A2"] if use[0] else None, z["Y_A"] if use[1] else None,
                       label_adj if use[2] else None).cpu().numpy()
    ref = orc.finalize(ds, bool(use[0]), bool(use[1]), bool(use[2]), label_adj, z["H_A2"], z["Y_A"])
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (mode, np.abs(got - ref).max())
    # finalize replaced the learnable adjacency by the decoded one: whatever the last step left behind for the next
    # forward is stale, a following step must recompute it (and must not silently use it)
    sc = eng.step(want_scalars=True)
    assert np.isfinite(sc["loss"])
    eng.close()


def test_projection_bisection_matches_oracle(pkg, torch_):
    z = H.load_case("s80_mse_proj")
    eng = H.engine_from(pkg, z)
    orc = H.oracle_from(z)
    for t in range(int(z["epochs"])):
        eng.step(); orc.step()
        a = eng.get_adj_changes().cpu().numpy()
        assert np.abs(a - O.pack_tril(orc.M)).max() < 2e-3
        assert a.min() >= 0 and a.max() <= 1
        assert a.sum() <= float(z["num_edges"]) * (1 + 1e-3) + 1e-2
    eng.close()


def test_engine_invariants_and_determinism(pkg, torch_):
    """Integer-like structure must be bit exact: zero diagonal, exact symmetry of the
    mirrored state, clamp range; two runs give identical bits (deterministic reductions)."""
    z = H.load_case("s200_hsic")
    outs = []
    for rep in range(2):
        eng = H.engine_from(pkg, z)
        for _ in range(3):
            eng.step()
        M = eng.buffer("M").cpu().numpy()
        outs.append(M)
        assert np.array_equal(M, M.T)
        assert np.all(np.diag(M) == 0)
        assert M.min() >= 0 and M.max() <= 1
        m, v = eng.buffer("adam_m").cpu().numpy(), eng.buffer("adam_v").cpu().numpy()
        assert np.array_equal(m, m.T) and np.array_equal(v, v.T)
        eng.close()
    assert np.array_equal(outs[0], outs[1])


def test_unsupported_arguments_fail_loudly(pkg, torch_):
    z = H.load_case("s48_mse")
    with pytest.raises(ValueError):
        H.engine_from(pkg, z, measure="Wasserstein")          # (topology_attack.py:194-208 knows six measures; all six are built)
    from mc_gra_amd._lib import McgraNotSupported
    wide = H.synthetic_case(300, 11, (16, 48), 4, seed=1, measure="KDE")
    wide["emb_nlayer"] = np.array(2)
    with pytest.raises(McgraNotSupported, match="KDE"):      # the c x c joint of utils.MutualInformation: widths <= 32
        H.engine_from(pkg, wide)


# ------------------------------------------------------------------ class surface + Cora
def _args(measure, dataset="cora"):
    import argparse
    return argparse.Namespace(max_eval=100, lr=0, dataset=dataset, eps=0, measure=measure, useH_A=True,
                              useY_A=True, useY=True, w1=0, w2=0, w6=0, w7=0, w8=0, w9=0, w10=0)


def test_pgdattack_class_small(pkg, torch_):
    z = H.load_case("s48_hsic")
    w = H.weights_from(z)
    victim, emb = H.FakeGCN(w), H.FakeGCN(w)
    model = pkg.PGDAttack(model=victim, embedding=emb, H_A=torch_.tensor(z["H_A2"]), Y_A=torch_.tensor(z["Y_A"]),
                          nnodes=48, loss_type="CE", device="cuda:0")
    lab = z["labels"]
    ret = model.attack(_args("HSIC"), None, float(z["lr"]), 0, float(z["weight_sup"]), tuple(z["weight_param"]),
                       torch_.tensor(z["feature_adj"]), 0, 0, 0, None, None, np.arange(8), torch_.tensor(z["adj"]),
                       torch_.tensor(z["features"]), torch_.zeros(48, 48), torch_.tensor(lab), z["idx_attack"],
                       float(z["num_edges"]), 0, epochs=int(z["epochs"]),
                       label_adj=(lab[:, None] == lab[None, :]).astype(np.float32))
    assert ret == (0, 0, 0, 0)
    final = model.modified_adj.cpu().numpy()
    assert np.abs(final - z["final"]).max() < 1e-3
    assert abs(O.metric_pool(z["adj"], final, z["idx_attack"]) - float(z["auc"])) < 1e-4
    assert len(model.history["acc_test"]) == int(z["epochs"])


def test_pgdattack_class_nonzero_ori_adj(pkg, torch_):
    """A non-zero ori_adj through the class (topology_attack.py:164-165, :185, :188, :302; main.py only ever passes zeros):
    modified_adj = clamp(adj_changes + ori_adj) with its gradient gate, the embedding on modified_adj - ori_adj, + ori_adj
    in modified_adj1 and in the returned adjacency.  Reference fixture with a random symmetric 0/1 ori_adj."""
    z = H.load_case("s48_hsic_ori")
    w = H.weights_from(z)
    victim, emb = H.FakeGCN(w), H.FakeGCN(w)
    model = pkg.PGDAttack(model=victim, embedding=emb, H_A=torch_.tensor(z["H_A2"]), Y_A=torch_.tensor(z["Y_A"]),
                          nnodes=48, loss_type="CE", device="cuda:0")
    model.adj_changes = H.a0_of(z)
    lab = z["labels"]
    model.attack(_args("HSIC"), None, float(z["lr"]), 0, float(z["weight_sup"]), tuple(z["weight_param"]),
                 torch_.tensor(z["feature_adj"]), 0, 0, 0, None, None, np.arange(8), torch_.tensor(z["adj"]),
                 torch_.tensor(z["features"]), torch_.tensor(z["ori_adj"]), torch_.tensor(lab), z["idx_attack"],
                 float(z["num_edges"]), 0, epochs=int(z["epochs"]), label_adj=(lab[:, None] == lab[None, :]).astype(np.float32))
    final = model.modified_adj.cpu().numpy()
    assert np.abs(final - z["final"]).max() < 1e-3 * max(1.0, np.abs(z["final"]).max())
    assert abs(O.metric_pool(z["adj"], final, z["idx_attack"]) - float(z["auc"])) < 1e-4


def test_pgdattack_class_loss_type_cw(pkg, torch_):
    """loss_type='CW': the reference back-propagates the margin loss but calls optimizer.step() only for 'CE'
    (topology_attack.py:277-280), so its run returns the post-loop ensemble of the untouched adjacency; the fixture is
    that run (tests/golden/make_golden.py --only cw)."""
    z = dict(np.load(os.path.join(H.GOLDEN, "cw_s48_mse_cw.npz"), allow_pickle=False))
    w = H.weights_from(z)
    victim, emb = H.FakeGCN(w), H.FakeGCN(w)
    model = pkg.PGDAttack(model=victim, embedding=emb, H_A=torch_.tensor(z["H_A2"]), Y_A=torch_.tensor(z["Y_A"]),
                          nnodes=48, loss_type="CW", device="cuda:0")
    lab = z["labels"]
    model.attack(_args("MSELoss"), None, float(z["lr"]), 0, float(z["weight_sup"]), tuple(z["weight_param"]),
                 torch_.tensor(z["feature_adj"]), 0, 0, 0, None, None, np.arange(8), torch_.tensor(z["adj"]),
                 torch_.tensor(z["features"]), torch_.zeros(48, 48), torch_.tensor(lab), z["idx_attack"],
                 float(z["num_edges"]), 0, epochs=int(z["epochs"]),
                 label_adj=(lab[:, None] == lab[None, :]).astype(np.float32))
    final = model.modified_adj.cpu().numpy()
    assert np.abs(final - z["final"]).max() < 1e-3
    assert abs(O.metric_pool(z["adj"], final, z["idx_attack"]) - float(z["auc"])) < 1e-4
    assert len(model.history["acc_test"]) == int(z["epochs"]) and model.engine.path_stats() == {"lowrank_steps": 0, "general_steps": 0}


@pytest.mark.parametrize("name,fake", [("s48_gat_hsic_init", "FakeGAT"), ("s48_sage_kl", "FakeSAGE")])
def test_pgdattack_class_other_victims(pkg, torch_, name, fake):
    """main.py --arch gat / sage: PGDAttack reads the victim family off the model object."""
    z = H.load_case(name)
    w = H.weights_from(z)
    victim, emb = getattr(H, fake)(w), getattr(H, fake)(w)
    n = z["adj"].shape[0]
    model = pkg.PGDAttack(model=victim, embedding=emb, H_A=torch_.tensor(z["H_A2"]), Y_A=torch_.tensor(z["Y_A"]),
                          nnodes=n, loss_type="CE", device="cuda:0")
    if H.a0_of(z) is not None:
        model.adj_changes = H.a0_of(z)
    lab = z["labels"]
    model.attack(_args(str(z["measure"])), None, float(z["lr"]), 0, float(z["weight_sup"]), tuple(z["weight_param"]),
                 torch_.tensor(z["feature_adj"]), 0, 0, 0, None, None, np.arange(8), torch_.tensor(z["adj"]),
                 torch_.tensor(z["features"]), torch_.zeros(n, n), torch_.tensor(lab), z["idx_attack"],
                 float(z["num_edges"]), 0, epochs=int(z["epochs"]),
                 label_adj=(lab[:, None] == lab[None, :]).astype(np.float32))
    final = model.modified_adj.cpu().numpy()
    assert np.abs(final - z["final"]).max() < 1e-3 * max(1.0, np.abs(z["final"]).max())
    assert abs(O.metric_pool(z["adj"], final, z["idx_attack"]) - float(z["auc"])) < 1e-4


def test_pgdattack_class_refuses_kde_on_a_wide_embedding_by_name(pkg, torch_):
    """measure = KDE with a GAT victim (utils.MutualInformation(num_bins = H_A1.shape[1]), topology_attack.py:244-249; the fixture's
    GAT is 3 heads x 16 = 48 wide, main.py's 5 x 16 = 80): the
    c x c joint pdf of this path is built for widths <= 32 -- PGDAttack.attack says so, naming the width and the reference lines,
    before an engine exists."""
    z = H.load_case("s48_gat_hsic_init")
    w = H.weights_from(z)
    victim, emb = H.FakeGAT(w), H.FakeGAT(w)
    n = z["adj"].shape[0]
    model = pkg.PGDAttack(model=victim, embedding=emb, H_A=torch_.tensor(z["H_A2"]), Y_A=torch_.tensor(z["Y_A"]),
                          nnodes=n, loss_type="CE", device="cuda:0")
    lab = z["labels"]
    with pytest.raises(NotImplementedError, match="width 48"):
        model.attack(_args("KDE"), None, float(z["lr"]), 0, float(z["weight_sup"]), tuple(z["weight_param"]),
                     torch_.tensor(z["feature_adj"]), 0, 0, 0, None, None, np.arange(8), torch_.tensor(z["adj"]),
                     torch_.tensor(z["features"]), torch_.zeros(n, n), torch_.tensor(lab), z["idx_attack"],
                     float(z["num_edges"]), 0, epochs=1, label_adj=(lab[:, None] == lab[None, :]).astype(np.float32))
    assert model.engine is None


def _run_cora(pkg, t, name, epochs=None):
    z = H.load_cora(name)
    if epochs is not None:
        z["epochs"] = np.array(epochs)
    w = O.GCNWeights([z["W0"], z["W1"]], [z["b0"], z["b1"]], z["Wlin"], z["blin"])
    victim, emb = H.FakeGCN(w), H.FakeGCN(w)
    X, adj, lab = z["features"], z["adj"], z["labels"]
    fadj = H.cora_feature_adj(X)
    from mc_gra_amd import engine as E
    Wd = [dev(t, x) for x in w.W]; bd = [dev(t, x) for x in w.b]
    Y_A, H_A2 = E.gcn_forward(dev(t, X), dev(t, adj), Wd, bd, dev(t, w.Wlin), dev(t, w.blin), emb_nlayer=2)
    model = pkg.PGDAttack(model=victim, embedding=emb, H_A=H_A2, Y_A=Y_A, nnodes=adj.shape[0], loss_type="CE",
                          device="cuda:0")
    if H.a0_of(z) is not None:
        model.adj_changes = H.a0_of(z)          # public attribute of the reference class (topology_attack.py:77)
    model.attack(_args(str(z["measure"])), None, float(z["lr"]), 0, float(z["weight_sup"]), tuple(z["weight_param"]),
                 fadj, 0, 0, 0, None, None, z["idx_test"], adj, X, np.zeros_like(adj), lab, z["idx_attack"],
                 float(z["num_edges"]), 0, epochs=int(z["epochs"]),
                 label_adj=(lab[:, None] == lab[None, :]).astype(np.float32))
    final = model.modified_adj.cpu().numpy()
    return z, final, O.metric_pool(adj, final, z["idx_attack"])


@pytest.mark.parametrize("name", ["cora_mse_short", "cora_hsic", "cora_hsic_sparse"])
def test_cora_auc_matches_reference(pkg, torch_, name):
    """BASELINE configs[0]/[1]: Cora, 2-layer GCN trained by the reference, priors H_A+Y_A+Y.
    north_star bar: recovered-adjacency AUC within 1e-4 of the reference CPU path, on horizons where the
    reference itself is reproducible to that level."""
    z, final, auc = _run_cora(pkg, torch_, name)
    assert abs(auc - float(z["auc"])) <= 1e-4, (auc, float(z["auc"]))
    sp = z["sample_pos"]
    # entries whose gradient sits at the fp32 noise level move by +-lr on noise (Adam); they are a small
    # fraction and do not move the AUC
    assert np.mean(np.abs(final[sp[:, 0], sp[:, 1]] - z["final_sample"]) > 2e-2) < 0.01
    assert abs(final.astype(np.float64).sum() - float(z["final_sum"])) <= 1e-4 * abs(float(z["final_sum"]))


CKPT = np.load(os.path.join(H.GOLDEN, "cora_mse_checkpoints.npz"))


def _engine_from_bits(pkg, z):
    """AttackEngine from a fixture that carries its graph as bits (tests.helpers.load_cora layout) and its own start."""
    w = H.weights_from(z)
    n = z["adj"].shape[0]
    dims = [w.W[0].shape[0]] + [x.shape[1] for x in w.W]
    eng = pkg.AttackEngine(n, dims, w.Wlin.shape[0], 2, str(z["measure"]), float(z["weight_sup"]),
                           tuple(float(x) for x in z["weight_param"]), float(z["lr"]), float(z["num_edges"]),
                           len(z["idx_attack"]))
    eng.set_model(w.W, w.b, w.Wlin, w.blin)
    eng.set_graph(z["features"], z["adj"], None, H.cora_feature_adj(z["features"]), z["labels"], z["idx_attack"])
    eng.set_adj_changes(H.init_adj_changes(n, z["a0_seed"], z["a0_scale"]))
    return eng


def test_mid_size_reference_run_on_the_fused_path(pkg, torch_):
    """A REFERENCE run (not the oracle) at a size where the default path is the fused low-rank step with the fp16-split
    product (n = 1200 >= 1024), from the bench's start rule, where the N x N terms carry the gradient (from the dense
    start of the other fixtures they are rounded away at n >= 300: VERDICT round 2, item 1c).  Step 0 starts from the
    reference's own state; there the reference's own fp32 gradient is 4.6e-4 of the gradient's largest magnitude away from
    a float64 evaluation of the same algorithm (tests/golden/make_truth64.py mid), so -- as at N = 10 000 -- the engine is
    held to 3e-4 of the EXACT gradient and to the reference within the reference's own distance from it plus 3e-4; later
    steps free-run."""
    z = H.load_cora("mid_s1200_hsic_sparse")
    z64 = np.load(os.path.join(H.GOLDEN, "mid_s1200_hsic_sparse_fp64.npz"))
    assert np.array_equal(z64["packed_pos"], z["packed_pos"])
    n = z["adj"].shape[0]
    eng = _engine_from_bits(pkg, z)
    assert eng.product_mode() == 3
    assert np.abs(eng.buffer("YA").cpu().numpy() - z["Y_A"]).max() <= 5e-5
    pi, pj = H.tril_pos(z["packed_pos"])
    ti, tj = torch_.as_tensor(pi, device="cuda:0"), torch_.as_tensor(pj, device="cuda:0")
    lr = float(z["lr"])
    for t in range(int(z["epochs"])):
        eng.step(); eng.monitor()
        Gs = eng.buffer("G_sym")
        g = Gs[ti, tj].cpu().numpy()
        gmax = float(z["step_g_absmax"][t])
        err = np.abs(g - z["step_g"][t]).max() / gmax
        ref_flips = 0.0
        if t == 0:
            g64 = z64["step0_g64"]
            err_true, ref_true = np.abs(g - g64).max() / gmax, np.abs(z["step_g"][0] - g64).max() / gmax
            ref_flips = float((np.sign(z["step_g"][0]) != np.sign(g64)).mean())
            assert err_true <= 3e-4 and err <= ref_true + 3e-4, (err_true, err, ref_true)
        else:
            assert err <= 2e-3, (t, err)
        assert abs(float(Gs.abs().max()) - gmax) <= 1e-3 * gmax
        a = eng.buffer("M")[ti, tj].cpu().numpy()
        moved = np.abs(a - np.clip(z["step_a"][t], 0, 1)) > 0.05 * lr
        assert moved.mean() <= (ref_flips + 0.001 if t == 0 else 0.01), (t, moved.mean(), ref_flips)
    assert eng.fused_steps() == int(z["epochs"]) and eng.path_stats()["general_steps"] == 0
    lab = z["labels"]
    final = eng.finalize(0, z["H_A2"], z["Y_A"], (lab[:, None] == lab[None, :]).astype(np.float32)).cpu().numpy()
    assert abs(O.metric_pool(z["adj"], final, z["idx_attack"]) - float(z["auc"])) <= 1e-4
    sp = z["sample_pos"]
    assert np.mean(np.abs(final[sp[:, 0], sp[:, 1]] - z["final_sample"]) > 1e-3 * max(1.0, np.abs(z["final_sample"]).max())) < 0.01


@pytest.mark.parametrize("path", ["fused", "general"])
@pytest.mark.parametrize("epochs", [10, 40])
def test_cora_mse_checkpoints(pkg, torch_, monkeypatch, epochs, path):
    """README Cora run at intermediate horizons against the reference's own AUC there
    (tests/golden/make_adam_noise.py).  The reference, the fp32 oracle and the fp64 oracle -- three evaluations of ONE
    algorithm on identical inputs -- agree to 1.6e-7 / 6e-6 / 1.2e-4 / 1.4e-3 at 10 / 20 / 40 / 100 epochs
    (profiles/r02_adam_noise_experiment.json): Adam turns fp32 rounding noise on near-zero gradients into +-lr moves,
    so the bar of north_star (1e-4) is meaningful up to ~20 epochs; beyond, the engine is held to the measured spread
    of those three at that horizon: x 2 for the general step (fp32 arithmetic throughout, like the three), x 4 for the
    fused MSELoss step, whose rank-k terms -- the chains' backward, i.e. the gradient of most entries -- are 3-product
    fp16 splits on 22-bit operands (2^-22 against fp32's 2^-24 per operand: four times the noise floor under the same
    amplification).  Measured at 40 epochs, fused: 3.0e-4 and 3.6e-4 from the reference in two builds that differ only
    in the order of the decode's partial sums -- a draw of that noise, like the spread itself (1.2e-4)."""
    i = list(CKPT["epochs"]).index(epochs)
    ref, o32, o64 = float(CKPT["auc_reference"][i]), float(CKPT["auc_oracle_fp32"][i]), float(CKPT["auc_oracle_fp64"][i])
    spread = max(abs(ref - o32), abs(ref - o64), abs(o32 - o64))
    if path == "general":
        monkeypatch.setenv("MCGRA_NO_FUSED_LR", "1")
    auc = _run_cora(pkg, torch_, "cora_mse_readme", epochs=epochs)[2]
    bar = max(1e-4, 4 * spread) if path == "fused" else max(1e-4, 2 * spread)
    assert abs(auc - ref) <= bar, (epochs, path, auc, ref, spread)


def test_cora_readme_100_epochs(pkg, torch_):
    """README headline run (MSELoss, 100 epochs).  Bar: twice the measured spread of the reference, the fp32 oracle and
    the fp64 oracle at 100 epochs (1.4e-3, see test_cora_mse_checkpoints): no fp32 implementation is reproducible more
    tightly on that horizon, the float64 evaluation of the reference's own algorithm included."""
    z, final, auc = _run_cora(pkg, torch_, "cora_mse_readme")
    i = list(CKPT["epochs"]).index(100)
    ref, o32, o64 = float(CKPT["auc_reference"][i]), float(CKPT["auc_oracle_fp32"][i]), float(CKPT["auc_oracle_fp64"][i])
    assert ref == float(z["auc"])
    spread = max(abs(ref - o32), abs(ref - o64), abs(o32 - o64))
    assert abs(auc - ref) <= 2 * spread, (auc, ref, spread)


def test_main_entry_end_to_end(pkg, torch_, tmp_path, monkeypatch):
    """main.py flow (dataset npz -> victim training -> priors -> attack -> AUC) on Cora rebuilt from the fixture."""
    import scipy.sparse as sp
    z = H.load_cora("cora_mse_short")
    A = sp.csr_matrix(np.triu(z["adj"], 1)); X = sp.csr_matrix(z["features"])
    root = tmp_path / "dataset"; root.mkdir()
    np.savez(root / "cora.npz", adj_data=A.data, adj_indices=A.indices, adj_indptr=A.indptr, adj_shape=A.shape,
             attr_data=X.data, attr_indices=X.indices, attr_indptr=X.indptr, attr_shape=X.shape, labels=z["labels"])
    monkeypatch.chdir(tmp_path)
    from mc_gra_amd import main as M
    args = M.build_parser().parse_args(["--dataset", "cora", "--dataset_root", str(root), "--w1", "0.01", "--w6", "10",
                                        "--w7", "10", "--w9", "10", "--w10", "1000", "--lr", "-2", "--useH_A",
                                        "--useY_A", "--useY", "--measure", "MSELoss", "--epochs", "20"])
    res = M.run(args)
    # the victim is trained here (not the reference's weights), so only the level is checked: the reference gets
    # 0.898 after 20 epochs with its own victim
    assert 0.85 < res["auc_all"] < 0.95, res
    assert os.path.exists(tmp_path / "results" / "result.txt")


def _dataset_npz_from_fixture(z, path):
    import scipy.sparse as sp
    A = sp.csr_matrix(np.triu(z["adj"], 1)); X = sp.csr_matrix(z["features"])
    np.savez(path, adj_data=A.data, adj_indices=A.indices, adj_indptr=A.indptr, adj_shape=A.shape,
             attr_data=X.data, attr_indices=X.indices, attr_indptr=X.indptr, attr_shape=X.shape, labels=z["labels"])


def test_main_entry_gat_citeseer_end_to_end(pkg, torch_, tmp_path, monkeypatch):
    """BASELINE configs[2] through the entry point: `main.py --arch gat --dataset citeseer --useH_A --useY` with the
    README's citeseer weights (main.py:213-231: GAT.fit, embedding_gat sharing the attention layers, H_A from the
    train-mode embedding as the reference computes it), Citeseer rebuilt from the fixture.  The victim is trained here,
    so the AUC is checked for level against the reference run of the fixture (0.9157 with its own victim)."""
    z = H.load_cora("citeseer_gat_kl")
    root = tmp_path / "dataset"; root.mkdir()
    _dataset_npz_from_fixture(z, root / "citeseer.npz")
    monkeypatch.chdir(tmp_path)
    from mc_gra_amd import main as M
    args = M.build_parser().parse_args(["--dataset", "citeseer", "--arch", "gat", "--dataset_root", str(root), "--w1", "0.001",
                                        "--w2", "10000", "--w6", "0.0001", "--w7", "100", "--w9", "100", "--lr", "-1",
                                        "--useH_A", "--useY", "--measure", "KL", "--epochs", "3"])
    args.gat_train_iters = 30
    res = M.run(args)
    assert abs(res["auc_all"] - float(z["auc"])) < 0.03, res
    assert os.path.exists(tmp_path / "results" / "result.txt")


def test_main_entry_sage_cora_end_to_end(pkg, torch_, tmp_path, monkeypatch):
    """`main.py --arch sage` (main.py:193-210: graphsage.fit, embedding_graphsage with a deep copy of its layers) on Cora
    rebuilt from the fixture, HSIC; level check (the reference's GCN-victim run of the same configuration: 0.8955)."""
    z = H.load_cora("cora_hsic")
    root = tmp_path / "dataset"; root.mkdir()
    _dataset_npz_from_fixture(z, root / "cora.npz")
    monkeypatch.chdir(tmp_path)
    from mc_gra_amd import main as M
    args = M.build_parser().parse_args(["--dataset", "cora", "--arch", "sage", "--dataset_root", str(root), "--w1", "0.01",
                                        "--w2", "0.01", "--w6", "10", "--w7", "10", "--w9", "10", "--w10", "1000", "--lr", "-2",
                                        "--useH_A", "--useY_A", "--useY", "--measure", "HSIC", "--epochs", "8"])
    res = M.run(args)
    assert 0.82 < res["auc_all"] < 0.95, res


NXN_ONLY = (0.01, 0.01, 0, 0, 0, 10, 10, 0, 0, 0)        # c1, c2, c6, c7 only: the N x N terms carry the whole gradient


# ---- row-block sharded step (scope row (e)) -----------------------------------------------------------------
def _shard_engines(pkg, z, world, joint=False, **kw):
    from mc_gra_amd.sharded import RowBlockPlan, HipShardBackend, lockstep_backends
    n = z["adj"].shape[0]
    plans = [RowBlockPlan(n, world, r) for r in range(world)]
    if joint:      # arenas as rows of one tensor: run_lockstep's collectives become single strided copies
        return plans, lockstep_backends([H.engine_from(pkg, z, plan=p, **kw) for p in plans], plans)
    return plans, [HipShardBackend(H.engine_from(pkg, z, plan=p, **kw), p) for p in plans]


def _gather_rows(bks):
    import torch
    return torch.cat([b.eng.get_rows() for b in bks if b.plan.has_rows], 0)


@pytest.mark.parametrize("n,widths,world,wp", [(1100, (16, 16), 2, None), (1100, (16, 16), 3, None), (1283, (16, 8), 2, NXN_ONLY),
                                               (1030, (16, 16, 16), 4, None), (1100, (16, 16), 1, None),
                                               (600, (16, 16), 5, None)])
def test_sharded_ranks_match_monolithic_step(pkg, n, widths, world, wp, monkeypatch):
    """`world` row-block ranks on one GPU (each an engine that owns its rows of M and of the Adam moments and touches
    only those rows in every N x N pass), advanced in lockstep with the collectives of the protocol executed as copies
    between their arenas (sharded.run_lockstep): the union of their rows equals the monolithic fused step's adjacency
    to fp32 rounding, the loss terms agree on every rank, monitor + adopted forward included.  world 4 on n = 1030
    leaves the last ranks with 6 rows / none; world 1 is the degenerate protocol; n = 600 needs MCGRA_SPLIT_BF16."""
    import torch
    from mc_gra_amd import sharded as S
    kw = {} if wp is None else {"weight_param": wp}
    z = _synthetic_case(n, 11, widths, 4, seed=n, **kw)
    if n < 1024:
        monkeypatch.setenv("MCGRA_SPLIT_BF16", "3")
    mono = H.engine_from(pkg, z)
    plans, bks = _shard_engines(pkg, z, world, joint=world in (3, 4))
    lr = float(z["lr"])
    for t in range(3):
        a = mono.step(want_scalars=True); mono.monitor()
        sc = S.run_lockstep(bks, S.SHARD_STEP, want_scalars=True)
        S.run_lockstep(bks, S.SHARD_MONITOR)
        rows = _gather_rows(bks)
        M = mono.buffer("M")
        assert rows.shape == M.shape
        assert float(((rows - M).abs() > 0.05 * lr).float().mean()) < 2e-3, t      # Adam: +-lr on noise-level gradients
        assert float((rows - rows.T).abs().max()) == 0.0, "ranks must agree on mirrored entries bit for bit"
        for k in ("loss", "c1", "c2", "c6", "c7", "c9", "c10", "nll", "clamp_sum"):
            for b in sc:
                assert b[k] == pytest.approx(a[k], rel=3e-5, abs=1e-6 * max(1.0, abs(a["loss"]))), (t, k, b[k], a[k])
            assert all(b[k] == sc[0][k] for b in sc), "scalars are identical on every rank"
    assert all(b.eng.fused_steps() == 3 for b in bks) and mono.fused_steps() == 3
    # 2 <= world <= 4: every rank with rows computed its peers' row panels first and handed them to the all-to-all while its
    # own panels were still running; a larger world cuts only where a whole round of the chip ends
    # behind the peers' tiles (never at these sizes)
    cut = 3 if 2 <= world <= 4 else 0
    assert [b.eng.cut_product_steps() for b in bks] == [cut if p.has_rows else 0 for p in plans]


@pytest.mark.parametrize("world", [2, 3, 5])
def test_cut_product_for_the_all_to_all_matches_the_uncut_step(pkg, world, monkeypatch):
    """The product of a row-block rank cut for the all-to-all (peers' row panels first, rotated and wrapping around the own
    ones; the exchange point reached behind the first part, the own panels joined behind the exchange) against the same ranks
    with the product in one piece (MCGRA_A2A_OVERLAP=0): the same tiles by the same arithmetic -- only which of them fall
    into a ragged round's split-K differs -- so rows, mirrored bits and loss terms agree as the sharded and the monolithic
    step do.  world 5 (forced on; two ranks without rows) covers a rotation that wraps in the middle of the row panels."""
    import torch
    from mc_gra_amd import sharded as S
    z = _synthetic_case(1283, 11, (16, 16), 4, seed=1283)
    monkeypatch.setenv("MCGRA_A2A_OVERLAP", "1")
    plans, cut = _shard_engines(pkg, z, world, joint=world == 3)
    monkeypatch.setenv("MCGRA_A2A_OVERLAP", "0")
    _, whole = _shard_engines(pkg, z, world)
    lr = float(z["lr"])
    for t in range(3):
        sa = S.run_lockstep(cut, S.SHARD_STEP, want_scalars=True)
        sb = S.run_lockstep(whole, S.SHARD_STEP, want_scalars=True)
        S.run_lockstep(cut, S.SHARD_MONITOR); S.run_lockstep(whole, S.SHARD_MONITOR)
        ra, rb = _gather_rows(cut), _gather_rows(whole)
        assert float(((ra - rb).abs() > 0.05 * lr).float().mean()) < 2e-3, t
        assert float((ra - ra.T).abs().max()) == 0.0
        for k in ("loss", "c1", "c2", "c6", "c7", "c9", "c10", "nll", "clamp_sum"):
            assert sa[0][k] == pytest.approx(sb[0][k], rel=3e-5, abs=1e-6 * max(1.0, abs(sb[0]["loss"]))), (t, k)
            assert all(b[k] == sa[0][k] for b in sa)
    assert [b.eng.cut_product_steps() for b in cut] == [3 if p.has_rows else 0 for p in plans]
    assert all(b.eng.cut_product_steps() == 0 for b in whole)


def test_sharded_ranks_hand_masked_steps_to_the_general_path(pkg, monkeypatch):
    """A decode-masked step on row-block ranks: M and the Adam moments are all-gathered and every rank redoes the step
    with the general (Gram) path, replicated; the ranks' rows then equal the monolithic engine's bit for bit."""
    import torch
    from mc_gra_amd import sharded as S
    z = _synthetic_case(600, 11, (16, 16), 4, seed=9, weight_param=(0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 0))
    monkeypatch.setenv("MCGRA_SPLIT_BF16", "3")
    w = H.masked_weights(z)      # about half of em dies: the decode masks pairs
    mono = H.engine_from(pkg, z)
    mono.set_model(w.W, w.b, w.Wlin, w.blin, w.Ws)
    plans, bks = _shard_engines(pkg, z, 2)
    for b in bks:
        b.eng.set_model(w.W, w.b, w.Wlin, w.blin, w.Ws)
    for t in range(2):
        mono.step(); mono.monitor()
        S.run_lockstep(bks, S.SHARD_STEP)
        S.run_lockstep(bks, S.SHARD_MONITOR)
    assert mono.path_stats()["general_steps"] == 2 and all(b.eng.path_stats()["general_steps"] == 2 for b in bks)
    assert torch.equal(_gather_rows(bks), mono.buffer("M"))


def test_row_block_forward_forks_the_next_products_and_a_product_nobody_takes_is_dropped(pkg, monkeypatch):
    """A row-block rank's forward (the monitor call's, adopted by the next step) forks the NEXT step's pack and N x N x N product as
    soon as the degree vector is complete (attack_fused.hip: fork_p1_early).  (1) Against the same ranks with MCGRA_EARLY_P1=0
    -- product forked by the step, centred planes: the same adjacency to fp32 rounding, step by step.  (2) One product launch
    per step either way: the forward's launch is the step's, MCGRA_SHARD_MONITOR_LAST starts none.  (3) A forked product
    nobody takes -- the caller sets another adj_changes, calls the forward twice, finalizes -- is dropped: results equal a rank
    that never forked."""
    import torch
    from mc_gra_amd import sharded as S
    z = _synthetic_case(1100, 11, (16, 16), 4, seed=5)
    lr = float(z["lr"])
    _, early = _shard_engines(pkg, z, 2)
    monkeypatch.setenv("MCGRA_EARLY_P1", "0")
    _, late = _shard_engines(pkg, z, 2)
    monkeypatch.delenv("MCGRA_EARLY_P1")
    for b in early + late:
        b.eng.profile(True); b.eng.gemm_stats(reset=True)
    for t in range(3):
        S.run_lockstep(early, S.SHARD_STEP); S.run_lockstep(late, S.SHARD_STEP)
        re_, rl = _gather_rows(early), _gather_rows(late)
        assert float(((re_ - rl).abs() > 0.05 * lr).float().mean()) < 2e-3, t
        last = S.SHARD_MONITOR_LAST if t == 2 else S.SHARD_MONITOR
        S.run_lockstep(early, last); S.run_lockstep(late, last)
        torch.cuda.synchronize()
        # launches so far: t + 1 steps, + the one the forward forked for the next step (not behind the last)
        assert [b.eng.gemm_stats(reset=False)["launches"] for b in early] == [t + 1 + (t < 2)] * 2, t
        assert [b.eng.gemm_stats(reset=False)["launches"] for b in late] == [t + 1] * 2, t
    assert all(b.eng.fused_steps() == 3 for b in early + late)
    assert [b.eng.cut_product_steps() for b in early] == [b.eng.cut_product_steps() for b in late] == [3, 3]

    # (3) a forked product nobody takes
    _, (a,) = _shard_engines(pkg, z, 1)
    monkeypatch.setenv("MCGRA_EARLY_P1", "0")
    _, (b,) = _shard_engines(pkg, z, 1)
    monkeypatch.delenv("MCGRA_EARLY_P1")
    for e in (a, b):
        S.run_lockstep([e], S.SHARD_STEP); S.run_lockstep([e], S.SHARD_MONITOR)      # a: a product in flight
    a0 = a.eng.get_adj_changes() * 0.5
    for e in (a, b):
        e.eng.set_adj_changes(a0)                       # another M: the product of the old one is dropped
        S.run_lockstep([e], S.SHARD_MONITOR); S.run_lockstep([e], S.SHARD_MONITOR)      # two forwards in a row
        S.run_lockstep([e], S.SHARD_STEP)
    ra, rb = a.eng.get_rows(), b.eng.get_rows()
    assert float(((ra - rb).abs() > 0.05 * lr).float().mean()) < 2e-3
    for e in (a, b):
        S.run_lockstep([e], S.SHARD_MONITOR)            # a: forks again -- and finalize takes the engine from there
    HA, YA = a.eng.buffer("HA"), a.eng.buffer("YA")
    lab = torch.as_tensor(z["labels"], device="cuda")
    la = (lab[:, None] == lab[None, :]).float()
    fa, fb = a.eng.finalize(0, HA, YA, la), b.eng.finalize(0, HA, YA, la)
    assert float((fa - fb).abs().max()) <= 1e-4 * float(fb.abs().max())
    S.run_lockstep([a], S.SHARD_STEP); S.run_lockstep([b], S.SHARD_STEP)      # ... and the attack goes on
    assert a.eng.fused_steps() == b.eng.fused_steps() == 3


def test_ab_switches_are_ignored_without_MCGRA_AB(pkg, monkeypatch, capfd):
    """A variable left in the environment of a real run changes nothing: the engine reads its A/B switches only beside
    MCGRA_AB=1 (attack.hip: ab_env) and says on stderr what it ignored.  MCGRA_NO_FUSED_LR=1 would send every step through the
    general path, MCGRA_KEEP_GSYM=1 would keep the mirrored gradient."""
    z = _synthetic_case(1100, 11, (16, 16), 4, seed=3)
    monkeypatch.delenv("MCGRA_AB", raising=False)
    monkeypatch.setenv("MCGRA_NO_FUSED_LR", "1")
    monkeypatch.setenv("MCGRA_KEEP_GSYM", "1")
    eng = H.engine_from(pkg, z)
    eng.step()
    assert eng.fused_steps() == 1
    with pytest.raises(Exception, match="MCGRA_KEEP_GSYM"):
        eng.buffer("G_sym")
    err = capfd.readouterr().err
    assert "MCGRA_NO_FUSED_LR=1 is ignored" in err and "MCGRA_AB=1" in err
    monkeypatch.setenv("MCGRA_AB", "1")
    eng2 = H.engine_from(pkg, z)
    eng2.step()
    assert eng2.fused_steps() == 0 and eng2.path_stats()["lowrank_steps"] == 1      # (the unfused low-rank step of attack.hip)
    assert eng2.buffer("G_sym").shape[0] == 1100


def test_abandoned_row_block_step_is_dropped_cleanly(pkg):
    """A row-block step the caller gives up on after a failed collective (mcgra_attack_shard_begin again without having
    reached XCHG_DONE) has already enqueued its masked-pair post, forked the product and the small-operand terms: the
    next step drains that, takes the device's post counter and runs as if nothing had happened -- bit-identical to a
    rank that was never interrupted (everything an unfinished step writes is scratch)."""
    import torch
    from mc_gra_amd import sharded as S
    z = _synthetic_case(1100, 11, (16, 16), 4, seed=3)
    _, (a,) = _shard_engines(pkg, z, 1)
    _, (b,) = _shard_engines(pkg, z, 1)
    for t in range(2):
        a.begin(S.SHARD_STEP, False)
        # t = 0: the forward's three gathers, then [decode backward | first low-rank product], the second product and the
        # backward level's -- i.e. past the decode's post, the forked product and the side-stream terms
        kinds = [a.next()[0] for _ in range(6 if t == 0 else 3)]
        assert S.XCHG_DONE not in kinds
        S.run_lockstep([a], S.SHARD_STEP)                                # begun again from the top, run to the end
        S.run_lockstep([b], S.SHARD_STEP)
        assert torch.equal(a.eng.get_rows(), b.eng.get_rows()), t
        S.run_lockstep([a], S.SHARD_MONITOR); S.run_lockstep([b], S.SHARD_MONITOR)
    sa = S.run_lockstep([a], S.SHARD_STEP, want_scalars=True)[0]
    sb = S.run_lockstep([b], S.SHARD_STEP, want_scalars=True)[0]
    assert sa == sb and torch.equal(a.eng.get_rows(), b.eng.get_rows())
    assert a.eng.fused_steps() == b.eng.fused_steps() == 3


def test_sharded_stepper_on_a_one_rank_rccl_group(pkg):
    """RCCL on hardware: torch.distributed with backend "nccl" (= RCCL), world size 1, drives the product's
    ShardedStepper -- all_gather_into_tensor and all_to_all_single on views of the engine's arena are executed by the
    collective library, and the rank's rows equal the monolithic step's.  Eight collectives per step + monitor at L = 2
    (three gathers of the forward; [decode backward | first low-rank product], the second product, the backward level, the
    all-to-all of P1 tile blocks, gd), one more gather when the loss terms are asked for."""
    import socket
    import torch
    import torch.distributed as dist
    from mc_gra_amd import sharded as S
    z = _synthetic_case(1100, 11, (16, 16), 4, seed=3)
    mono = H.engine_from(pkg, z)
    plans, bks = _shard_engines(pkg, z, 1)
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0)); port = sck.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        st = S.ShardedStepper(bks[0], plans[0], dist=dist, clone_input=True, always=True)
        for t in range(2):
            mono.step(); mono.monitor()
            st.step(want_scalars=(t == 1)); st.monitor()
        assert st.exchanges == 2 * 8 + 3 + 1          # (+ 3: the first step finds no monitor forward to adopt and runs its own)
        rows, M = bks[0].eng.get_rows(), mono.buffer("M")
        assert float(((rows - M).abs() > 0.05 * float(z["lr"])).float().mean()) < 2e-3
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("case", ["s200_hsic_init", "s48_hsic_eps", "s80_hsic_l3", "s48_sage_hsic_init", "s200_hsic"])
def test_lowrank_hsic_matches_gram_path(pkg, case, monkeypatch):
    """Same steps through the low-rank path (default for a ReLU embedding) and through the Gram path
    (MCGRA_NO_LOWRANK=1): gradients agree to fp32 rounding and the fast path is the one that ran."""
    import torch
    z = H.load_case(case)
    fast = H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_NO_LOWRANK", "1")
    gram = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_NO_LOWRANK")
    for t in range(3):
        noise = H.noise_of(z, t)
        noise = None if noise is None else torch.tensor(noise, device="cuda")
        a = fast.step(want_scalars=True, noise=noise)
        b = gram.step(want_scalars=True, noise=noise)
        gf, gg = fast.buffer("G_sym").cpu().numpy(), gram.buffer("G_sym").cpu().numpy()
        assert np.abs(gf - gg).max() <= 2e-5 * np.abs(gg).max(), (t, np.abs(gf - gg).max(), np.abs(gg).max())
        for k in ("loss", "c1", "c2"):
            assert a[k] == pytest.approx(b[k], rel=2e-5, abs=1e-7), (t, k)
        gram.set_adj_changes(fast.get_adj_changes())          # keep both on the same trajectory
    assert fast.path_stats() == {"lowrank_steps": 3, "general_steps": 0}
    assert gram.path_stats() == {"lowrank_steps": 0, "general_steps": 3}


def test_lowrank_falls_back_when_decode_masks_a_pair(pkg):
    """Orthogonal or dead embedding rows make S_ij = 0: relu'(0) = 0 masks those pairs in the reference's
    backward, so the engine must take the Gram path on such a step (and still match the oracle)."""
    z = H.load_case("s48_hsic_init")
    w = H.weights_from(z)
    le = H.cfg_from(z).emb_nlayer - 1
    probe = H.oracle_from(z)
    probe.step()
    pre = probe.last["em"]                                     # relu output with the golden bias
    w.b = [b.copy() for b in w.b]
    w.b[le] = (w.b[le] - np.quantile(pre, 0.5, axis=0)).astype(np.float32)      # about half of the entries die
    orc = O.PGDAttackOracle(w, z["features"], z["adj"], np.zeros_like(z["adj"]), z["feature_adj"], z["labels"],
                            z["idx_attack"], H.cfg_from(z))
    orc.set_adj_changes(H.a0_of(z))
    eng = H.engine_from(pkg, z)
    eng.set_model(w.W, w.b, w.Wlin, w.blin, w.Ws)
    eng.step()
    orc.step()
    S = orc.last["S"]
    off = ~np.eye(S.shape[0], dtype=bool)
    assert ((S <= 0) & off).sum() > 0, "test precondition: some decode pairs must be masked"
    assert eng.path_stats() == {"lowrank_steps": 0, "general_steps": 1}
    g = O.pack_tril(eng.buffer("G_sym").cpu().numpy())
    gr = O.pack_tril(orc.last["G_sym"])
    assert np.abs(g - gr).max() <= 3e-4 * max(np.abs(gr).max(), 1e-30)


# ---- synthetic problems checked against the oracle directly (shapes the goldens do not cover) ------------------
_synthetic_case = H.synthetic_case


@pytest.mark.parametrize("n,widths,expect", [(97, (12, 12), "lowrank"), (130, (24, 8), "lowrank"),
                                             (96, (40, 40), "general"), (150, (16, 16, 16), "lowrank"),
                                             (1100, (16, 16), "lowrank"), (1283, (16, 8), "lowrank")])
def test_engine_matches_oracle_on_odd_shapes(pkg, n, widths, expect):
    """n not a multiple of 4 / 128, embedding widths outside {8, 16, 32} (unfused decode backward) and > 32 (Gram
    evaluation): teacher-forced per-step gradient against the oracle."""
    z = _synthetic_case(n, 11, widths, 4, seed=n)
    orc = H.oracle_from(z)
    eng = H.engine_from(pkg, z)
    for t in range(3):
        orc.step()
        eng.step()
        g = O.pack_tril(eng.buffer("G_sym").cpu().numpy())
        gr = O.pack_tril(orc.last["G_sym"])
        assert np.abs(g - gr).max() <= 3e-4 * np.abs(gr).max(), (t, np.abs(g - gr).max(), np.abs(gr).max())
        eng.set_adj_changes(O.pack_tril(orc.M))           # teacher forcing
    ps = eng.path_stats()
    assert (ps["lowrank_steps"], ps["general_steps"]) == ((3, 0) if expect == "lowrank" else (0, 3))
    # n >= 1024 takes the default split product (mode 3: two fp16 planes) unless the environment overrides it
    if "MCGRA_SPLIT_BF16" not in os.environ:
        assert eng.product_mode() == (3 if (n >= 1024 and expect == "lowrank") else 0)


@pytest.mark.parametrize("wp", [(0.01, 0, 0, 0, 0, 10, 10, 0, 10, 1000), (0, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000),
                                (0.01, 0.01, 0, 0, 0, 0, 0, 0, 0, 0)])
def test_lowrank_with_single_terms(pkg, wp, monkeypatch):
    """c1 only (no mask readback), c2 only (no N x N x N product at all), no entropy terms: low-rank == Gram."""
    z = H.load_case("s200_hsic_init")
    fast = H.engine_from(pkg, z, weight_param=wp)
    monkeypatch.setenv("MCGRA_NO_LOWRANK", "1")
    gram = H.engine_from(pkg, z, weight_param=wp)
    monkeypatch.delenv("MCGRA_NO_LOWRANK")
    for t in range(2):
        a, b = fast.step(want_scalars=True), gram.step(want_scalars=True)
        gf, gg = fast.buffer("G_sym").cpu().numpy(), gram.buffer("G_sym").cpu().numpy()
        assert np.abs(gf - gg).max() <= 2e-5 * np.abs(gg).max()
        assert a["loss"] == pytest.approx(b["loss"], rel=2e-5, abs=1e-7)
        gram.set_adj_changes(fast.get_adj_changes())
    assert fast.path_stats()["lowrank_steps"] == 2


# ---- split evaluations of the P1 product on the 16-bit matrix cores (split_bf16.hip, split_symm_bf16.hip) -----------
@pytest.mark.parametrize("mode", ["2", "3"])
@pytest.mark.parametrize("case", ["s200_hsic_init", "s48_hsic"])
def test_split_bf16_matches_fp32_path(pkg, case, mode, monkeypatch):
    """MCGRA_SPLIT_BF16=2/3: same gradients as the fp32 MFMA path to fp32 rounding (the splits keep 24 / 22
    mantissa bits; 3 = two fp16 planes, the default of graphs with n >= 1024)."""
    z = H.load_case(case)
    ref = H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_SPLIT_BF16", mode)      # 2: three bf16 planes, 3: two fp16 planes
    try:
        spl = H.engine_from(pkg, z)
        assert spl.product_mode() == int(mode)
    except Exception as e:                       # hipBLASLt missing on the box: the option refuses loudly
        pytest.skip(f"split path unavailable: {e}")
    finally:
        monkeypatch.delenv("MCGRA_SPLIT_BF16")
    for t in range(3):
        a, b = ref.step(want_scalars=True), spl.step(want_scalars=True)
        gr, gs = ref.buffer("G_sym").cpu().numpy(), spl.buffer("G_sym").cpu().numpy()
        assert np.abs(gs - gr).max() <= 2e-5 * np.abs(gr).max(), (t, np.abs(gs - gr).max(), np.abs(gr).max())
        assert b["loss"] == pytest.approx(a["loss"], rel=2e-5, abs=1e-7)
        spl.set_adj_changes(ref.get_adj_changes())
    # run-to-run determinism (for the library GEMM: no atomics in the chosen algorithm)
    monkeypatch.setenv("MCGRA_SPLIT_BF16", mode)
    e1, e2 = H.engine_from(pkg, z), H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_SPLIT_BF16")
    for t in range(2):
        e1.step(); e2.step()
    import torch
    assert torch.equal(e1.get_adj_changes(), e2.get_adj_changes())


def test_single_plane_product_is_a_named_mode_with_fp16_accuracy(pkg, monkeypatch):
    """MCGRA_SPLIT_BF16=1 (round 6; the arithmetic "bf16 MFMA" in BASELINE.json's configs[2] / [4] names, taken literally): the one
    N x N x N product of a fused low-rank step as the SINGLE plane product x0 y0 of the same fp16 x 2 operands -- the low planes
    are compiled out of the kernel, not zeroed.  Never a default: an engine created without the variable runs the 3-product split
    (product_mode 3 from n = 1024, the fp32 kernel below).  With it: product_mode 1, every step fused, and the first gradient
    sits at fp16 distance from the default's -- between 1e-5 (it IS a different arithmetic: the test would not see a mode that
    silently ran the default) and 5e-3 of its largest magnitude on a workload whose N x N terms carry the gradient."""
    import torch
    z = _synthetic_case(1283, 11, (16, 16), 4, seed=1283, weight_param=NXN_ONLY)
    ref = H.engine_from(pkg, z)
    assert ref.product_mode() == 3
    monkeypatch.setenv("MCGRA_SPLIT_BF16", "1")
    one = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_SPLIT_BF16")
    assert one.product_mode() == 1
    for t in range(3):
        a, b = ref.step(want_scalars=True), one.step(want_scalars=True)
        gr, gs = ref.buffer("G_sym").cpu().numpy(), one.buffer("G_sym").cpu().numpy()
        err = np.abs(gs - gr).max() / np.abs(gr).max()
        assert 1e-5 < err <= 5e-3, (t, err)
        assert b["loss"] == pytest.approx(a["loss"], rel=1e-3)
        gsf = one.buffer("G_sym")
        assert bool((gsf == gsf.T).all())
        one.set_adj_changes(ref.get_adj_changes())
    assert one.fused_steps() == 3 and ref.fused_steps() == 3
    small = H.engine_from(pkg, _synthetic_case(300, 11, (16, 16), 4, seed=300))
    assert small.product_mode() == 0


@pytest.mark.parametrize("case", ["s200_hsic_init", "s200_mse", "s48_kl", "s48_gat_hsic_init"])
def test_monitor_forward_reuse_is_bit_identical(pkg, case, monkeypatch):
    """The next step adopts the monitoring forward of :290-296 instead of recomputing it: same bits as without reuse,
    with or without a monitor call between steps, and set_adj_changes in between invalidates it."""
    import torch
    z = H.load_case(case)
    a = H.engine_from(pkg, z)                       # reuse on, monitor between steps
    monkeypatch.setenv("MCGRA_NO_FWD_REUSE", "1")
    b = H.engine_from(pkg, z)                       # reuse off, monitor between steps
    monkeypatch.delenv("MCGRA_NO_FWD_REUSE")
    c = H.engine_from(pkg, z)                       # reuse on, no monitor calls at all
    for t in range(4):
        ra, rb = a.step(want_scalars=True), b.step(want_scalars=True)
        c.step()
        (la, sa), (lb, sb) = a.monitor(want_sparsity=True), b.monitor(want_sparsity=True)
        assert ra == rb, t
        assert torch.equal(a.get_adj_changes(), b.get_adj_changes()) and torch.equal(a.get_adj_changes(), c.get_adj_changes())
        assert torch.equal(la, lb) and sa == sb
        if t == 1:                                   # stale cache must not survive a state change
            x = a.get_adj_changes() * 0.5
            for e in (a, b, c):
                e.set_adj_changes(x)
    assert torch.equal(a.buffer("adj_norm"), b.buffer("adj_norm"))      # what finalize (:300) decodes from


def test_side_stream_overlap_is_bit_identical(pkg, monkeypatch):
    """MCGRA_OVERLAP=1 only moves the N x N x N product onto the engine's own stream: same bits."""
    import torch
    z = H.load_case("s200_hsic_init")
    a = H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_OVERLAP", "1")
    b = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_OVERLAP")
    for t in range(3):
        ra, rb = a.step(want_scalars=True), b.step(want_scalars=True)
        a.monitor(); b.monitor()
        assert ra == rb and torch.equal(a.get_adj_changes(), b.get_adj_changes())


def test_side_stream_default_of_split_product_is_bit_identical(pkg, monkeypatch):
    """n >= 1024: the fp16-split product runs on the engine's own stream by default; MCGRA_OVERLAP=0 gives the same bits."""
    import torch
    z = _synthetic_case(1100, 11, (16, 8), 4, seed=3)
    a = H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_OVERLAP", "0")
    b = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_OVERLAP")
    assert a.product_mode() == 3 and b.product_mode() == 3
    for t in range(4):
        ra, rb = a.step(want_scalars=True), b.step(want_scalars=True)
        a.monitor(); b.monitor()
        assert ra == rb and torch.equal(a.get_adj_changes(), b.get_adj_changes())
    assert a.path_stats()["lowrank_steps"] == 4


@pytest.mark.parametrize("arith", ["bf16", "f16"])
@pytest.mark.parametrize("n", [33, 256, 257, 511, 1000, 1537, 2708, 4100])
def test_split_bf16_product_against_fp64(pkg, torch_, n, arith):
    """mcgra_ssymm_split_bf16 / _f16 at sizes around the 256-row panels, the 16-wide K chunks (the fp16 kernel steps
    two at a time: odd chunk counts are padded) and the split-K tail: error against fp64 within the fp32-product class
    (<= 1e-6 of |S||B|), asymmetric B exposes any transposition, repeat runs are bit-identical (the barrier structure
    of a new kernel is screened over several runs and sizes)."""
    import torch
    from mc_gra_amd import engine as E
    split = E.ssymm_split_bf16 if arith == "bf16" else E.ssymm_split_f16
    rng = np.random.RandomState(n)
    F = rng.randn(n, 24).astype(np.float32)
    S = (F @ F.T).astype(np.float32)
    S = (S + S.T) * 0.5
    X = (rng.rand(n, n).astype(np.float32) - 0.3) * 0.1            # NOT symmetric
    sub = rng.rand(n).astype(np.float32) * 0.05
    Sd, Xd, sd = dev(torch_, S), dev(torch_, X), dev(torch_, sub)
    outs = [split(Sd, Xd, sd).clone() for _ in range(4)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    B = (X.astype(np.float64) - sub.astype(np.float64)[:, None])
    ref = S.astype(np.float64) @ B.T
    scale = np.abs(S).astype(np.float64) @ np.abs(B).T
    err = np.abs(outs[0].cpu().numpy().astype(np.float64) - ref) / scale
    assert err.max() <= 1e-6, err.max()
    # the fp32 MFMA SYMM on the same operands (lower tile storage) for comparison of the error class
    Bt = torch.tensor(np.ascontiguousarray(B.T.astype(np.float32)), device="cuda")
    f32 = E.ssymm_lower(Sd, Bt).cpu().numpy().astype(np.float64)
    assert err.max() <= 4 * (np.abs(f32 - ref) / scale).max() + 2e-7


@pytest.mark.parametrize("scale", [1.0, 3.0e-12, 7.0e11])
def test_split_f16_product_operand_scales(pkg, torch_, scale):
    """The fp16 planes live in [2^-24, 2^16): the kernel's exact power-of-two operand scales must make the result
    independent of the operands' magnitude (here 1e-12 ... 1e12, far outside fp16) -- scaling both operands by powers
    of two gives bit-identical mantissas -- and elements 2^20 below their operand's maximum keep the error class."""
    import torch
    from mc_gra_amd import engine as E
    n = 700
    rng = np.random.RandomState(5)
    S = rng.randn(n, n).astype(np.float32) * np.exp2(rng.randint(-20, 1, (n, n))).astype(np.float32)
    S = (S + S.T) * 0.5
    X = rng.randn(n, n).astype(np.float32) * np.exp2(rng.randint(-20, 1, (n, n))).astype(np.float32)
    Ss = (S * np.float32(scale)).astype(np.float32)
    Xs = (X * np.float32(4.0 / scale)).astype(np.float32)
    got = E.ssymm_split_f16(dev(torch_, Ss), dev(torch_, Xs)).cpu().numpy()
    ref = Ss.astype(np.float64) @ Xs.astype(np.float64).T
    den = np.abs(Ss).astype(np.float64) @ np.abs(Xs).astype(np.float64).T
    assert (np.abs(got - ref) / den).max() <= 1e-6
    # power-of-two rescaling of the operands only shifts exponents
    base = E.ssymm_split_f16(dev(torch_, S), dev(torch_, X)).cpu().numpy()
    sc = E.ssymm_split_f16(dev(torch_, S * np.float32(2.0 ** 31)), dev(torch_, X * np.float32(2.0 ** -40))).cpu().numpy()
    assert np.array_equal(sc, base * np.float32(2.0 ** -9))
    # an all-zero operand is a zero product, not a NaN
    z = E.ssymm_split_f16(dev(torch_, np.zeros((n, n), np.float32)), dev(torch_, X)).cpu().numpy()
    assert np.array_equal(z, np.zeros_like(z))


@pytest.mark.parametrize("n,measure,widths", [(700, "HSIC", (16, 8)), (1100, "HSIC", (16, 8)), (700, "MSELoss", (16, 8)),
                                              (700, "HSIC", (80, 80)), (900, "MSELoss", (48, 40)), (600, "HSIC", (128, 128)),
                                              (4200, "HSIC", (80, 80))])      # (n > 4096: the products' split-K slabs share G_A with the tail's row sums)
def test_fused_tail_matches_separate_kernels(pkg, n, measure, widths, monkeypatch):
    """n >= 256: normalisation-backward apply + rank-k update + gradient mirror + Adam run as one kernel over the lower
    tile pairs (k_rankk_apply_adam); MCGRA_NO_FUSED_TAIL=1 keeps rankk_nt + k_adam_sym.  Same per-element arithmetic:
    the mirrored gradient agrees to rounding of the final sum and the state stays symmetric bit for bit.
    Chains wider than 64 columns in all (a GAT victim's two layers of 5 x 16: 160) take their panels through LDS 64 columns at a
    time -- two, three and four rounds here; the separate kernels then are the apply pass, the fp32 MFMA GEMM and k_adam_sym."""
    import torch
    z = _synthetic_case(n, 11, widths, 4, seed=n, measure=measure)
    monkeypatch.setenv("MCGRA_NO_FUSED_LR", "1")       # both through the general path (n >= 1024 would take the fused step)
    a = H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_NO_FUSED_TAIL", "1")
    b = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_NO_FUSED_TAIL")
    monkeypatch.delenv("MCGRA_NO_FUSED_LR")
    for t in range(3):
        a.step(); b.step()
        assert a.fused_steps() == 0 and b.fused_steps() == 0
        ga, gb = a.buffer("G_sym"), b.buffer("G_sym")
        assert float((ga - gb).abs().max()) <= 1e-6 * float(gb.abs().max()), t
        ma = a.buffer("M")
        assert torch.equal(ma[:n, :n], ma[:n, :n].T)
        if sum((w + 3) // 4 * 4 for w in widths) <= 64:
            assert float((a.get_adj_changes() - b.get_adj_changes()).abs().max()) <= 1e-6
        else:
            # (the separate kernels of a wide chain sum the rank-k update on the matrix cores, in another order: Adam turns a gradient
            #  entry's rounding into lr * dg / |g| of the state -- entries whose gradient is not itself rounding noise agree, the rest
            #  stay inside one Adam step of each other)
            d = (ma[:n, :n] - b.buffer("M")[:n, :n]).abs()
            big = gb[:n, :n].abs() >= 1e-2 * float(gb.abs().max())
            assert float(d[big].max()) <= 2e-6 and float(d.max()) <= 2 * 0.01, (t, float(d[big].max()), float(d.max()))
        b.set_adj_changes(a.get_adj_changes())


# ---- the fused low-rank step (attack_fused.hip): N x N quantities from M and n-vectors only -------------------------
@pytest.mark.parametrize("n,widths", [(1100, (16, 16)), (1283, (16, 8)), (1030, (16, 16, 16))])
def test_skinny_products_from_the_operand_planes(pkg, n, widths, monkeypatch):
    """planes_mm.hip (default from n = 8192, forced on here): the skinny products of the fused step that run beside the
    N x N x N product read the fp16 planes of its operand, M W = R^-1 adj_norm (R^-1 W) - W at 22 significant bits, instead
    of M through the fp32 kernel.  Same step, same start: the mirrored gradient agrees to 1e-5 of its largest magnitude,
    the loss terms to 1e-6, the state stays symmetric bit for bit."""
    import torch
    z = _synthetic_case(n, 11, widths, 4, seed=n)
    monkeypatch.setenv("MCGRA_PLANES_MM", "1")
    a = H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_PLANES_MM", "0")
    b = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_PLANES_MM")
    for t in range(3):
        sa = a.step(want_scalars=True); a.monitor()
        sb = b.step(want_scalars=True); b.monitor()
        ga, gb = a.buffer("G_sym"), b.buffer("G_sym")
        assert float((ga - gb).abs().max()) <= 1e-5 * float(gb.abs().max()), t
        for k in ("loss", "c1", "c2", "c9", "c10"):
            assert sa[k] == pytest.approx(sb[k], rel=1e-6, abs=1e-6 * abs(sb["loss"])), (t, k)
        M = a.buffer("M")
        assert torch.equal(M, M.t())
        b.set_adj_changes(a.get_adj_changes())
    assert a.fused_steps() == 3 and b.fused_steps() == 3


@pytest.mark.parametrize("n,widths", [(1100, (16, 16)), (1283, (16, 8)), (1100, (16, 16, 16))])
def test_forward_post_pass_in_one_launch_gives_the_separate_kernels_bits(pkg, n, widths, monkeypatch):
    """The forward of the fused step runs a layer's post pass together with what follows it on the same rows -- the next
    layer's T of both chains and the next product's right-hand side, or both linear heads with their log-softmax -- in one
    launch (k_fl_post_fused).  Same operations in the same order as the separate kernels (MCGRA_NO_FUSED_POST=1): the whole
    state after three free-running steps, every loss term and the monitoring forward's output are identical bit for bit,
    also with three layers and unequal widths."""
    import torch
    z = _synthetic_case(n, 11, widths, 4, seed=n)
    a = H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_NO_FUSED_POST", "1")
    b = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_NO_FUSED_POST")
    for t in range(3):
        sa = a.step(want_scalars=True); oa, _ = a.monitor()
        sb = b.step(want_scalars=True); ob, _ = b.monitor()
        assert sa == sb, (t, sa, sb)
        assert torch.equal(oa, ob)
        for name in ("M", "G_sym", "em"):
            assert torch.equal(a.buffer(name), b.buffer(name)), (t, name)
    assert a.fused_steps() == 3 and b.fused_steps() == 3


@pytest.mark.parametrize("n,widths,wp,split", [
    (1100, (16, 16), None, None), (1283, (16, 8), NXN_ONLY, None), (700, (16, 16), NXN_ONLY, "3"),
    (1100, (16, 16, 16), None, None), (515, (8, 8), (0, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000), "2"),
    (1030, (16, 16), (0.01, 0, 0, 0, 0, 10, 0, 0, 10, 0), None)])
def test_fused_lowrank_step_matches_general_path_and_oracle(pkg, n, widths, wp, split, monkeypatch):
    """Same step through attack_fused.hip and through the general path (MCGRA_NO_FUSED_LR=1), teacher-forced, with the
    monitor forward adopted in between: mirrored gradient, every loss term and the updated adjacency agree to fp32
    rounding; and the fused gradient matches the ORACLE (with the small-operand terms off the N x N terms carry the
    whole gradient, so this pins the restructured algebra to the reference's algorithm)."""
    kw = {} if wp is None else {"weight_param": wp}
    z = _synthetic_case(n, 11, widths, 4, seed=n, **kw)
    if split:
        monkeypatch.setenv("MCGRA_SPLIT_BF16", split)
    fused = H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_NO_FUSED_LR", "1")
    gen = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_NO_FUSED_LR")
    orc = H.oracle_from(z)
    for t in range(3):
        a, b = fused.step(want_scalars=True), gen.step(want_scalars=True)
        orc.step()
        gf, gg = fused.buffer("G_sym").cpu().numpy(), gen.buffer("G_sym").cpu().numpy()
        gr = orc.last["G_sym"]
        scale = np.abs(gr).max()
        assert np.abs(gf - gg).max() <= 3e-5 * scale, (t, np.abs(gf - gg).max() / scale)
        assert np.abs(gf - gr).max() <= 3e-4 * scale, (t, np.abs(gf - gr).max() / scale)
        for k in ("loss", "c1", "c2", "c6", "c7", "c9", "c10", "nll", "origin_loss"):
            assert a[k] == pytest.approx(b[k], rel=2e-5, abs=1e-6 * max(1.0, abs(b["loss"]))), (t, k, a[k], b[k])
        if wp is NXN_ONLY:      # (with c9 / c10 on, the oracle's Gram-form fp32 VALUE of the small-operand HSIC is the noisy one)
            assert abs(a["loss"] - orc.last["loss"]) <= 2e-4 * abs(orc.last["loss"]) + 1e-5
        Mf = fused.buffer("M")
        assert bool((Mf == Mf.T).all()), "the learnable adjacency must stay symmetric bit for bit"
        gsf = fused.buffer("G_sym")
        assert bool((gsf == gsf.T).all())
        ma, mb = fused.get_adj_changes(), gen.get_adj_changes()
        lr = float(z["lr"])
        assert float(((ma - mb).abs() > 0.05 * lr).float().mean()) < 2e-3       # Adam: +-lr on noise-level gradients
        la, _ = fused.monitor(); lb, _ = gen.monitor()
        assert float((la - lb).abs().max()) < 1e-4
        nxt = O.pack_tril(orc.M)
        fused.set_adj_changes(nxt); gen.set_adj_changes(nxt)                   # teacher forcing (drops the adopted forward)
    assert fused.fused_steps() == 3 and gen.fused_steps() == 0
    assert fused.path_stats() == gen.path_stats() == {"lowrank_steps": 3, "general_steps": 0}


def test_fused_lowrank_adopts_the_monitor_forward_bit_identically(pkg, monkeypatch):
    """Free-running fused steps with and without monitor calls in between give the same bits (the monitor's forward
    on the updated adjacency IS the next step's forward), and finalize after a fused loop equals finalize after the
    same loop on the general path to fp32 rounding (em_last stands in for embedding(features, adj_norm))."""
    import torch
    z = _synthetic_case(1100, 11, (16, 16), 4, seed=5)
    e1, e2 = H.engine_from(pkg, z), H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_NO_FUSED_LR", "1")
    e3 = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_NO_FUSED_LR")
    for t in range(3):
        e1.step(); e1.monitor()
        e2.step()
        e3.step(); e3.monitor()
    assert torch.equal(e1.get_adj_changes(), e2.get_adj_changes())
    assert e1.fused_steps() == 3
    lab = z["labels"]
    label_adj = (lab[:, None] == lab[None, :]).astype(np.float32)
    o = H.oracle_from(z)
    HA, YA = o.HA, o.YA
    f1 = e1.finalize(0, HA, YA, label_adj).cpu().numpy()
    f3 = e3.finalize(0, HA, YA, label_adj).cpu().numpy()
    assert np.mean(np.abs(f1 - f3) > 1e-3) < 0.01
    assert abs(O.metric_pool(z["adj"], f1, z["idx_attack"]) - O.metric_pool(z["adj"], f3, z["idx_attack"])) < 1e-4


@pytest.mark.parametrize("wp", [(0.01, 0.01, 0, 0, 0, 10, 10, 0, 0, 0), (0.01, 0, 0, 0, 0, 10, 10, 0, 0, 0),
                                (0, 0.01, 0, 0, 0, 0, 10, 10, 0, 0)], ids=["c1c2", "c1only", "c2only"])
def test_gram_evaluation_on_the_split_kernel_matches_fp32_symm(pkg, monkeypatch, wp):
    """Steps the low-rank forms do not cover evaluate HSIC from the Grams; for n >= 1024 their four N x N x N products
    run on the 2-plane fp16 kernel (mcgra_attack_gram_split_steps).  Same gradient, values and update as the fp32 SYMM
    evaluation of the same step (MCGRA_GRAM_SPLIT=0) to fp32 rounding, and as the oracle."""
    import torch
    z = _synthetic_case(1100, 11, (16, 16), 4, seed=21, weight_param=wp)
    monkeypatch.setenv("MCGRA_NO_LOWRANK", "1")
    split = H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_GRAM_SPLIT", "0")
    f32 = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_GRAM_SPLIT"); monkeypatch.delenv("MCGRA_NO_LOWRANK")
    o = H.oracle_from(z)
    for t in range(3):
        a, b = split.step(want_scalars=True), f32.step(want_scalars=True)
        o.step()
        ga, gb = split.buffer("G_sym"), f32.buffer("G_sym")
        scale = float(gb.abs().max())
        assert float((ga - gb).abs().max()) / scale < 2e-5, t
        # (not for c2 alone: that gradient is a small difference of large Gram sums, any fp32 evaluation of it carries
        # 1e-3 .. 1e-2 of noise -- the two HIP evaluations above agree 100 x closer than either does with numpy's)
        if wp[0] != 0:
            g_or = torch.from_numpy(o.last["G_sym"]).to(ga.device)
            assert float((ga - g_or).abs().max()) / scale < 1e-4, t
        for k in ("loss", "c1", "c2", "c6", "c7"):
            assert a[k] == pytest.approx(b[k], rel=2e-5, abs=1e-7), (t, k)
        f32.set_adj_changes(split.get_adj_changes())
        o.set_adj_changes(split.get_adj_changes().cpu().numpy())
    assert split.gram_split_steps() == 3 and f32.gram_split_steps() == 0
    assert split.path_stats() == f32.path_stats() == {"lowrank_steps": 0, "general_steps": 3}


def test_cka_steps_of_a_large_graph_use_the_split_kernel(pkg, monkeypatch):
    """linear_CKA (:486) at n >= 1024: the two Grams and the two gradient products of a step on the 2-plane fp16 kernel;
    same gradient and values as the fp32 SYRK / SYMM evaluation (MCGRA_GRAM_SPLIT=0) and as the oracle."""
    import torch
    z = _synthetic_case(1100, 11, (16, 16), 4, seed=33, measure="CKA", weight_param=(1.0, 1.0, 0, 0, 0, 10, 10, 0, 0, 0))
    split = H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_GRAM_SPLIT", "0")
    f32 = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_GRAM_SPLIT")
    o = H.oracle_from(z)
    for t in range(2):
        a, b = split.step(want_scalars=True), f32.step(want_scalars=True)
        o.step()
        ga, gb = split.buffer("G_sym"), f32.buffer("G_sym")
        scale = float(gb.abs().max())
        assert float((ga - gb).abs().max()) / scale < 2e-5, t
        # (the oracle only loosely: the CKA gradient is the difference of two normalised Gram terms that nearly cancel,
        # numpy's fp32 evaluation of it carries ~1e-3 of noise; the two HIP evaluations agree 100 x closer)
        g_or = torch.from_numpy(o.last["G_sym"]).to(ga.device)
        assert float((ga - g_or).abs().max()) / scale < 1e-2, (t, float((ga - g_or).abs().max()) / scale)
        for k in ("loss", "c1", "c2"):
            assert a[k] == pytest.approx(b[k], rel=2e-5, abs=1e-7), (t, k)
        f32.set_adj_changes(split.get_adj_changes())
        o.set_adj_changes(split.get_adj_changes().cpu().numpy())
    assert split.gram_split_steps() == 2 and f32.gram_split_steps() == 0


def test_masked_steps_of_a_large_graph_use_the_split_gram_evaluation(pkg, monkeypatch):
    """n >= 1024 with a decode that masks pairs: the fused step hands over, the Gram evaluation runs on the split kernel
    and agrees with the fp32 SYMM evaluation of the same steps."""
    import torch
    z = _synthetic_case(1100, 11, (16, 16), 4, seed=9, weight_param=(0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 0))
    w = H.masked_weights(z)      # about half of em dies: the decode masks pairs
    engs = []
    for gs in ("1", "0"):
        monkeypatch.setenv("MCGRA_GRAM_SPLIT", gs)
        e = H.engine_from(pkg, z)
        e.set_model(w.W, w.b, w.Wlin, w.b and w.blin, w.Ws)
        engs.append(e)
    monkeypatch.delenv("MCGRA_GRAM_SPLIT")
    for t in range(2):
        a, b = [e.step(want_scalars=True) for e in engs]
        ga, gb = engs[0].buffer("G_sym"), engs[1].buffer("G_sym")
        assert float((ga - gb).abs().max()) / float(gb.abs().max()) < 2e-5, t
        for k in ("loss", "c1", "c2"):
            assert a[k] == pytest.approx(b[k], rel=2e-5, abs=1e-7), (t, k)
        engs[1].set_adj_changes(engs[0].get_adj_changes())
    assert engs[0].gram_split_steps() == 2 and engs[1].gram_split_steps() == 0
    assert engs[0].path_stats() == {"lowrank_steps": 0, "general_steps": 2} and engs[0].fused_steps() == 0
    # once the decode no longer masks a pair the fused step takes over again (the masked-pair counter the general path
    # leaves behind must not stick)
    w0 = H.weights_from(z)
    engs[0].set_model(w0.W, w0.b, w0.Wlin, w0.blin, w0.Ws)
    engs[0].step(); engs[0].monitor(); engs[0].step()
    assert engs[0].fused_steps() == 2 and engs[0].path_stats()["general_steps"] == 2


def test_fused_lowrank_hands_masked_steps_to_the_general_path(pkg, monkeypatch):
    """A decode that masks a pair (S_ij <= 0) cannot take the low-rank algebra: the fused step detects it from Zn and
    the general (Gram) path redoes the step -- same result as an engine that never tries the fused step."""
    import torch
    z = _synthetic_case(300, 11, (16, 16), 4, seed=9, weight_param=(0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 0))
    monkeypatch.setenv("MCGRA_SPLIT_BF16", "3")
    w = H.masked_weights(z)      # about half of em dies: the decode masks pairs
    engs = []
    for no_fused in ("0", "1"):
        monkeypatch.setenv("MCGRA_NO_FUSED_LR", no_fused)
        e = H.engine_from(pkg, z)
        e.set_model(w.W, w.b, w.Wlin, w.b and w.blin, w.Ws)
        e.step(); e.monitor(); e.step()
        engs.append(e)
    assert engs[0].path_stats() == engs[1].path_stats() == {"lowrank_steps": 0, "general_steps": 2}
    assert engs[0].fused_steps() == 0
    assert torch.equal(engs[0].get_adj_changes(), engs[1].get_adj_changes())


def test_set_graph_without_ori_restores_the_fused_path(pkg):
    """set_graph with a non-zero ori_adj sends the handle to the general step (the low-rank / fused forms assume modified_adj ==
    M); a later set_graph WITHOUT ori_adj on the same handle must get the create-time paths back (it stayed on
    the general step for good, 3x slower at N = 10 000, with no indication) -- and give the gradient bits of a fresh engine."""
    import torch
    z = _synthetic_case(1100, 11, (16, 16), 4, seed=5)
    fresh = H.engine_from(pkg, z)
    eng = H.engine_from(pkg, z)
    ori = np.triu((np.random.RandomState(1).rand(1100, 1100) < 0.01).astype(np.float32), 1)
    ori = ori + ori.T
    eng.set_graph(z["features"], z["adj"], ori, z["feature_adj"], z["labels"], z["idx_attack"])
    eng.set_adj_changes(H.a0_of(z))
    eng.step()
    assert eng.fused_steps() == 0 and eng.path_stats()["general_steps"] == 1
    eng.set_graph(z["features"], z["adj"], None, z["feature_adj"], z["labels"], z["idx_attack"])
    eng.set_adj_changes(H.a0_of(z))
    eng.step(); fresh.step()
    assert eng.fused_steps() == 1 and fresh.fused_steps() == 1 and eng.path_stats()["general_steps"] == 1
    # (the Adam moments of the handle have seen the ori step, so only the gradient -- which does not -- is compared)
    assert torch.equal(eng.buffer("G_sym"), fresh.buffer("G_sym"))
    eng.monitor(); eng.step()
    assert eng.fused_steps() == 2


def test_early_pack_beside_the_forward_is_bit_identical(pkg, monkeypatch):
    """The planes of the N x N x N product's operand are packed on the product's stream as soon as r is known, beside the
    forward's two skinny products (attack_fused.hip: early pack), instead of behind them on the caller's stream
    (MCGRA_EARLY_PACK=0): same kernels on the same data, so the same bits -- also when the adjacency is replaced while a
    monitor call's pack is still in flight (the stale planes must be dropped, not multiplied), and through finalize."""
    import torch
    z = _synthetic_case(1283, 11, (16, 16), 4, seed=21)
    early = H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_EARLY_PACK", "0")
    late = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_EARLY_PACK")
    a1 = H.init_adj_changes(1283, 99, 0.03)
    for t in range(4):
        for e in (early, late):
            e.step(); e.monitor()
            if t == 1:
                e.set_adj_changes(a1)          # the monitor call above has forked a pack of the OLD adjacency
        assert torch.equal(early.get_adj_changes(), late.get_adj_changes()), t
        assert torch.equal(early.buffer("G_sym"), late.buffer("G_sym")), t
    assert early.fused_steps() == 4 and late.fused_steps() == 4
    lab = z["labels"]
    la = (lab[:, None] == lab[None, :]).astype(np.float32)
    fa = early.finalize(0, early.buffer("HA"), early.buffer("YA"), la)
    fb = late.finalize(0, late.buffer("HA"), late.buffer("YA"), la)
    assert torch.equal(fa, fb)


def _few_masked_pairs_case(n=1100, seed=31, n_high=2, weight_param=None):
    """A synthetic case whose decode relu-masks a FEW pairs exactly and has no dead embedding row.  The second GCN layer's
    input is (almost) rank one, T_1 = 1 u^T + 1e-3 noise, so em_i = relu(s_i u + b) with s_i = rowsum_i(modified_adj): the
    first eight coordinates are alive for s_i < 50 (u = -1, b = 50), the last eight for s_i > 15 (u = +1, b = -15).  The
    seeded start has row sums ~ 27 (all sixteen alive); three rows are scaled down (s ~ 6: first eight only) and two up
    (s ~ 80: last eight only): their 3 x 2 cross pairs have embeddings with disjoint supports, S_ij == 0 exactly."""
    z = _synthetic_case(n, 11, (16, 16), 4, seed=seed, **({"weight_param": weight_param} if weight_param else {}))
    z["lr"] = np.array(1e-4)                                   # (Adam moves every row sum by <= n lr = 0.11 per step: the supports stay)
    rng = np.random.RandomState(seed)
    z["W0"][:, 0] = 0.0
    z["b0"][0] = 1.0                                           # H_0[:, 0] == 1 on every node
    u = np.concatenate([-np.ones(8), np.ones(8)]).astype(np.float32)
    W1 = (rng.randn(16, 16) * 1e-3).astype(np.float32)
    W1[0] += u
    z["W1"] = W1
    z["b1"] = np.concatenate([50.0 * np.ones(8), -15.0 * np.ones(8)]).astype(np.float32)
    A = O.unpack_sym(H.a0_of(z), n)
    f = np.ones(n, np.float32)
    low = [5, 400, 901]
    high = [9, 777] if n_high == 2 else [9 + 10 * k for k in range(n_high)]
    f[low] = 0.2; f[high] = 3.0
    A = (A * f[:, None] * f[None, :]).astype(np.float32)
    return z, O.pack_tril(A), low, high


@pytest.mark.parametrize("n_high,wp", [(2, None), (100, None), (100, NXN_ONLY), (100, (0.0, 1.0, 0, 0, 0, 0, 0, 0, 0, 0))])
def test_relu_masked_pairs_of_live_rows_keep_the_fused_step(pkg, monkeypatch, n_high, wp):
    """the fused path was a cliff -- ONE relu-masked decode pair sent the step to the Gram evaluation,
    3x slower at N = 10 000.  It need not: with a ReLU embedding zn >= 0, so a masked pair has S_ij == 0 EXACTLY -- the value
    of modified_adj1 = offdiag relu(S) is still Z Z^T - D, every forward quantity of the low-rank step stands -- and what
    relu'(0) = 0 removes from the decode backward, g_ij zn_j on row i, lies on coordinates where em_i is zero (disjoint
    supports), which the embedding layer's own ReLU backward masks anyway; its component along zn_i is zn_i . zn_j = 0.
    So only a DEAD row (em_i == 0) voids the algebra (it takes |zn_i| = 1).  Here the decode masks 12 or 600 pairs on every
    step, no row is dead: the step stays fused, and its gradient equals the Gram evaluation's (MCGRA_NO_LOWRANK=1: the
    reference's formulation, which masks per pair) to 3e-5 and the oracle's to 3e-4 of the gradient's largest magnitude --
    with the README weights, with the N x N terms alone and with c2 alone (where the decode backward IS the gradient)."""
    import torch
    z, a0, low, high = _few_masked_pairs_case(n_high=n_high, weight_param=wp)
    orc = H.oracle_from(z)
    orc.set_adj_changes(a0)
    fused = H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_NO_LOWRANK", "1")
    gram = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_NO_LOWRANK")
    for e in (fused, gram):
        e.set_adj_changes(a0)
    for t in range(3):
        orc.step()
        S = orc.last["S"]
        off = ~np.eye(S.shape[0], dtype=bool)
        assert int(((S <= 0) & off).sum()) == 2 * len(low) * len(high) and float(np.diag(S).min()) > 0.5, "the case must mask pairs, no dead row"
        a = fused.step(want_scalars=True); fused.monitor()
        b = gram.step(want_scalars=True); gram.monitor()
        gf, gg = fused.buffer("G_sym"), gram.buffer("G_sym")
        gmax = float(gg.abs().max())
        assert float((gf - gg).abs().max()) <= 3e-5 * gmax, t
        go = torch.as_tensor(orc.last["G_sym"], device=gf.device)
        assert float((gf - go).abs().max()) <= 3e-4 * float(go.abs().max()), t
        for k in ("loss", "c1", "c2", "c7", "c9"):
            assert a[k] == pytest.approx(b[k], rel=3e-4, abs=1e-6), (t, k)
        gram.set_adj_changes(fused.get_adj_changes())           # teacher forcing: one state, two evaluations
        orc.set_adj_changes(fused.get_adj_changes().cpu().numpy())
    assert fused.fused_steps() == 3 and fused.masked_fused_steps() == 3 and fused.path_stats()["general_steps"] == 0
    assert gram.path_stats()["general_steps"] == 3 and gram.masked_fused_steps() == 0


KDE_OPS = np.load(os.path.join(H.GOLDEN, "ops_kde.npz"))


@pytest.mark.parametrize("tag", [str(c) for c in KDE_OPS["cases"]])
def test_mutual_information_op_against_the_reference(pkg, torch_, tag):
    """mcgra_mutual_information (measure KDE's `calc`, utils.py:980-1049) on the operand shapes of its call sites -- N x N with
    num_bins = N (values in [0, 1], and up to 2 as with a non-zero ori_adj), embeddings, log-probs against a softmax, a wide
    value range -- against the reference module's value and autograd gradients: within 3e-6 of its float64 run, and of its
    float32 run within that run's own distance from float64.  Also through the reference's class surface (utils.MutualInformation)."""
    from mc_gra_amd import engine as E
    from mc_gra_amd import utils as U
    z = KDE_OPS
    X, Y = dev(torch_, z[f"{tag}_x"]), dev(torch_, z[f"{tag}_y"])
    val, gx, gy = E.mutual_information(X, Y, want_grad=True)
    for g, nm in ((gx, "gx"), (gy, "gy")):
        g = g.cpu().numpy()
        g64 = z[f"{tag}_{nm}64"].astype(np.float64)
        ref_true = np.abs(z[f"{tag}_{nm}"] - g64).max() / np.abs(g64).max()
        assert np.abs(g - g64).max() / np.abs(g64).max() <= 3e-6, (tag, nm)
        assert np.abs(g - z[f"{tag}_{nm}"]).max() / np.abs(g64).max() <= ref_true + 3e-6, (tag, nm, ref_true)
    v64 = float(z[f"{tag}_val64"][0])
    assert abs(float(val) - v64) <= 3e-6 * max(1.0, abs(v64))
    mi = U.MutualInformation(sigma=0.4, num_bins=X.shape[1], normalize=True)
    out = mi(X, Y)
    assert tuple(out.shape) == (1,) and float(out[0]) == float(val)


@pytest.mark.parametrize("n,widths,nclass", [(1100, (16, 16), 4), (1030, (24, 32, 8), 9)])
def test_kde_steps_of_a_large_graph_match_the_oracle(pkg, torch_, n, widths, nclass):
    """measure KDE at n >= 1024 (the launch shapes of large graphs; a 3-layer victim with a 32-wide embedding and 9 classes:
    the 32-column form of the joint) with all four terms: per-step gradient against the oracle (which the reference fixtures
    pin: attack_*_kde*.npz, README lines 13 / 133), scalars, free-run state."""
    z = H.synthetic_case(n, 11, widths, nclass, seed=n, measure="KDE")
    eng, o = H.engine_from(pkg, z), H.oracle_from(z)
    for t in range(3):
        sc = eng.step(want_scalars=True)
        o.step()
        g, g_or = eng.buffer("G_sym").cpu().numpy(), o.last["G_sym"]
        assert np.abs(g - g_or).max() <= 1e-4 * np.abs(g_or).max(), (t, np.abs(g - g_or).max(), np.abs(g_or).max())
        # the N x N terms reach the first KDE_NXN_COLS columns of adj_norm / modified_adj1 only
        assert abs(sc["loss"] - o.last["loss"]) <= 2e-4 * abs(o.last["loss"]) + 1e-5
        # (the normalised mutual information 2 (H1 + H2 - H12) / (H1 + H2) of the N x N operands is ~ 5e-6: a difference of
        # entropies that agree to six digits -- its float32 kernel values bound the VALUE to ~ 2e-8 absolute, times the weight)
        wp = z["weight_param"]
        K = {"c1": wp[0] * 1e5, "c2": wp[1] * 1e5, "c9": wp[8], "c10": wp[9]}
        for k in ("c1", "c2", "c9", "c10"):
            assert sc[k] == pytest.approx(o.last["terms"][k], rel=1e-4, abs=2e-8 * K[k]), (t, k)
        o.set_adj_changes(eng.get_adj_changes().cpu().numpy())
    assert eng.path_stats() == {"lowrank_steps": 0, "general_steps": 0} and eng.fused_steps() == 0
    eng.close()


@pytest.mark.parametrize("case", ["s200_hsic_init", "s48_kl", "s48_gat_hsic_init", "s48_cka_init", "s300_mse_eps"])
def test_small_operand_terms_beside_the_general_step_are_bit_identical(pkg, case, monkeypatch):
    """The general step runs its small-operand terms c9 / c10 (~20 launches that need the forward only) on the third stream,
    forked behind the forward chains and joined in front of the modified_adj chain's backward (round 6; the fused steps have
    done so since round 3).  Same launches in the same order on the caller's stream (MCGRA_SMALL_SIDE=0): the same bits --
    gradient, loss terms and state over three steps, on HSIC (low-rank general step), KL, a GAT victim, CKA and an eps != 0 run."""
    import torch
    z = H.load_case(case)
    outs = []
    for side in ("1", "0"):
        monkeypatch.setenv("MCGRA_SMALL_SIDE", side)
        monkeypatch.setenv("MCGRA_NO_FUSED_LR", "1")
        eng = H.engine_from(pkg, z)
        res = []
        nsteps = min(3, z["noise"].shape[0]) if "noise" in z else 3
        for t in range(nsteps):
            nz = H.noise_of(z, t)
            sc = eng.step(noise=None if nz is None else torch.as_tensor(nz, device="cuda:0"), want_scalars=True)
            eng.monitor()
            res.append((eng.buffer("G_sym").clone(), sc))
        outs.append((res, eng.buffer("M").clone()))
        eng.close()
    for t in range(len(outs[0][0])):
        assert torch.equal(outs[0][0][t][0], outs[1][0][t][0]), t
        assert outs[0][0][t][1] == outs[1][0][t][1], t
    assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("measure,n", [("MSELoss", 1100), ("KL", 1100), ("MSELoss", 2050)])
def test_elementwise_fused_step_streams_on_a_small_graph_are_bit_identical(pkg, measure, n, monkeypatch):
    """A fused MSELoss / KL step has no factor chain beside its decode: below n = 4096 the decode stays on the caller's stream
    (a fork and a join cost it ~17 us each, and it would only wait), and MSELoss' two one-launch small-operand terms do too, with
    the zero fills of what they accumulate into riding in the row normalisation's launch (round 6: Cora-shaped MSELoss
    0.248 -> 0.194 ms, KL 0.307 -> 0.271).  The same work on the fourth / third stream (MCGRA_MSE_DECODE_SIDE=1,
    MCGRA_MSE_SMALL_INLINE=0: the form of rounds 3 - 5): the same bits -- gradient, loss terms and state over four steps."""
    import torch
    z = _synthetic_case(n, 11, (16, 16), 4, seed=n, measure=measure)
    outs = []
    for side in (False, True):
        if side:
            monkeypatch.setenv("MCGRA_MSE_DECODE_SIDE", "1"); monkeypatch.setenv("MCGRA_MSE_SMALL_INLINE", "0")
        eng = H.engine_from(pkg, z)
        res = []
        for t in range(4):
            sc = eng.step(want_scalars=True); eng.monitor()
            res.append((eng.buffer("G_sym").clone(), sc))
        assert eng.fused_steps() == 4
        outs.append((res, eng.buffer("M").clone()))
        eng.close()
    for t in range(4):
        assert torch.equal(outs[0][0][t][0], outs[1][0][t][0]), t
        assert outs[0][0][t][1] == outs[1][0][t][1], t
    assert torch.equal(outs[0][1], outs[1][1])


def test_gram_kx_forked_by_the_monitoring_forward_is_bit_identical(pkg, monkeypatch):
    """A configuration whose every step is a Gram evaluation (here MCGRA_NO_LOWRANK=1 at n = 1100; GAT / SAGE victims and CKA
    likewise): the monitoring forward packs both orientations of Xc and forks the step's first product Kx = Xc Xc^T as soon as
    adj_norm stands (round 6), and the next step adopts both with the forward.  Same launches on the same data as a step that
    does it itself (MCGRA_GRAM_KX_EARLY=0): the same bits -- gradient and state over three steps; a product nobody takes (a
    finalize, a new start, a second monitor call behind the monitor) is dropped and the run goes on with the same bits."""
    import torch
    z = _synthetic_case(1100, 11, (16, 16), 4, seed=5)
    monkeypatch.setenv("MCGRA_NO_LOWRANK", "1")
    outs = []
    for early in ("1", "0"):
        monkeypatch.setenv("MCGRA_GRAM_KX_EARLY", early)
        eng = H.engine_from(pkg, z)
        gs = []
        for t in range(3):
            eng.step(); eng.monitor()
            if t == 1:
                eng.monitor()                                   # a second forward: the first one's product is dropped, a new one forked
            gs.append(eng.buffer("G_sym").clone())
        lab = z["labels"]
        la = (lab[:, None] == lab[None, :]).astype(np.float32)
        fin = eng.finalize(0, eng.buffer("HA"), eng.buffer("YA"), la).clone()      # behind a monitor call: its product is dropped
        a = eng.get_adj_changes().clone()
        eng.set_adj_changes(a)                                  # a new start (drops the adopted forward)
        eng.step(); eng.monitor(); eng.step()
        outs.append((gs, fin, eng.buffer("M").clone()))
        assert eng.gram_split_steps() == 5 and eng.fused_steps() == 0
        eng.close()
    for t in range(3):
        assert torch.equal(outs[0][0][t], outs[1][0][t]), t
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


def test_kde_columns_follow_the_reach_of_feature_adj(pkg, torch_):
    """The N x N KDE terms are evaluated on the columns whose bins the operands' values can reach in float32: 8 for the operands
    the reference builds (all <= 1).  A caller-supplied feature_adj with larger entries moves that bound -- set_graph measures
    max |feature_adj| and widens the column count (here entries of 9.0 / 8.7 in column 9 and 10.9 in column 11: 16 columns) --
    and values whose bins lie beyond the 32-column tables are refused by name instead of being dropped silently."""
    z = H.synthetic_case(300, 11, (16, 16), 4, seed=3, measure="KDE")
    g0 = None
    for wide in (False, True):
        zz = dict(z)
        if wide:
            F = z["feature_adj"].copy()
            F[5, 9], F[17, 9], F[40, 11], F[41, 11] = 9.0, 8.7, 10.9, 11.2
            zz["feature_adj"] = F
        eng, o = H.engine_from(pkg, zz), H.oracle_from(zz)
        sc = eng.step(want_scalars=True)
        o.step()
        g, g_or = eng.buffer("G_sym").cpu().numpy(), o.last["G_sym"]
        assert np.abs(g - g_or).max() <= 1e-4 * np.abs(g_or).max(), (wide, np.abs(g - g_or).max(), np.abs(g_or).max())
        assert sc["c1"] == pytest.approx(o.last["terms"]["c1"], rel=1e-4, abs=2e-8 * z["weight_param"][0] * 1e5), wide
        if not wide:
            g0, c10 = g_or, o.last["terms"]["c1"]
        else:      # the wide entries do move the term: the test sees the columns beyond the eighth
            assert abs(o.last["terms"]["c1"] - c10) > 1e-3 * abs(c10) or np.abs(g_or - g0).max() > 1e-3 * np.abs(g0).max()
        eng.close()
    F = z["feature_adj"].copy()
    F[0, 0] = 40.0
    with pytest.raises(pkg._lib.McgraNotSupported, match="max .feature_adj. = 40"):
        H.engine_from(pkg, dict(z, feature_adj=F))


@pytest.mark.parametrize("wp", [(0.01, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000), (0.01, 0, 0, 0, 0, 10, 10, 0, 10, 0)], ids=["all", "c1only"])
def test_gram_products_beside_the_step_are_bit_identical(pkg, monkeypatch, wp):
    """Gram-evaluation steps (here MCGRA_NO_LOWRANK=1 at n = 1100) run their four N x N x N products on the side stream -- Kx
    beside the forward chains and the decode, Ky beside the small-operand terms and the victim chain's backward, G_A1 += LX Yc
    beside the rank-k update of G_adjn, G_adjn += LY Xc beside the decode backward -- with the combine of the centred Grams and
    the packs of its two results in one pass.  Same launches in the same order on ONE stream (MCGRA_GRAM_OVERLAP=0): the same
    bits, gradient and state, over three steps with the monitoring forward in between."""
    z = _synthetic_case(1100, 11, (16, 16), 4, seed=5, weight_param=wp)
    monkeypatch.setenv("MCGRA_NO_LOWRANK", "1")
    outs = []
    for ovl in ("1", "0"):
        monkeypatch.setenv("MCGRA_GRAM_OVERLAP", ovl)
        eng = H.engine_from(pkg, z)
        gs = []
        for t in range(3):
            eng.step(); eng.monitor()
            gs.append(eng.buffer("G_sym").clone())
        outs.append((gs, eng.buffer("M").clone()))
        assert eng.gram_split_steps() == 3 and eng.fused_steps() == 0
        eng.close()
    import torch
    for t in range(3):
        assert torch.equal(outs[0][0][t], outs[1][0][t]), t
    assert torch.equal(outs[0][1], outs[1][1])


# ---- the fused MSELoss step (round 5): calc = MSELoss evaluated from M, feature_adj, r and Zn -- no N x N intermediate ------
# ---- and the fused KL step (round 6): calc = calc_kl (:197-198, :483-487), the same data flow + per-row softmax statistics
@pytest.mark.parametrize("measure", ["MSELoss", "KL"])
@pytest.mark.parametrize("n,widths,wp", [
    (1100, (16, 16), None), (300, (16, 16), NXN_ONLY), (515, (8, 8), (0, 0.01, 0, 0, 0, 10, 10, 0, 10, 1000)),
    (1030, (16, 16, 16), None), (700, (16, 32), (0.01, 0, 0, 0, 0, 0, 10, 0, 10, 0))])
def test_fused_mse_step_matches_general_path_and_oracle(pkg, n, widths, wp, measure, monkeypatch):
    """measure = MSELoss through attack_fused.hip (adj_norm, modified_adj1 and the gradients w.r.t. them are never stored: the
    decode carries d / d modified_adj1, the tail's first pass d / d adj_norm with feature_adj in the place of the HSIC product
    and S = Zn Zn^T as a third rank-k group) against the general step (MCGRA_NO_FUSED_LR=1) and the oracle, teacher-forced,
    with the monitoring forward adopted in between: mirrored gradient, every loss term, the updated adjacency.
    measure = KL: the same through the fused KL step (softmax(feature_adj) in feature_adj's place, the row statistics of
    adj_norm and modified_adj1 from one more per-pair pass, an asymmetric per-pair term symmetrised in the decode and the tail)."""
    kw = {} if wp is None else {"weight_param": wp}
    z = _synthetic_case(n, 11, widths, 4, seed=n, measure=measure, **kw)
    fused = H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_NO_FUSED_LR", "1")
    gen = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_NO_FUSED_LR")
    orc = H.oracle_from(z)
    for t in range(3):
        a, b = fused.step(want_scalars=True), gen.step(want_scalars=True)
        orc.step()
        gf, gg = fused.buffer("G_sym").cpu().numpy(), gen.buffer("G_sym").cpu().numpy()
        gr = orc.last["G_sym"]
        scale = np.abs(gr).max()
        assert np.abs(gf - gg).max() <= 3e-5 * scale, (t, np.abs(gf - gg).max() / scale)
        assert np.abs(gf - gr).max() <= 1e-4 * scale, (t, np.abs(gf - gr).max() / scale)
        # (KL: the VALUE of calc_kl on the N x N operands is sum p (log p - log_softmax(adj_norm)) / n ~ 1e-2 -- a difference of
        # row log-sum-exps of ~ log n that agree to three digits, so the float32 rounding of a row's logsumexp (6e-8 x log n) shows
        # at 4e-5 of the value; the two paths sum the rows' exponentials differently -- fp64 partials here, fp32 block sums there)
        vtol = 2e-4 if measure == "KL" else 2e-5
        for k in ("loss", "c1", "c2", "c6", "c7", "c9", "c10", "nll", "origin_loss"):
            assert a[k] == pytest.approx(b[k], rel=vtol if k in ("loss", "c1", "c2") else 2e-5, abs=1e-6 * max(1.0, abs(b["loss"]))), (t, k, a[k], b[k])
        assert abs(a["loss"] - orc.last["loss"]) <= 2e-4 * abs(orc.last["loss"]) + 1e-5
        if measure == "KL":
            for k in ("c1", "c2"):
                assert a[k] == pytest.approx(orc.last["terms"].get(k, 0.0), rel=2e-4, abs=1e-6 * max(1.0, abs(b["loss"]))), (t, k)
        Mf = fused.buffer("M")
        assert bool((Mf == Mf.T).all()), "the learnable adjacency must stay symmetric bit for bit"
        gsf = fused.buffer("G_sym")
        assert bool((gsf == gsf.T).all())
        ma, mb = fused.get_adj_changes(), gen.get_adj_changes()
        lr = float(z["lr"])
        assert float(((ma - mb).abs() > 0.05 * lr).float().mean()) < 2e-3       # Adam: +-lr on noise-level gradients
        la, _ = fused.monitor(); lb, _ = gen.monitor()
        assert float((la - lb).abs().max()) < 1e-4
        nxt = O.pack_tril(orc.M)
        fused.set_adj_changes(nxt); gen.set_adj_changes(nxt)                   # teacher forcing (drops the adopted forward)
    assert fused.fused_steps() == 3 and gen.fused_steps() == 0
    assert fused.path_stats() == gen.path_stats() == {"lowrank_steps": 0, "general_steps": 0}


@pytest.mark.parametrize("measure", ["MSELoss", "KL"])
def test_fused_elementwise_steps_mask_the_pairs_the_reference_masks(pkg, measure, monkeypatch):
    """The fused MSELoss / KL steps have no data-dependent fallback: their decode backward masks exactly the pairs relu'(0) = 0 masks
    in the reference (S_ij <= 0), dead embedding rows included (zn_i = 0: modified_adj1's row is zero, its softmax uniform).  Weights
    whose embedding bias kills about half of em (tests/helpers.py:masked_weights: thousands of masked pairs, a third of the rows
    dead) -- fused against the general step, every step fused.  (Not against the oracle here: the construction puts the median of
    every embedding column AT the ReLU threshold, so rows survive on one coordinate of rounding-noise size whose direction --
    hence modified_adj1_ij = 1 or 0 against every other such row -- is decided by the last bit of the forward's summation order:
    oracle and engine, 1.7e-5 apart in em, differ by 1.0 on entries of 402 rows, c7 by 0.4 %.  The general step is what the
    reference's masked fixtures pin; the two engine steps agree to 3e-7 / 1.2e-6 of the gradient's largest magnitude.)"""
    z = _synthetic_case(600, 11, (16, 16), 4, seed=9, measure=measure)
    w = H.masked_weights(z)
    engs = []
    for nofuse in (False, True):
        if nofuse:
            monkeypatch.setenv("MCGRA_NO_FUSED_LR", "1")
        e = H.engine_from(pkg, z)
        e.set_model(w.W, w.b, w.Wlin, w.blin, w.Ws)
        e.set_graph(z["features"], z["adj"], None, z["feature_adj"], z["labels"], z["idx_attack"])
        e.set_adj_changes(H.a0_of(z))
        engs.append(e)
    monkeypatch.delenv("MCGRA_NO_FUSED_LR")
    fused, gen = engs
    probe = O.PGDAttackOracle(w, z["features"], z["adj"], np.zeros_like(z["adj"]), z["feature_adj"], z["labels"], z["idx_attack"], H.cfg_from(z))
    probe.set_adj_changes(H.a0_of(z))
    probe.step()
    em = probe.last["em"]
    assert (em == 0).mean() > 0.5 and (np.abs(em).sum(1) == 0).sum() > 50      # the bias does kill most of em, and whole rows
    for t in range(3):
        a, b = fused.step(want_scalars=True), gen.step(want_scalars=True)
        gf, gg = fused.buffer("G_sym").cpu().numpy(), gen.buffer("G_sym").cpu().numpy()
        scale = np.abs(gg).max()
        assert np.abs(gf - gg).max() <= 3e-5 * scale, (t, np.abs(gf - gg).max() / scale)
        for k in ("loss", "c1", "c2", "c6", "c7", "c9", "c10"):
            assert a[k] == pytest.approx(b[k], rel=2e-4 if measure == "KL" else 2e-5, abs=1e-6 * max(1.0, abs(b["loss"]))), (t, k, a[k], b[k])
        fused.set_adj_changes(gen.get_adj_changes())
    assert fused.fused_steps() == 3 and gen.fused_steps() == 0


@pytest.mark.parametrize("measure", ["MSELoss", "KL"])
def test_fused_mse_free_run_and_finalize(pkg, measure, monkeypatch):
    """Free-running fused MSELoss (KL) steps with and without monitor calls in between give the same bits, and the post-loop
    ensemble after a fused loop equals the one after the same loop on the general path (em_last stands in for
    embedding(features, adj_norm))."""
    import torch
    z = _synthetic_case(1100, 11, (16, 16), 4, seed=5, measure=measure)
    e1, e2 = H.engine_from(pkg, z), H.engine_from(pkg, z)
    monkeypatch.setenv("MCGRA_NO_FUSED_LR", "1")
    e3 = H.engine_from(pkg, z)
    monkeypatch.delenv("MCGRA_NO_FUSED_LR")
    for t in range(3):
        e1.step(); e1.monitor()
        e2.step()
        e3.step(); e3.monitor()
    assert torch.equal(e1.get_adj_changes(), e2.get_adj_changes())
    lr = float(z["lr"])
    assert float(((e1.get_adj_changes() - e3.get_adj_changes()).abs() > 0.05 * lr).float().mean()) < 2e-3
    lab = z["labels"]
    la = (lab[:, None] == lab[None, :]).astype(np.float32)
    f1 = e1.finalize(0, e1.buffer("HA"), e1.buffer("YA"), la)
    f3 = e3.finalize(0, e3.buffer("HA"), e3.buffer("YA"), la)
    assert float((f1 - f3).abs().max()) < 2e-3 * float(f3.abs().max())
    assert abs(O.metric_pool(z["adj"], f1.cpu().numpy(), z["idx_attack"]) - O.metric_pool(z["adj"], f3.cpu().numpy(), z["idx_attack"])) < 1e-4
    assert e1.fused_steps() == 3 and e3.fused_steps() == 0


@pytest.mark.parametrize("measure", ["MSELoss", "KL"])
@pytest.mark.parametrize("n,widths,world,wp", [(1100, (16, 16), 2, None), (1100, (16, 16), 3, (0.01, 1.0, 0, 0, 0, 10, 10, 0, 10, 1000)),
                                               (600, (16, 16, 16), 4, None)])
def test_sharded_mse_ranks_match_monolithic_step(pkg, n, widths, world, wp, measure):
    """The fused MSELoss (KL) step as `world` row-block ranks in lockstep: no N x N exchange at all (all-gathers of node arrays with
    the partial scalars in their lane only; KL: one more gather, of the rows' softmax statistics) -- the union of the ranks' rows
    equals the monolithic fused step, mirrored entries bit for bit across the ranks, loss terms identical on every rank."""
    from mc_gra_amd import sharded as S
    import torch
    kw = {} if wp is None else {"weight_param": wp}
    z = _synthetic_case(n, 11, widths, 4, seed=n, measure=measure, **kw)
    mono = H.engine_from(pkg, z)
    plans, bks = _shard_engines(pkg, z, world, joint=world == 3)
    lr = float(z["lr"])
    nex = 0
    for t in range(3):
        a = mono.step(want_scalars=True); mono.monitor()
        sc = S.run_lockstep(bks, S.SHARD_STEP, want_scalars=True)
        S.run_lockstep(bks, S.SHARD_MONITOR)
        # the mirrored gradient on every rank's own rows: a rank holds only its own rows of M current, and the MSELoss part of the
        # decode forms adj_norm_ij per pair -- from THOSE rows (wp with w2 = 1 makes that part a tenth of the gradient).  Second
        # step: the two runs' states differ on the entries Adam moved by +-lr on a noise-level gradient's sign (lr = 0.01 here), and
        # the gradient is not continuous in the state everywhere (Info_entropy's clamp at 1e-4, the decode's relu): 99.99 % of the
        # entries within 3e-4, every entry within 5e-3 (calc_kl, 3 layers, 4 ranks: ONE mirrored pair at 1.25e-3; with the ranks'
        # state forced to the monolithic engine's the second step agrees to 5e-9: scripts/diag_r6.py 2)
        if t <= 1:
            gm = mono.buffer("G_sym")
            for b, pl in zip(bks, plans):
                if pl.has_rows:
                    gr = b.eng.buffer("G_sym")[pl.row_begin:pl.row_end]
                    d = (gr - gm[pl.row_begin:pl.row_end]).abs()
                    gmax = float(gm.abs().max())
                    if t == 0:
                        assert float(d.max()) <= 3e-6 * gmax, (t, pl.rank)
                    else:
                        assert float((d > 3e-4 * gmax).float().mean()) <= 1e-4 and float(d.max()) <= 5e-3 * gmax, (t, pl.rank, float(d.max()) / gmax)
        rows = _gather_rows(bks)
        M = mono.buffer("M")
        assert rows.shape == M.shape
        assert float(((rows - M).abs() > 0.05 * lr).float().mean()) < 2e-3, t
        assert float((rows - rows.T).abs().max()) == 0.0, "ranks must agree on mirrored entries bit for bit"
        for k in ("loss", "c1", "c2", "c6", "c7", "c9", "c10", "nll", "clamp_sum"):
            # (KL: the value of calc_kl is a difference of row log-sum-exps that agree to three digits -- see
            # test_fused_mse_step_matches_general_path_and_oracle -- and the ranks cut the rows' sums into other slices)
            tol = 2e-4 if measure == "KL" and k in ("loss", "c1", "c2") else 3e-5
            for b in sc:
                assert b[k] == pytest.approx(a[k], rel=tol, abs=1e-6 * max(1.0, abs(a["loss"]))), (t, k, b[k], a[k])
            assert all(b[k] == sc[0][k] for b in sc), "scalars are identical on every rank"
    assert all(b.eng.fused_steps() == 3 for b in bks) and mono.fused_steps() == 3
    assert all(b.eng.cut_product_steps() == 0 for b in bks)
    # (the collectives per step are counted across a process boundary: tests/test_gpu_multiproc.py -- 6 for MSELoss, 7 for KL)
