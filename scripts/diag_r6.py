"""Round-6 debugging aid: (1) fused MSELoss / KL steps under embedding weights that mask decode pairs, against the general step and
the oracle; (2) the sharded KL step at t = 1 (n = 600, 3 layers, world 4) with the ranks' state forced to the monolithic engine's."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from oracle import mcgra_oracle as O
import mcgra_loader
pkg = mcgra_loader.load()
import torch

def part1(measure):
    z = H.synthetic_case(600, 11, (16, 16), 4, seed=9, measure=measure)
    w = H.masked_weights(z)
    engs = []
    for nofuse in (False, True):
        if nofuse: os.environ["MCGRA_NO_FUSED_LR"] = "1"
        else: os.environ.pop("MCGRA_NO_FUSED_LR", None)
        cfg = H.cfg_from(z)
        dims = [w.W[0].shape[0]] + [x.shape[1] for x in w.W]
        e = pkg.AttackEngine(600, dims, w.Wlin.shape[0], cfg.emb_nlayer, cfg.measure, cfg.weight_sup, cfg.weight_param, cfg.lr, cfg.num_edges,
                             len(z["idx_attack"]), eps=0.0, device="cuda:0", act="relu", head_act="none", has_self=False, fin_layers=cfg.fin_layers)
        e.set_model(w.W, w.b, w.Wlin, w.blin, None)
        e.set_graph(z["features"], z["adj"], None, z["feature_adj"], z["labels"], z["idx_attack"])
        e.set_adj_changes(H.a0_of(z))
        engs.append(e)
    os.environ.pop("MCGRA_NO_FUSED_LR", None)
    fused, gen = engs
    orc = O.PGDAttackOracle(w, z["features"], z["adj"], np.zeros_like(z["adj"]), z["feature_adj"], z["labels"], z["idx_attack"], H.cfg_from(z))
    orc.set_adj_changes(H.a0_of(z))
    a = fused.step(want_scalars=True); b = gen.step(want_scalars=True); orc.step()
    em = orc.last["em"]
    print(measure, "dead rows", int((np.abs(em).sum(1) == 0).sum()), "zero frac of em", float((em == 0).mean()), "fused steps", fused.fused_steps())
    gf, gg, gr = fused.buffer("G_sym").cpu().numpy(), gen.buffer("G_sym").cpu().numpy(), orc.last["G_sym"]
    sc = np.abs(gr).max()
    print("  fused-general %.2e  fused-oracle %.2e  general-oracle %.2e" % (np.abs(gf - gg).max() / sc, np.abs(gf - gr).max() / sc, np.abs(gg - gr).max() / sc))
    d = np.abs(gf - gr); i, j = np.unravel_index(d.argmax(), d.shape)
    dead = np.abs(em).sum(1) == 0
    print("  worst entry", i, j, "dead?", bool(dead[i]), bool(dead[j]), "gf", gf[i, j], "gg", gg[i, j], "gr", gr[i, j])
    rows_bad = (d.max(1) > 1e-4 * sc)
    print("  rows with error > 1e-4:", int(rows_bad.sum()), "of which dead:", int((rows_bad & dead).sum()), " dead rows total", int(dead.sum()))
    for k in ("loss", "c1", "c2", "c6", "c7", "c9", "c10"):
        print("   ", k, a[k], b[k], orc.last["terms"].get(k, orc.last["loss"] if k == "loss" else 0.0))

def part2():
    from mc_gra_amd import sharded as S
    n, widths, world = 600, (16, 16, 16), 4
    z = H.synthetic_case(n, 11, widths, 4, seed=n, measure="KL")
    mono = H.engine_from(pkg, z)
    plans = [S.RowBlockPlan(n, world, r) for r in range(world)]
    bks = [S.HipShardBackend(H.engine_from(pkg, z, plan=p), p) for p in plans]
    for force in (False, True):
        for t in range(2):
            mono.step(); mono.monitor()
            S.run_lockstep(bks, S.SHARD_STEP); S.run_lockstep(bks, S.SHARD_MONITOR)
            gm = mono.buffer("G_sym")
            for b, pl in zip(bks, plans):
                if pl.has_rows:
                    gr = b.eng.buffer("G_sym")[pl.row_begin:pl.row_end]
                    d = (gr - gm[pl.row_begin:pl.row_end]).abs()
                    i, j = np.unravel_index(int(d.argmax()), d.shape)
                    print("forced" if force else "free  ", "t", t, "rank", pl.rank, "max diff %.2e of %.2e" % (float(d.max()), float(gm.abs().max())), "at", pl.row_begin + i, j)
            if force or t == 1:
                a = mono.get_adj_changes()
                mono.set_adj_changes(a)
                for b in bks: b.eng.set_adj_changes(a)

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "1"):
        part1("MSELoss"); part1("KL")
    if what in ("all", "2"):
        part2()


def part3():
    """general engine's em / Zn / A1 against the oracle's under masked weights"""
    z = H.synthetic_case(600, 11, (16, 16), 4, seed=9, measure="MSELoss")
    w = H.masked_weights(z)
    os.environ["MCGRA_NO_FUSED_LR"] = "1"
    cfg = H.cfg_from(z)
    dims = [w.W[0].shape[0]] + [x.shape[1] for x in w.W]
    e = pkg.AttackEngine(600, dims, w.Wlin.shape[0], cfg.emb_nlayer, cfg.measure, cfg.weight_sup, cfg.weight_param, cfg.lr, cfg.num_edges,
                         len(z["idx_attack"]), eps=0.0, device="cuda:0", act="relu", head_act="none", has_self=False, fin_layers=cfg.fin_layers)
    e.set_model(w.W, w.b, w.Wlin, w.blin, None)
    e.set_graph(z["features"], z["adj"], None, z["feature_adj"], z["labels"], z["idx_attack"])
    e.set_adj_changes(H.a0_of(z))
    orc = O.PGDAttackOracle(w, z["features"], z["adj"], np.zeros_like(z["adj"]), z["feature_adj"], z["labels"], z["idx_attack"], H.cfg_from(z))
    orc.set_adj_changes(H.a0_of(z))
    e.step(); orc.step()
    for name, key in (("em", "em"), ("Zn", "Zn"), ("A1", "A1"), ("adj_norm", "adj_norm")):
        if key not in orc.last:
            print(name, "not in oracle.last:", sorted(orc.last.keys())[:40]); continue
        a = e.buffer(name).cpu().numpy(); b = orc.last[key]
        a = a[:, :b.shape[1]]
        d = np.abs(a - b)
        i, j = np.unravel_index(d.argmax(), d.shape)
        print(name, "max diff %.3e (max %.3e) at" % (d.max(), np.abs(b).max()), i, j, a[i, j], b[i, j], "rows differing > 1e-5:", int((d.max(1) > 1e-5).sum()))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "3":
    part3()
