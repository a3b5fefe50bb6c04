"""GPU box: fused MSELoss step against the general step on a README fixture, step by step (diagnostic)."""
import os, sys, argparse
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MCGRA_KEEP_GSYM"] = "1"; os.environ["MCGRA_AB"] = "1"
import torch
import mcgra_loader
pkg = mcgra_loader.load()
from tests import helpers as H
from oracle import mcgra_oracle as O
from mc_gra_amd.topology_attack import _decode_mode
name = sys.argv[1] if len(sys.argv) > 1 else "readme_usair_mse_y"
z = H.load_readme(name)
wp = [float(x) for x in z["weight_param"]]
if not (z["feature_adj"].max() != z["feature_adj"].min()):
    wp[0] = 0.0
print(name, "wp", wp, "wsup", float(z["weight_sup"]), "lr", float(z["lr"]), "n", len(z["labels"]))
fused = H.engine_from(pkg, z, weight_param=tuple(wp))
os.environ["MCGRA_NO_FUSED_LR"] = "1"
gen = H.engine_from(pkg, z, weight_param=tuple(wp))
del os.environ["MCGRA_NO_FUSED_LR"]
for t in range(int(z["epochs"])):
    fused.step(); gen.step()
    gf, gg = fused.buffer("G_sym"), gen.buffer("G_sym")
    mf, mg = fused.buffer("M"), gen.buffer("M")
    if len(sys.argv) > 2:
        for nm in ("G_adjn", "G_em", "GPu", "GPv", "GZn", "Zn", "gd", "d", "r", "em", "sm2"):
            b = fused.buffer(nm)
            print("   ", nm, "nan", int(torch.isnan(b).sum()), "inf", int(torch.isinf(b).sum()), "absmax", float(torch.nan_to_num(b, nan=0.0, posinf=0.0, neginf=0.0).abs().max()))
    print(t, "G max", float(gg.abs().max()), "dG", float((gf - gg).abs().max()), "dM", float((mf - mg).abs().max()), "M max", float(mg.max()),
          "nan", bool(torch.isnan(mf).any()), "fused", fused.fused_steps())
use = [bool(u) for u in z["use"]]
lab = z["labels"]
la = (lab[:, None] == lab[None, :]).astype(np.float32)
args = argparse.Namespace(dataset=str(z["dataset"]), useH_A=use[0], useY_A=use[1], useY=use[2])
ff = fused.finalize(_decode_mode(args), z["H_A2"] if use[0] else None, z["Y_A"] if use[1] else None, la if use[2] else None)
fg = gen.finalize(_decode_mode(args), z["H_A2"] if use[0] else None, z["Y_A"] if use[1] else None, la if use[2] else None)
sp = z["sample_pos"]
print("final diff fused-gen", float((ff - fg).abs().max()), "gen-ref", np.abs(fg.cpu().numpy()[sp[:, 0], sp[:, 1]] - z["final_sample"]).max(),
      "fused-ref", np.abs(ff.cpu().numpy()[sp[:, 0], sp[:, 1]] - z["final_sample"]).max())
eml = fused.buffer("em")
print("em (fused, Hu chain) max", float(eml.abs().max()))
