"""GPU box: `steps` engine steps (+ monitoring forward) of BASELINE.json configs[2]'s shape -- Citeseer (N = 3312), GAT victim
5 x 16 ELU, priors H_A + Y, measure HSIC (Gram evaluation on the split kernel) or KL -- for a rocprofv3 kernel table:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x -- python3 scripts/citeseer_gat_steps.py hsic 40"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mcgra_loader
pkg = mcgra_loader.load()
from tests import helpers as H
from oracle import mcgra_oracle as O

which, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
z = H.load_cora(f"citeseer_gat_{which}")
w = O.GCNWeights([z["W0"], z["W1"]], [z["b0"], z["b1"]], z["Wlin"], z["blin"], None, str(z["act"]), str(z["head_act"]))
n = z["adj"].shape[0]
dims = [w.W[0].shape[0]] + [x.shape[1] for x in w.W]
eng = pkg.AttackEngine(n, dims, w.Wlin.shape[0], int(z["emb_nlayer"]), str(z["measure"]), float(z["weight_sup"]),
                       tuple(float(x) for x in z["weight_param"]), float(z["lr"]), float(z["num_edges"]),
                       len(z["idx_attack"]), act="elu", head_act="elu", fin_layers=tuple(int(x) for x in z["fin_layers"]))
eng.set_model(w.W, w.b, w.Wlin, w.blin)
eng.set_graph(z["features"], z["adj"], None, H.cora_feature_adj(z["features"]), z["labels"], z["idx_attack"])
for _ in range(5):
    eng.step(); eng.monitor()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.step(); eng.monitor()
torch.cuda.synchronize()
print("citeseer_gat", which, "n", n, "dims", dims, "ms/step", round(1e3 * (time.perf_counter() - t0) / steps, 4), eng.path_stats(),
      "gram_split_steps", eng.gram_split_steps(), "state", hashlib.sha256(eng.buffer("M").cpu().numpy().tobytes()).hexdigest()[:16])
