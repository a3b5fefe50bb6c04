"""GPU box: `steps` engine steps (+ monitoring forward) of ONE README-line fixture, for a rocprofv3 kernel table of that line:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x -- python3 scripts/readme_line_steps.py readme_cora_kde_Y 40"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mcgra_loader
pkg = mcgra_loader.load()
from tests import helpers as H

name, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
z = H.load_readme(name)
wp = [float(x) for x in z["weight_param"]]
if not (z["feature_adj"].max() != z["feature_adj"].min()):
    wp[0] = 0.0
eng = H.engine_from(pkg, z, weight_param=tuple(wp))
for _ in range(5):
    eng.step(); eng.monitor()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.step(); eng.monitor()
torch.cuda.synchronize()
print(name, "n", len(z["labels"]), str(z["measure"]), "ms/step", round(1e3 * (time.perf_counter() - t0) / steps, 4), "fused", eng.fused_steps(),
      eng.path_stats(), "state", hashlib.sha256(eng.buffer("M").cpu().numpy().tobytes()).hexdigest()[:16])
