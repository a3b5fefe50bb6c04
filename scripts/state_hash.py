"""sha256 of the learnable adjacency after a few steps of a bench workload (A/B of two builds for bit identity:
MCGRA_LIB_PATH=<other .so> python scripts/state_hash.py <workload> [steps])."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mcgra_loader
pkg = mcgra_loader.load()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "synthetic-4k-hsic"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
eng, inp, adj = bench.build_engine(pkg, torch, dev, wl, 0)
for _ in range(steps):
    eng.step(); eng.monitor()
M = eng.buffer("M").cpu().numpy()
print(wl, steps, hashlib.sha256(M.tobytes()).hexdigest()[:16], "fused", eng.fused_steps(), os.environ.get("MCGRA_LIB_PATH", "tree"))
