"""How much host time a step leaves unused: the bench loop of a workload with a busy-wait of D microseconds added per step on the
host.  If ms/step does not move, the step is bound by the GPU's chain of launches and the host has at least D of slack.
    python scripts/host_slack.py [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import mcgra_loader
pkg = mcgra_loader.load()
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "cora-shape-hsic"
dev = torch.device("cuda:0")
eng, inp, adj = bench.build_engine(pkg, torch, dev, wl, 0)
def run(delay_us, steps=400):
    for _ in range(40):
        eng.step(); eng.monitor()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(); eng.monitor()
        if delay_us:
            t = time.perf_counter() + delay_us * 1e-6
            while time.perf_counter() < t:
                pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for d in (0, 25, 50, 100, 150, 200, 0):
    print(f"{wl}: +{d:3d} us of host time per step -> {run(d):.4f} ms/step")
