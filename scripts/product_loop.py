"""The N x N x N product of the bench workload by itself for a few seconds: split2_m16_kernel back to back on the operand
planes of a bench step (mcgra_attack_product_replay), nothing else on the chip.  For scripts/power_trace.py.
    python scripts/product_loop.py [--seconds 5] [--workload synthetic-10k-hsic]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=5.0)
ap.add_argument("--workload", default="synthetic-10k-hsic")
a = ap.parse_args()
import torch
import mcgra_loader
pkg = mcgra_loader.load()
import bench
dev = torch.device("cuda:0")
eng, inp, adj_dev = bench.build_engine(pkg, torch, dev, a.workload, 0)
eng.step(); eng.monitor(); eng.step()
eng.product_replay(5)
t0 = time.time(); calls = 0; ms = []
while time.time() - t0 < a.seconds:
    ms.append(eng.product_replay(50)); calls += 50
dt = time.time() - t0
n = bench.WORKLOADS[a.workload][0]
avg = sum(ms) / len(ms)
print(json.dumps({"what": "product_loop", "n": n, "calls": calls, "ms_per_call": avg, "wall_ms_per_call": 1e3 * dt / calls,
                  "issued_pflops": 6.0 * n ** 3 / (avg * 1e-3) / 1e15, "ms_min_max": [min(ms), max(ms)]}))
