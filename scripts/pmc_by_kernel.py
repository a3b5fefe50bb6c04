"""Memory-side bytes per kernel and step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace --output-format csv):
bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, summed over a kernel's launches and divided by the steps of the pass.
    python3 scripts/pmc_by_kernel.py <fetch dir> <write dir> <steps in the pass> [rows] [--by-grid]
(--by-grid: launches of one kernel with different grids apart, e.g. the N x N fills of the allocations from the per-step n-vector fills)"""
import csv, glob, os, re, sys
BY_GRID = "--by-grid" in sys.argv
if BY_GRID: sys.argv.remove("--by-grid")
fd, wd, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]); rows = int(sys.argv[4]) if len(sys.argv) > 4 else 16
def load(d, ctr):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr: continue
            s = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            m = re.search(r"([\w:]+)(<[^(]*>)?\(", s); k = ((m.group(1) + (m.group(2) or "")) if m else s)[:70]
            if BY_GRID: k += f" grid={r['Grid_Size']}"
            e = out.setdefault(k, [0, 0.0]); e[0] += 1; e[1] += float(r["Counter_Value"])
    return out
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
tot = 0.0; lines = []
for k in sorted(set(F) | set(W)):
    n = max(F.get(k, [0, 0])[0], W.get(k, [0, 0])[0])
    b = (2.0 * F.get(k, [0, 0.0])[1] + W.get(k, [0, 0.0])[1]) * 1024.0 / steps
    if n >= steps: tot += b; lines.append((b, n / steps, k))
print(f"memory-side bytes per step, kernels launched at least once per step: {tot / 1e9:.3f} GB")
for b, n, k in sorted(lines, reverse=True)[:rows]:
    print(f"{b / 1e9:8.3f} GB  {n:5.1f} launches/step  {k}")
