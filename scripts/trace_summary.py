"""Per-kernel totals of a rocprofv3 --kernel-trace CSV over a window of steps.
    python scripts/trace_summary.py <kernel_trace.csv> [--from-product K] [--per N] [--top T]
--from-product K: the window starts at the K-th launch of the N x N x N product (split2_m16_kernel main grid); --per N divides
the totals by N (steps, or rank-steps of a lockstep emulation)."""
import argparse, collections, csv, re
ap = argparse.ArgumentParser()
ap.add_argument("csv"); ap.add_argument("--from-product", type=int, default=0); ap.add_argument("--per", type=float, default=1.0)
ap.add_argument("--top", type=int, default=45)
a = ap.parse_args()
rows = []
for r in csv.DictReader(open(a.csv)):
    rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort(key=lambda x: x[1])
prods = [i for i, r in enumerate(rows) if "split2_m16" in r[0]]
# main launches only (the split-K tail launch of a step follows its main launch directly)
mains = [i for k, i in enumerate(prods) if k == 0 or prods[k - 1] != i - 1]
start = rows[mains[a.from_product]][1] if mains and a.from_product < len(mains) else rows[0][1]
sel = [r for r in rows if r[1] >= start]
tot, cnt = collections.Counter(), collections.Counter()
short = lambda s: re.sub(r"^void ", "", re.sub(r"\(.*$", "", s))[:70]
for r in sel:
    k = short(r[0]); tot[k] += r[2] - r[1]; cnt[k] += 1
T = sum(tot.values()); span = sel[-1][2] - sel[0][1]
print(f"window: {len(sel)} launches, kernel time {T/1e6:.3f} ms, span {span/1e6:.3f} ms; per unit: {len(sel)/a.per:.1f} launches, {T/1e6/a.per:.4f} ms kernel, {span/1e6/a.per:.4f} ms span")
for k, v in tot.most_common(a.top):
    print(f"{k:<72s} {cnt[k]/a.per:7.2f} x {v/cnt[k]/1e3:8.2f} us = {v/1e6/a.per:8.4f} ms")
