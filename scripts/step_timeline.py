"""Timeline of ONE step from a rocprofv3 --kernel-trace CSV of bench.py (overlapped streams, no counters): every launch between
two launches of the N x N x N product, with the gap to the previous launch on the same stream and the kernels that are on the
step's exposed path marked.  python scripts/step_timeline.py <kernel_trace.csv> [--step K]"""
import argparse, csv, re
ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--step", type=int, default=-3)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    s = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+)(<[^(]*>)?\(", s)
    return (m.group(1) + (m.group(2) or ""))[:50] if m else s[:50]
gx = lambda r: int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
main = [i for i, r in enumerate(rows) if "split2_m16" in r["Kernel_Name"] and gx(r) >= 1000]
i0, i1 = main[a.step], main[a.step + 1]
t0 = int(rows[i0]["Start_Timestamp"])
pend = int(rows[i0]["End_Timestamp"])
last = {}
print(f"product: {(pend - t0) / 1e3:.1f} us; step (product start to product start): {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us")
for r in rows[i0:i1 + 1]:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    if e < pend - 50000 and gx(r) < 1000 and (e - s) < 30000:
        continue                                   # (short launches deep inside the product's shadow: not printed)
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {gap:7.1f} q{q} g{gx(r):6d}x{r['Grid_Size_Y']:>3} {nm(r)}")
