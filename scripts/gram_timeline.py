"""Launch list of ONE Gram-evaluation step (MCGRA_NO_LOWRANK=1) from a rocprofv3 --kernel-trace CSV of bench.py: every launch
between the first product launch of a step and the first product launch of the next, with its start relative to the step, its
duration and the gap to the previous launch of its queue; then the step's time by kernel.
    python scripts/gram_timeline.py <kernel_trace.csv> [--step K]"""
import argparse, collections, csv, re
ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--step", type=int, default=-3)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.csv)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    s = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+)(<[^(]*>)?\(", s)
    return (m.group(1) + (m.group(2) or ""))[:60] if m else s[:60]
gx = lambda r: int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
# a step starts at its k_prep / prep_from_partials launch (forward_common): the first launch after the Adam pass
adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
i0, i1 = adam[a.step] + 1, adam[a.step + 1] + 1
t0 = int(rows[i0]["Start_Timestamp"])
print(f"step: {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us, {i1 - i0} launches")
last, tot, cnt = {}, collections.Counter(), collections.Counter()
for r in rows[i0:i1]:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    tot[nm(r)] += e - s; cnt[nm(r)] += 1
    if (e - s) >= 20000:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {gap:7.1f} q{q} g{gx(r):6d}x{r['Grid_Size_Y']:>3} {nm(r)}")
print("--- by kernel")
for k, v in tot.most_common(40):
    print(f"{k:<62s} {cnt[k]:3d} x {v / cnt[k] / 1e3:9.1f} us = {v / 1e6:8.3f} ms")
print(f"kernel time {sum(tot.values()) / 1e6:.3f} ms")
