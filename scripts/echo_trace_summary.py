"""Kernel table of ONE rank's step from a rocprofv3 --kernel-trace of `scripts/shard_emulate.py --echo` (one world): the last
`steps` steps (a step = SHARD_STEP + SHARD_MONITOR), per step -- the product, what runs beside it, what follows the join.
    python scripts/echo_trace_summary.py <kernel_trace.csv> <steps>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    s = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+)(<[^(]*>)?\(", s)
    return (m.group(1) + (m.group(2) or ""))[:56] if m else s[:56]
# a step ends with its Adam pass (k_tail_adam); the timed steps are the last `steps` of them
adam = [i for i, r in enumerate(rows) if "k_tail_adam" in r["Kernel_Name"]]
a, b = adam[-steps - 1] + 1, adam[-1] + 1
t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b - 1]["End_Timestamp"])
cnt, dur = collections.Counter(), collections.Counter()
prod = 0
for r in rows[a:b]:
    k = nm(r); d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[k] += 1; dur[k] += d
    if "split2_m16" in k or "k_split3_reduce" in k:
        prod += d
tot = sum(dur.values())
print(f"{steps} steps: wall {(t1 - t0) / 1e3 / steps:.0f} us per step; kernel time {tot / 1e3 / steps:.0f} us ({prod / 1e3 / steps:.0f} us the N x N x N "
      f"product, {(tot - prod) / 1e3 / steps:.0f} us everything else), {sum(cnt.values()) / steps:.0f} launches per step")
print(f"{'us/step':>10} {'launches':>8}  kernel")
for k, v in dur.most_common(45):
    print(f"{v / 1e3 / steps:10.1f} {cnt[k] / steps:8.2f}  {k}")
if len(sys.argv) > 3 and sys.argv[3] == "--timeline":
    # launch list of the last step, per queue: start relative to the step, duration, gap to the previous launch of the queue
    a2, b2 = adam[-2] + 1, adam[-1] + 1
    t0 = int(rows[a2]["Start_Timestamp"])
    last = {}
    print(f"--- last step, launch by launch ({b2 - a2} launches, {(int(rows[b2 - 1]['End_Timestamp']) - t0) / 1e3:.0f} us)")
    for r in rows[a2:b2]:
        s_, e_, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
        gap = (s_ - last[q]) / 1e3 if q in last else 0.0
        last[q] = e_
        g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        print(f"{(s_ - t0) / 1e3:8.1f} {(e_ - s_) / 1e3:7.1f} gap {gap:6.1f} q{q} g{g:5d}x{r['Grid_Size_Y']:>3} {nm(r)}")
