"""Per rank-step kernel table of a rocprofv3 --kernel-trace of scripts/shard_emulate.py (one world): which kernels a row-block
rank spends its step in.  python scripts/em_trace_summary.py <kernel_trace.csv> <world> [steps]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
world = int(sys.argv[2]); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    s = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+)(<[^(]*>)?\(", s)
    return (m.group(1) + (m.group(2) or ""))[:56] if m else s[:56]
idx = [i for i, r in enumerate(rows) if "split2_m16" in r["Kernel_Name"] and int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) >= 100]
per = len(idx) // (steps + 3)                      # main launches per step over all ranks (2 warm-up + steps + 1 recorded)
a, b = idx[2 * per], idx[(2 + steps) * per]
t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
rs = world * steps
cnt, dur = collections.Counter(), collections.Counter()
for r in rows[a:b]:
    cnt[nm(r)] += 1; dur[nm(r)] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"world {world}: {steps} steps, wall {(t1 - t0) / 1e6:.2f} ms = {(t1 - t0) / 1e3 / rs:.0f} us per rank and step; "
      f"kernel time {sum(dur.values()) / 1e3 / rs:.0f} us, {sum(cnt.values()) / rs:.0f} launches per rank and step")
print(f"{'us/rank-step':>12} {'launches':>8}  kernel")
for k, v in dur.most_common(40):
    print(f"{v / 1e3 / rs:12.1f} {cnt[k] / rs:8.2f}  {k}")
