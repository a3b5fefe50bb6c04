"""Launch list of ONE general-path step (+ its monitoring forward) from a rocprofv3 --kernel-trace CSV: every launch between two
Adam passes (k_adam_sym, or the fused tail k_rankk_apply_adam) with its start relative to the step, duration, gap to the previous launch of its queue, blocks.
    python3 scripts/general_step_timeline.py <kernel_trace.csv> [step index from the end, default 3] [delimiting kernel, e.g. k_tail_adam for a fused step]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
def nm(r):
    s = r["Kernel_Name"].replace("(anonymous namespace)::", ""); m = re.search(r"(\w+)(<[^(]*>)?\(", s)
    return (m.group(1) + (m.group(2) or ""))[:50] if m else s[:50]
delim = (sys.argv[3],) if len(sys.argv) > 3 else ("k_adam_sym", "k_rankk_apply_adam")
adam = [i for i, r in enumerate(rows) if any(d in r["Kernel_Name"] for d in delim)]
i0, i1 = adam[-k] + 1, adam[-k + 1] + 1
t0 = int(rows[i0]["Start_Timestamp"]); last = {}
print(f"step {(int(rows[i1 - 1]['End_Timestamp']) - t0) / 1e3:.1f} us, {i1 - i0} launches   [start us, duration us, gap to the queue's previous launch, queue, blocks, kernel]")
for r in rows[i0:i1]:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    gap = (s - last[q]) / 1e3 if q in last else 0.0; last[q] = e
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap {gap:6.1f} q{q} g{int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):5d} {nm(r)}")
