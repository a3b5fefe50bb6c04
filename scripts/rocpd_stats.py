#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 --kernel-trace result database (rocpd sqlite): the same table as
`--stats` prints, plus (with --steps K --skip S) the ordered launch list of one timed step.
    python scripts/rocpd_stats.py gpurun_out/<dir>/ks_results.db [--csv out.csv] [--timeline N]"""
import argparse
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"mcgra::", "", name)
    m = re.match(r"([\w:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv")
    ap.add_argument("--timeline", type=int, default=0, help="print the last N launches in order")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    rows = db.execute("select * from kernels order by start").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    stats = {}
    for r in rows:
        nm = short(r[ix["name"]])
        d = r[ix["end"]] - r[ix["start"]]
        s = stats.setdefault(nm, [0, 0, 1 << 62, 0])
        s[0] += 1; s[1] += d; s[2] = min(s[2], d); s[3] = max(s[3], d)
    tot = sum(s[1] for s in stats.values())
    out = sorted(stats.items(), key=lambda kv: -kv[1][1])
    print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for nm, (c, t, mn, mx) in out[:45]:
        print(f"{nm:90s} {c:6d} {t / 1e6:10.3f} {t / c / 1e3:10.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f} {100.0 * t / tot:6.2f}")
    if a.csv:
        with open(a.csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for nm, (c, t, mn, mx) in out:
                w.writerow([nm, c, t, t / c, 100.0 * t / tot, mn, mx])
    if a.timeline:
        t0 = rows[-a.timeline][ix["start"]]
        for r in rows[-a.timeline:]:
            print(f"{(r[ix['start']] - t0) / 1e3:10.1f} us  +{(r[ix['end']] - r[ix['start']]) / 1e3:9.1f} us  {short(r[ix['name']])}")


if __name__ == "__main__":
    main()
