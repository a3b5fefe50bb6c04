"""Print the top kernels of a rocprofv3 --stats run: scripts/kstats.py gpurun_out/<dir> [steps]"""
import csv, glob, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    nm = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("mcgra::", "").split("(")[0][:84]
    print(f'{nm:86s} {int(r["Calls"]):5d} calls {float(r["TotalDurationNs"])/1e6/steps:8.3f} ms/step  avg {float(r["AverageNs"])/1e3:9.1f} us {float(r["Percentage"]):6.2f}%')
