"""Summarise a rocprofv3 --pmc SQ_* pass over bench.py for the split kernel:
    python scripts/sq_counters_summary.py gpurun_out/<dir> profiles/<out>.json
(the pass: rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -- python3 bench.py --steps 2 --warmup 1
 --no-cpu-baseline --no-split-probe)"""
import csv, glob, json, sys
from collections import defaultdict
src = sorted(glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True))[-1]
per = defaultdict(lambda: defaultdict(list))
dur = defaultdict(dict)
name = {}
for r in csv.DictReader(open(src)):
    if "split3_symm_kernel" not in r["Kernel_Name"] and "split2_m16_kernel" not in r["Kernel_Name"]:
        continue
    g = int(r["Grid_Size"])
    per[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[g][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    name[g] = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
out = {"note": "rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
               "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -- python3 bench.py --steps 2 --warmup 1 "
               "--no-cpu-baseline --no-split-probe; per dispatch of split3_symm_kernel (main grid and split-K tail), means over the "
               "dispatches.  SQ_BUSY_CYCLES is summed over the 32 shader engines: clock = SQ_BUSY_CYCLES / 32 / duration; MFMA "
               "busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)", "dispatches": []}
for g in sorted(per, reverse=True):
    c = {k: sum(v) / len(v) for k, v in per[g].items()}
    ms = sum(dur[g].values()) / len(dur[g]) / 1e6
    cycles = c["SQ_BUSY_CYCLES"] / 32
    out["dispatches"].append({"kernel": name[g], "grid_size": g, "blocks": g // 512, "launches": len(dur[g]), "duration_ms": ms,
                              "sustained_clock_GHz": cycles / (ms * 1e6),
                              "mfma_busy_fraction": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles),
                              "lds_bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"],
                              "wait_any_fraction_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                              "wait_lds_fraction_of_wave_cycles": c.get("SQ_WAIT_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"],
                              "wait_inst_any_fraction_of_wave_cycles": c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"],
                              "active_inst_any_fraction_of_wave_cycles": c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"],
                              "counters": c})
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps([{k: v for k, v in d.items() if k != "counters"} for d in out["dispatches"]], indent=1))
