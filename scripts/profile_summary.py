"""Turn the rocprofv3 outputs of a bench.py run (gpurun_out/<tag>_{stats,fetch,write}/...) into the committed
summaries under profiles/: kernel stats CSV, per-kernel PMC summary, GEMM traffic JSON.

    python scripts/profile_summary.py r01          (after the three rocprofv3 passes listed in profiles/README.md)

FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1 KB as rocprofv3 prints them; FETCH_SIZE under-reports by 2x on
gfx950 (calibrated on k_rowsum: one 400 MB read shows 200 MB), so HBM-side bytes = 2 * FETCH + WRITE.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def one(pattern):
    fs = sorted(glob.glob(os.path.join(G, pattern), recursive=True), key=os.path.getmtime)
    if not fs:
        sys.exit(f"missing {pattern}")
    return fs[-1]


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


shutil.copy(one(f"{tag}_stats/**/*_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
per = defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "ns": 0.0})
for ctr, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    src = one(f"{tag}_{d}/**/*_counter_collection.csv")
    shutil.copy(src, os.path.join(P, f"{tag}_pmc_{d}_size.csv"))
    for row in csv.DictReader(open(src)):
        if row["Counter_Name"] != ctr:
            continue
        k = short(row["Kernel_Name"]) + f" grid={row['Grid_Size']}"
        per[k][ctr] += float(row["Counter_Value"])
        if ctr == "FETCH_SIZE":
            per[k]["n"] += 1
            per[k]["ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
rows = []
for k, v in per.items():
    n = max(v["n"], 1)
    hbm = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) / n * 1024
    rows.append((v["ns"], k, n, v["FETCH_SIZE"] / n, v["WRITE_SIZE"] / n, hbm, v["ns"] / n / 1e6))
rows.sort(reverse=True)
with open(os.path.join(P, f"{tag}_pmc_summary.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "fetch_size_kb_per_launch", "write_size_kb_per_launch",
                "hbm_bytes_per_launch_corrected", "avg_ms_under_pmc", "corrected_GBps"])
    for _, k, n, fe, wr, hbm, ms in rows:
        w.writerow([k, n, f"{fe:.1f}", f"{wr:.1f}", f"{hbm:.0f}", f"{ms:.4f}", f"{hbm / (ms * 1e-3) / 1e9:.1f}" if ms else ""])
gem = [r for r in rows if ("gemm_f32_kernel<128, 128" in r[1] or "split3_symm_kernel" in r[1] or "split2_m16_kernel" in r[1]) and r[5] > 2e9]
out = {"note": "HBM-side bytes per launch of the N x N x N fp32 MFMA GEMM launches (synthetic-10k-hsic; one launch = a batched "
               "pair of products), from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled per the "
               "gfx950 correction (calibrated on k_rowsum: 2 x FETCH_SIZE = 400.7 MB for a 400 MB read)",
       "kernels": [{"kernel": k, "role": "split_f16" if ("split3_symm_kernel<2" in k or "split2_m16_kernel" in k) else "split" if "split3_symm_kernel" in k else "symm" if ", 2, 2, 1>" in k else "syrk" if ", 2, 1, 1>" in k else "gemm", "launches": n, "fetch_size_kb": fe, "write_size_kb": wr, "hbm_bytes_corrected": hbm,
                    "avg_ms": ms} for _, k, n, fe, wr, hbm, ms in gem]}
# whole-step traffic outside the product: every mcgra kernel launched once per step or more (launch counts are multiples of
# the timed + warm-up steps of the pass), per step
steps_in_pass = int(sys.argv[2]) if len(sys.argv) > 2 else 3
# (set-up and post-loop kernels -- create-time zero fills, set_graph's products, finalize's decode -- are not part of a step:
#  excluded by name, and a launch count that is not a multiple of the steps is floored)
NOT_IN_STEP = ("fillBufferAligned grid=65536", "k_dd2_accum", "k_axpby2d", "rankk_nt_kernel<16>", "k_decode_post", "k_unpack_sym",
               "k_split_absmax", "k_center_cols", "k_rowsum", "copyBuffer")
step_rows = [(k, float(n // steps_in_pass), hbm, ms) for _, k, n, fe, wr, hbm, ms in rows
             if ("mcgra::" in k or "rocclr" in k) and n >= steps_in_pass and "gemm_f32_kernel<128, 128" not in k
             and not any(x in k for x in NOT_IN_STEP)]
prod = sum(hbm * c for k, c, hbm, ms in step_rows if "split" in k and "k_split3_reduce" not in k)
rest = sum(hbm * c for k, c, hbm, ms in step_rows if not ("split" in k and "k_split3_reduce" not in k))
rest_ms = sum(ms * c for k, c, hbm, ms in step_rows if not ("split" in k and "k_split3_reduce" not in k))
json.dump({"note": "HBM-side bytes per attack step (2 x FETCH_SIZE + WRITE_SIZE per the gfx950 correction), summed over the "
                   "kernels a step launches, from the serialised PMC passes; `outside_product` is everything but the N x N x N "
                   "product launches", "steps_in_pass": steps_in_pass,
           "product_bytes_per_step": prod, "outside_product_bytes_per_step": rest, "outside_product_ms_per_step_under_pmc": rest_ms,
           "outside_product_GBps": rest / (rest_ms * 1e-3) / 1e9 if rest_ms else None,
           "kernels": [{"kernel": k, "launches_per_step": c, "hbm_bytes_per_launch": hbm, "avg_ms": ms} for k, c, hbm, ms in
                       sorted(step_rows, key=lambda r: -r[1] * r[2])[:40]]},
          open(os.path.join(P, f"{tag}_step_traffic.json"), "w"), indent=1)
print("per step: product", prod / 1e9, "GB; outside", rest / 1e9, "GB in", rest_ms, "ms")
# the split product is ONE product per step in several launches (parts of a cut grid, the ragged round's split-K launch, the sum
# of its slabs): one entry for all of them, per product
sp = [(k, c, hbm, ms) for k, c, hbm, ms in step_rows if "split2_m16_kernel" in k or "split3_symm_kernel" in k or "k_split3_reduce" in k]
if sp:
    out["kernels"] = [k for k in out["kernels"] if k["role"] not in ("split_f16", "split")]
    out["kernels"].insert(0, {"kernel": "split2_m16_kernel: one product = " + " + ".join(f"{int(c)} x {k}" for k, c, hbm, ms in sp),
                              "role": "split_f16" if any("split2_m16" in k for k, *_ in sp) else "split", "launches": steps_in_pass,
                              "fetch_size_kb": None, "write_size_kb": None,
                              "hbm_bytes_corrected": sum(hbm * c for k, c, hbm, ms in sp),
                              "avg_ms": sum(ms * c for k, c, hbm, ms in sp)})
# entries of other evaluations of the product (earlier passes with MCGRA_SPLIT_BF16=0 / 2) stay in the file
old_path = os.path.join(P, f"{tag}_gemm_traffic.json")
if os.path.exists(old_path):
    have = {k["role"] for k in out["kernels"]}
    out["kernels"] += [k for k in json.load(open(old_path))["kernels"] if k.get("role") not in have]
json.dump(out, open(os.path.join(P, f"{tag}_gemm_traffic.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
