#!/bin/bash
# round 6: does the memory-side traffic of the N x N x N product buy clock?  The 32 tiles an XCD runs at a
# time are GROUP_M x 32 / GROUP_M (default 4 x 8: 12 operand panels through its 4 MiB L2 per 32 tiles); MCGRA_SPLIT_GROUP_M (A/B only,
# beside MCGRA_AB=1) re-rasters them: 1 x 32 (33 panels), 2 x 16 (18), 4 x 8 (12), 8 x 4 (12), 16 x 2 (18).  Per setting: bench.py's own
# line on one box -- the product's live PMC traffic per launch, its launch time in situ and alone (20 replays), the step.  Alternating
# passes so that box drift shows.  MCGRA_EARLY_TAIL=0 throughout (the tail's early pass assumes groups of four row panels).
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r6/ab_raster
mkdir -p $OUT
export MCGRA_AB=1 MCGRA_EARLY_TAIL=0
for pass in 1 2; do
  for g in 4 1 2 8 16; do
    MCGRA_SPLIT_GROUP_M=$g timeout 400 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $OUT/g${g}_p${pass}.json 2> $OUT/g${g}_p${pass}.err
    python - <<PY
import json
d = json.load(open("$OUT/g${g}_p${pass}.json"))
r = d.get("roofline") or {}
al = r.get("alone") or {}
print("GROUP_M=%-2s pass $pass: traffic %.2f GB/launch (%s)  product %.3f ms in situ, %.3f ms alone  step %.3f ms  %.1f steps/s" % (
    "$g", (r.get("traffic") or 0) / 1e9, (r.get("traffic_source") or "?")[:24], r.get("avg_launch_ms") or 0, al.get("avg_launch_ms") or 0,
    d["ms_per_step"], d["value"]))
PY
  done
done | tee $OUT/summary.txt
