#!/bin/bash
# the three rocprofv3 passes of profiles/README.md (kernel stats, FETCH_SIZE, WRITE_SIZE) + one SQ pass; TAG names the round
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
B="$R/bench.py --steps ${STATS_STEPS:-60} --warmup ${STATS_WARMUP:-20} --no-cpu-baseline --no-split-probe"      # (long enough to be out of the clock's settling: the first ~30 steps of a run are 5 - 10 % slower)
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_stats" -- python3 $B > "$R/gpurun_out/${TAG}_stats.log" 2>&1
B2="$R/bench.py --steps ${PMC_STEPS:-2} --warmup 1 --no-cpu-baseline --no-split-probe"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_fetch" -- python3 $B2 > "$R/gpurun_out/${TAG}_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_write" -- python3 $B2 > "$R/gpurun_out/${TAG}_write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_sq" -- python3 $B2 > "$R/gpurun_out/${TAG}_sq.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_grbm" -- python3 $B2 > "$R/gpurun_out/${TAG}_grbm.log" 2>&1
cd "$R"
find gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_sq gpurun_out/${TAG}_grbm -name "*kernel_trace*" -size +8M -delete 2>/dev/null
du -sh gpurun_out/${TAG}_* | tail -8
