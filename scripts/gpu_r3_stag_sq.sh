# SQ counters of the product kernel under the default and the staggered K loop (MCGRA_SPLIT_LOOP=0 / 3), same box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
B2="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split-probe"
for v in 0 3; do
  export MCGRA_SPLIT_LOOP=$v
  cd /tmp
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$R/gpurun_out/stag_sq_$v" -- python3 $B2 > "$R/gpurun_out/stag_sq_$v.log" 2>&1
  cd "$R"
  python scripts/sq_counters_summary.py gpurun_out/stag_sq_$v gpurun_out/stag_sq_loop$v.json | grep -v counters | head -40
  find gpurun_out/stag_sq_$v -name "*kernel_trace*" -size +8M -delete 2>/dev/null
done
