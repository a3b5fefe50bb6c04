#!/bin/bash
# round 6: kernel tables of the fused KL step (N = 10 000 and Cora shape) and of the fused MSELoss step beside it, from rocprofv3 --stats
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out/r6
export TMPDIR=/tmp
cd /tmp
for wl in synthetic-10k-kl synthetic-10k-mse cora-shape-kl; do
  rm -rf "$R/gpurun_out/r6/stats_$wl"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r6/stats_$wl" -- \
    python3 "$R/bench.py" --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/r6/stats_$wl.log" 2>&1
  (cd "$R"; echo "== $wl (50 steps incl. warm-up)"; python3 scripts/kstats.py gpurun_out/r6/stats_$wl 50 22) | tee "$R/gpurun_out/r6/kstats_$wl.txt"
  find "$R/gpurun_out/r6/stats_$wl" -name "*kernel_trace*" -size +8M -delete 2>/dev/null
done
