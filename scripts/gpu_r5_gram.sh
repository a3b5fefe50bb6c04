#!/bin/bash
# GPU box (round 5): where the Gram evaluation (MCGRA_NO_LOWRANK=1) spends its step at N = 10 000 -- kernel stats + a serial
# launch list of one step -> profiles/r05_gram_path_*; and the headline line in the driver's window as a sanity check
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${TAG:-r05a}
cd /tmp
rm -rf "$R/gpurun_out/${TAG}_gram"
MCGRA_AB=1 MCGRA_NO_LOWRANK=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_gram" -- \
  python3 "$R/bench.py" --steps 8 --warmup 3 --no-cpu-baseline --no-split-probe > "$R/gpurun_out/${TAG}_gram.log" 2>&1
cd "$R"
python3 scripts/kstats.py gpurun_out/${TAG}_gram 11 40 > gpurun_out/${TAG}_gram_kstats.txt 2>&1
T=$(find gpurun_out/${TAG}_gram -name "*kernel_trace.csv" | head -1)
python3 scripts/gram_timeline.py "$T" > gpurun_out/${TAG}_gram_timeline.txt 2>&1
find gpurun_out/${TAG}_gram -name "*kernel_trace*" -size +8M -delete 2>/dev/null
tail -1 gpurun_out/${TAG}_gram.log | cut -c1-600
head -45 gpurun_out/${TAG}_gram_kstats.txt
head -120 gpurun_out/${TAG}_gram_timeline.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench20.json 2> gpurun_out/${TAG}_bench20.err
cut -c1-500 gpurun_out/${TAG}_bench20.json
