# staggered K loop of the product kernel (MCGRA_SPLIT_LOOP=3): race screen + bit identity with the default loop, then A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3stag
for v in 0 3; do
  MCGRA_SPLIT_LOOP=$v timeout 600 python scripts/race_screen.py 20 > gpurun_out/r3stag/race_$v.txt 2>&1
  echo "race loop=$v rc=$?"; tail -8 gpurun_out/r3stag/race_$v.txt
done
MCGRA_SPLIT_LOOP=3 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "split" 2>&1 | tail -3
VAR=MCGRA_SPLIT_LOOP VALS="0 3 2" bash scripts/gpu_r3_abc.sh
