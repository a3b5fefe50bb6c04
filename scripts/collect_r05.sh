#!/bin/bash
# build container: copy what scripts/gpu_r5_final.sh left under gpurun_out/r05f into profiles/ under the round's names
cd "$(dirname "$0")/.."; F=gpurun_out/r05f; P=profiles
cp $F/bench.json $P/r05_bench.json
cp $F/bench_20_5.json $P/r05_bench_steps20_warmup5.json
for f in $F/o_bench_*.json; do b=$(basename $f .json); cp $f $P/r05_bench_${b#o_bench_}.json; done
cp $F/diag_10k.txt $P/r05_diag_10k_vs_reference_and_ref64.txt
cp $F/readme_lines.txt $P/r05_readme_lines.txt
for ep in 20 100; do [ -f $F/readme_horizon$ep.txt ] && cp $F/readme_horizon$ep.txt $P/r05_readme_horizon$ep.txt; done
cp $F/shard_echo_10k_hsic.log $P/r05_shard_emulate_10k_echo.log
cp $F/shard_echo_10k_mse.log $P/r05_shard_emulate_10k_mse_echo.log
cp $F/shard_echo_30k_hsic.log $P/r05_shard_emulate_30k_echo.log
cp $F/citeseer_gat_steps.txt $P/r05_citeseer_gat_steps.txt
[ -f $F/citeseer_gat_step_timeline.txt ] && cp $F/citeseer_gat_step_timeline.txt $P/r05_citeseer_gat_step_timeline.txt
cp $F/gemm_mid_bench.txt $P/r05_gemm_mid_bench.txt
cp $F/ab_early_p1.txt $P/r05_ab_early_p1.txt
grep -E "passed|failed" $F/pytest.log | tail -1
ls $P | grep -c "^r05"
