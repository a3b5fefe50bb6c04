#!/bin/bash
# build container: copy what scripts/gpu_r6_final.sh left under gpurun_out/r06f into profiles/ under the round's names
cd "$(dirname "$0")/.."; F=gpurun_out/r06f; P=profiles
cp $F/bench.json $P/r06_bench.json
cp $F/bench_20_5.json $P/r06_bench_steps20_warmup5.json
for f in $F/o_bench_*.json; do b=$(basename $f .json); cp $f $P/r06_bench_${b#o_bench_}.json; done
cp $F/diag_10k.txt $P/r06_diag_10k_vs_reference_and_ref64.txt
cp $F/readme_lines.txt $P/r06_readme_lines.txt
for ep in 20 100; do [ -f $F/readme_horizon$ep.txt ] && cp $F/readme_horizon$ep.txt $P/r06_readme_horizon$ep.txt; done
cp $F/shard_echo_10k_kl.log $P/r06_shard_emulate_10k_kl_echo.log
cp $F/shard_echo_10k_hsic.log $P/r06_shard_emulate_10k_echo.log
cp $F/single_plane_table.txt $P/r06_single_plane_table.txt
for wl in synthetic-10k-kl cora-shape-kl; do cp $F/kstats_$wl.txt $P/r06_kstats_$wl.txt; done
cp $F/citeseer_gat_step_timeline.txt $P/r06_citeseer_gat_step_timeline.txt
cp $F/gram_path_timeline.txt $P/r06_gram_path_timeline.txt
grep -E "passed|failed" $F/pytest.log | tail -1
ls $P | grep -c "^r06"
