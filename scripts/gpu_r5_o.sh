#!/bin/bash
# kernel tables of the README lines that are slow for their size (KDE, DP, CKA on cora / polblogs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for nm in readme_cora_kde_Y readme_cora_dp_all_eps_neg readme_polblogs_cka_yy readme_cora_cka_yY readme_citeseer_kl_all; do
  (cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r05o_$nm" -- python3 "$R/scripts/readme_line_steps.py" $nm 40 > "$R/gpurun_out/r05o_$nm.log" 2>&1)
  grep "ms/step" gpurun_out/r05o_$nm.log
  python3 scripts/kstats.py gpurun_out/r05o_$nm 45 14
  rm -rf gpurun_out/r05o_$nm
done
