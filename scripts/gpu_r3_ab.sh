# A/B of the product kernel variants on one box: $VAR=$A vs $VAR=$B, alternating
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3ab
if [ -n "$PRETEST" ]; then env $VAR=$B python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "$PRETEST" 2>&1 | tail -3; fi
for i in 1 2 3; do
for v in $A $B; do
env $VAR=$v python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r3ab/b.json 2> gpurun_out/r3ab/b.err
python - "$v" <<'PY'
import json,sys
j=json.load(open('gpurun_out/r3ab/b.json'))
print(sys.argv[1], '10k', round(j['value'],2), round(j['ms_per_step'],3), 'insitu', round(j['roofline']['avg_launch_ms'],3), 'alone', round(j['roofline'].get('alone',{}).get('avg_launch_ms',0),3), 'auc', j['auc'])
PY
done; done
