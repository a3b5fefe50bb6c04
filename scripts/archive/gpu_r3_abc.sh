# A/B/.. of env-switched variants on one box: for v in $VALS: $VAR=v, three rounds
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3ab
for i in 1 2 3; do
for v in $VALS; do
env $VAR=$v python bench.py --no-cpu-baseline --steps 20 > gpurun_out/r3ab/b.json 2> gpurun_out/r3ab/b.err
python - "$v" <<'PY'
import json,sys
j=json.load(open('gpurun_out/r3ab/b.json'))
print(sys.argv[1], '10k', round(j['value'],2), round(j['ms_per_step'],3), 'insitu', round(j['roofline']['avg_launch_ms'],3), 'alone', round(j['roofline'].get('alone',{}).get('avg_launch_ms',0),3), 'auc', j['auc'])
PY
done; done
