cd "${GRAFT_REPO_ROOT:-/root/repo}"
python scripts/state_hash.py cora-shape-hsic 50 2>&1 | grep -v amdgpu; python scripts/state_hash.py synthetic-4k-hsic 30 2>&1 | grep -v amdgpu; python scripts/state_hash.py synthetic-10k-hsic 12 2>&1 | grep -v amdgpu
python scripts/shard_state_hash.py synthetic-4k-hsic 4 12 2>&1 | grep -v amdgpu; python scripts/shard_state_hash.py cora-shape-mse 3 20 2>&1 | grep -v amdgpu
for w in hsic kl; do python3 scripts/citeseer_gat_steps.py $w 60 2>&1 | grep "ms/step"; done
python3 bench.py --workload cora-shape-hsic --steps 200 --warmup 20 --no-cpu-baseline --no-split-probe 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cora-shape-hsic', round(d['value'],1), round(d['ms_per_step'],4))"
python3 scripts/shard_emulate.py --echo --worlds 8 --steps 40 2>&1 | grep '^{"world"' | cut -c1-200
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-split-probe 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('10k', round(d['value'],1), round(d['ms_per_step'],4))"
python3 bench.py --workload synthetic-4k-hsic --steps 200 --warmup 20 --no-cpu-baseline --no-split-probe 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4k', round(d['value'],1), round(d['ms_per_step'],4))"
