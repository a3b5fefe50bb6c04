#!/bin/bash
# wall time of the driver's two bench commands on this box
cd "${GRAFT_REPO_ROOT:-.}"
t0=$(date +%s.%N); python bench.py > /tmp/b.json 2> /tmp/b.err; t1=$(date +%s.%N)
echo "python bench.py: $(echo "$t1 - $t0" | bc) s wall"
python -c "
import json; d=json.load(open('/tmp/b.json')); print(d['value'], d['ms_per_step'], list((d.get('other_workloads') or {}).keys()))"
t0=$(date +%s.%N); python bench.py --steps 20 --warmup 5 > /tmp/b2.json 2> /tmp/b2.err; t1=$(date +%s.%N)
echo "python bench.py --steps 20 --warmup 5: $(echo "$t1 - $t0" | bc) s wall"
python -c "
import json; d=json.load(open('/tmp/b2.json')); print(d['value'], d['ms_per_step'])"
