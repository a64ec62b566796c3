#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for args in "" "--mode infer --batch 2048 --steps 20 --warmup 5"; do
python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['config']['workload'][:40], 'ms/step', d['ms_per_step'], 'value', d['value'], '| nt:', r['achieved'], r['unit'], 'frac', r['frac'], 'avg_us', r['avg_launch_us'], 'launches', r['launches_per_step'])
ks=d.get('kernels') or {}
for k,v in ks.items(): print('   ', k, v.get('launches_per_step'), v.get('avg_us'), v.get('ms_per_step'))
"
done
