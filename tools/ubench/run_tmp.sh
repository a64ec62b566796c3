#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for args in "--config wide --case 6470rte --batch 32 --steps 10 --warmup 3" "--case 6470rte --batch 64 --steps 10 --warmup 3 --hub-frac 0.2" "--case 14 --batch 32 --steps 50 --warmup 10" "--loss masked_l2"; do
python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ks=d.get('kernels') or {}
print('RUN $args |', 'ms/step', d['ms_per_step'], 'graphs/s', d['value'], '|', ' '.join(f\"{k}:{v['avg_us']:.0f}/{v.get('achieved','')}\" for k,v in ks.items()))
"
done
