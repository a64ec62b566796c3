#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for e in "X=1" "PFN_NO_SIDE_STREAM=1"; do
for args in "" "--case 6470rte --batch 64 --steps 10 --warmup 3"; do
env $e python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ks=d.get('kernels') or {}
print('RUN $e', d['config']['workload'][:12], 'ms/step', d['ms_per_step'], ' '.join(f\"{k}:{v['avg_us']:.0f}/{v.get('achieved','')}\" for k,v in ks.items()))
"
done; done
