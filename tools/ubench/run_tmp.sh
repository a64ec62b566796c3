#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for args in "" "--case 6470rte --batch 64 --steps 10 --warmup 3"; do
PFN_NO_SIDE_STREAM=1 python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ks=d.get('kernels') or {}
print('NOSIDE', d['config']['workload'][:12], 'ms/step', d['ms_per_step'], 'tn', ks['gemm_tn']['avg_us'], ks['gemm_tn'].get('achieved'), 'reduce', ks.get('tn_reduce',{}).get('avg_us'), 'nt', ks['gemm_nt']['avg_us'], ks['gemm_nt'].get('achieved'))
"
done
