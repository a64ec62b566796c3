#!/bin/bash
# serpentine sweeps (consecutive row-streaming launches alternate their direction) against PFN_NO_SERPENTINE=1, same box
cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3; do for off in 1 0; do
  if [ $off = 1 ]; then export PFN_NO_SERPENTINE=1; else unset PFN_NO_SERPENTINE; fi
  python bench.py --mode infer --batch 2048 --no-cpu-baseline --no-live-traffic --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_serpentine=$off config 3', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done; done
for rep in 1 2; do for off in 1 0; do
  if [ $off = 1 ]; then export PFN_NO_SERPENTINE=1; else unset PFN_NO_SERPENTINE; fi
  python bench.py --case 6470rte --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_serpentine=$off config 4', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items() if v['ms_per_step']>0.3})"
  python bench.py --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_serpentine=$off config 2', d['ms_per_step'], d['median_ms_per_step'])"
done; done
