#!/bin/bash
# A/B of an environment switch inside the replayed step: tools/ab_step.sh VAR "v0 v1 ..." [reps]  (configs 3 and 4, step time only)
R=$GRAFT_REPO_ROOT; cd $R
VAR=$1; VALS=$2; REPS=${3:-2}
for rep in $(seq $REPS); do for v in $VALS; do
  for cfgargs in "--case 118v2 --batch 2048 --mode infer --steps 40 --warmup 5" "--case 6470rte --batch 64 --mode train --steps 12 --warmup 3"; do
    env $VAR=$v python bench.py $cfgargs --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --profile-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$VAR=$v', j['config']['workload'][:24], 'ms_per_step', j['ms_per_step'], 'median', j.get('median_ms_per_step'), 'min', j.get('min_ms_per_step'))
"
  done
done; done
