#!/bin/bash
export ASYNC_CHECK=${GRAFT_REPO_ROOT:-/root/repo}/tools/check_async_fragments.py   # (csrc/Makefile checks the ISA of the async-fragment objects it links)
# A/B inside the replayed step: the tree's library vs the library built from the sources under ab_ref/ (a copy of another commit's
# poweflownet_amd/csrc + include, made before the gpurun call: the box has no .git).  tools/ab_ref.sh [reps]
R=$GRAFT_REPO_ROOT; REPS=${1:-2}
d=/tmp/ab_r; rm -rf $d; mkdir -p $d; cp -r $R/poweflownet_amd $R/include $R/bench.py $R/configs $R/BASELINE.json $d/
cp $R/ab_ref/poweflownet_amd/csrc/*.hip $R/ab_ref/poweflownet_amd/csrc/*.hpp $d/poweflownet_amd/csrc/; cp $R/ab_ref/include/* $d/include/
( cd $d/poweflownet_amd/csrc && rm -f *.o libpfn_hip.so && make -j16 > /dev/null ) || exit 1
for rep in $(seq $REPS); do for side in A B; do
  if [ $side = A ]; then cd $R; tag="tree"; else cd $d; tag="ref "; fi
  for cfgargs in "--case 118v2 --batch 128 --mode train --steps 200 --warmup 20" "--case 118v2 --batch 2048 --mode infer --steps 40 --warmup 5" "--case 6470rte --batch 64 --mode train --steps 12 --warmup 3" $EXTRA_CFG; do
    python bench.py $cfgargs --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --profile-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('[$tag]', j['config']['workload'][:12], j['metric'][:30], 'ms_per_step', j['ms_per_step'], 'median', j.get('median_ms_per_step'), 'min', j.get('min_ms_per_step'))
"
  done
done; done
