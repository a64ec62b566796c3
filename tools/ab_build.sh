#!/bin/bash
export ASYNC_CHECK=${GRAFT_REPO_ROOT:-/root/repo}/tools/check_async_fragments.py   # (csrc/Makefile checks the ISA of the async-fragment objects it links)
# A/B of COMPILE-TIME switches inside the step: tools/ab_build.sh "-DFLAG ..." [reps] [kernel class]  -- the tree's library against the
# same sources built with the flags; configs 2, 3 and 4: step time and the named kernel class's average launch (eager profile pass)
R=$GRAFT_REPO_ROOT; FLAGS=$1; REPS=${2:-2}; KCLASS=${3:-gemm_tn}
d=/tmp/ab_b; rm -rf $d; mkdir -p $d; cp -r $R/poweflownet_amd $R/include $R/bench.py $R/configs $R/BASELINE.json $d/
( cd $d/poweflownet_amd/csrc && rm -f *.o libpfn_hip.so && make -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $FLAGS" > /dev/null ) || exit 1
for rep in $(seq $REPS); do for side in A B; do
  if [ $side = A ]; then cd $R; tag="tree"; else cd $d; tag="$FLAGS"; fi
  for cfgargs in "--case 118v2 --batch 128 --mode train --steps 200 --warmup 20" "--case 118v2 --batch 2048 --mode infer --steps 40 --warmup 5" "--case 6470rte --batch 64 --mode train --steps 12 --warmup 3"; do
    python bench.py $cfgargs --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --profile-steps 3 2>/dev/null | KCLASS=$KCLASS python -c "
import sys, json, os
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = (j.get('kernels') or {}).get(os.environ['KCLASS'], {})
        print('[$tag]', j['config']['workload'][:12], j['metric'][:22], 'ms_per_step', j['ms_per_step'], 'min', j.get('min_ms_per_step'), '|', os.environ['KCLASS'], k.get('avg_us'), 'us x', k.get('launches_per_step'))
"
  done
done; done
