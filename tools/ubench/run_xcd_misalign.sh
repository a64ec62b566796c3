#!/bin/bash
# Does a consumer gain from running on the XCD whose L2 its producer wrote?  Proxy: seg_lin_hops_kernel's graph index shifted by one
# (graph g then runs on XCD (g + 1) % 8 instead of g % 8: its input from ea_seg_fwd / its output to the next launches cross XCDs);
# results identical, only placement changes.  Config-2 step, alternating with the tree's library.
R=$GRAFT_REPO_ROOT
export ASYNC_CHECK=$R/tools/check_async_fragments.py HSA_ENABLE_IPC_MODE_LEGACY=0
d=/tmp/exp_mis; rm -rf $d; mkdir -p $d; cp -r $R/poweflownet_amd $R/include $R/bench.py $R/oracle $R/configs $d/ 2>/dev/null
sed -i 's/    const int r0 = blockIdx.x \* rows_pb, rows = min(rows_pb, n - r0);/    const int r0 = (int)((blockIdx.x + 1) % gridDim.x) * rows_pb, rows = min(rows_pb, n - r0);/' $d/poweflownet_amd/csrc/seg_lin_hops.hip
grep -c "blockIdx.x + 1" $d/poweflownet_amd/csrc/seg_lin_hops.hip
( cd $d/poweflownet_amd/csrc && rm -f seg_lin_hops.o libpfn_hip.so && make -j16 libpfn_hip.so > /tmp/mis_make.log 2>&1 ) || tail -5 /tmp/mis_make.log
for rep in 1 2 3; do for side in tree shifted; do
  if [ $side = tree ]; then cd $R; else cd $d; fi
  python bench.py --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('[$side]', d['ms_per_step'], d['median_ms_per_step'], {n:k[n]['avg_us'] for n in ('seg_lin_hops_fwd','seg_lin_hops_bwd','gemm_nt','ea_seg_fwd','ea_seg_bwd') if n in k})"
done; done
