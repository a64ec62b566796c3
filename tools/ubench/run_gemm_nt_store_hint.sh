#!/bin/bash
# gemm_nt's large-M output stores with a cache-policy hint (nt / sc1 / sc0 sc1) against plain stores: config-3 and config-4 steps
R=$GRAFT_REPO_ROOT
export ASYNC_CHECK=$R/tools/check_async_fragments.py
for v in plain nt "sc0 sc1" ; do
  tag=$(echo $v | tr -d ' ')
  d=/tmp/exp_sh_$tag; rm -rf $d; mkdir -p $d; cp -r $R/poweflownet_amd $R/include $R/bench.py $R/oracle $R/configs $d/ 2>/dev/null
  if [ "$v" != plain ]; then
    sed -i "s/else asm volatile(\"global_store_dwordx4 %0, %1, %2\\\\n/else asm volatile(\"global_store_dwordx4 %0, %1, %2 $v\\\\n/; s/global_store_dwordx4 %1, %2, %3\\\\n/global_store_dwordx4 %1, %2, %3 $v\\\\n/" $d/poweflownet_amd/csrc/gemm_nt.hip
    grep -c "dwordx4 %0, %1, %2 $v\|%1, %2, %3 $v" $d/poweflownet_amd/csrc/gemm_nt.hip
  fi
  ( cd $d/poweflownet_amd/csrc && rm -f gemm_nt.o libpfn_hip.so && make -j16 libpfn_hip.so > /tmp/sh_make.log 2>&1 ) || tail -5 /tmp/sh_make.log
  test -f $d/poweflownet_amd/csrc/libpfn_hip.so || exit 1
  for rep in 1 2; do
  ( cd $d && python bench.py --mode infer --batch 2048 --no-cpu-baseline --no-live-traffic --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] config 3 ms_per_step', d['ms_per_step'], 'gemm_nt', d['kernels']['gemm_nt']['avg_us'])" )
  done
  ( cd $d && python bench.py --case 6470rte --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] config 4 ms_per_step', d['ms_per_step'], 'gemm_nt', d['kernels']['gemm_nt']['avg_us'])" )
done
