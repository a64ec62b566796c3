#!/bin/bash
# What of gemm_nt's flush is the WRITE STREAM: every output store of gemm_nt.hip turned into s_nop (results wrong by design), the
# vector work of the flush kept; config-3 step + harness.  Companion of run_gemm_nt_noflush.sh.
R=$GRAFT_REPO_ROOT
export ASYNC_CHECK=$R/tools/check_async_fragments.py
for v in BASE NOSTORE; do
  d=/tmp/exp_ns_$v; rm -rf $d; mkdir -p $d; cp -r $R/poweflownet_amd $R/include $R/bench.py $R/oracle $R/configs $d/ 2>/dev/null
  if [ $v = NOSTORE ]; then sed -i 's/global_store_dwordx4 %[0-9], %[0-9], %[0-9]\( sc1\)\?/s_nop 0/; s/global_store_dwordx4 %0, %1, off\( sc1\)\?/s_nop 0/' $d/poweflownet_amd/csrc/gemm_nt.hip; grep -c "global_store" $d/poweflownet_amd/csrc/gemm_nt.hip; fi
  ( cd $d/poweflownet_amd/csrc && rm -f gemm_nt.o libpfn_hip.so && make -j16 libpfn_hip.so > /tmp/ns_make.log 2>&1 ) || { tail -5 /tmp/ns_make.log; }; test -f $d/poweflownet_amd/csrc/libpfn_hip.so || exit 1
  cd $R/tools/ubench
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w gemm_nt_bench.hip -L$d/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$d/poweflownet_amd/csrc -o $d/bench || exit 1
  echo "== $v harness"
  for cfg in "241664 129 129 1 1" "241664 129 129 2 2" "241664 129 129 4 1"; do PFN_NT_TINY_MAX_TILES=0 $d/bench $cfg 20 | grep -v "bad element\|waves:"; done
  echo "== $v config-3 step"
  ( cd $d && python bench.py --mode infer --batch 2048 --no-cpu-baseline --no-live-traffic --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'gemm_nt', d['kernels'].get('gemm_nt'))" )
done
