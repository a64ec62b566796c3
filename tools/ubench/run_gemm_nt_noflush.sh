#!/bin/bash
# Bound of ANY hidden / cheaper flush in the stationary gemm_nt kernel: the whole flush compiled away behind a run-time-false test
# (results WRONG by design, timings only), measured in the replayed config-3 step (realistic clocks) and in the harness.
R=$GRAFT_REPO_ROOT
export ASYNC_CHECK=$R/tools/check_async_fragments.py
for v in BASE NOFLUSH; do
  d=/tmp/exp_nf_$v; rm -rf $d; mkdir -p $d; cp -r $R/poweflownet_amd $R/include $R/bench.py $R/oracle $R/configs $d/ 2>/dev/null
  if [ $v = NOFLUSH ]; then sed -i 's/        if (flush_after) {$/        if (flush_after \&\& a.M < 0) {/' $d/poweflownet_amd/csrc/gemm_nt.hip; grep -c "flush_after && a.M < 0" $d/poweflownet_amd/csrc/gemm_nt.hip; fi
  ( cd $d/poweflownet_amd/csrc && rm -f gemm_nt.o libpfn_hip.so && make -j16 libpfn_hip.so > /tmp/nf_make.log 2>&1 ) || { tail -5 /tmp/nf_make.log; }; test -f $d/poweflownet_amd/csrc/libpfn_hip.so || { echo build failed; exit 1; }
  cd $R/tools/ubench
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w gemm_nt_bench.hip -L$d/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$d/poweflownet_amd/csrc -o $d/bench || exit 1
  echo "== $v harness"
  for cfg in "241664 129 129 1 1" "241664 129 129 2 2" "241664 129 129 4 1"; do PFN_NT_TINY_MAX_TILES=0 $d/bench $cfg 20 | grep -v "bad element\|waves:"; done
  echo "== $v config-3 step"
  ( cd $d && python bench.py --mode infer --batch 2048 --no-cpu-baseline --no-live-traffic --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'gemm_nt', d['kernels'].get('gemm_nt'))" )
done
