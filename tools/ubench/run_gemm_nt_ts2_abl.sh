#!/bin/bash
export ASYNC_CHECK=${GRAFT_REPO_ROOT:-/root/repo}/tools/check_async_fragments.py   # (csrc/Makefile checks the ISA of the async-fragment objects it links)
# per-wave cycle accounting (NT_EXP_TS2) with one ingredient removed at a time (results WRONG by design): what are the flush's cycles?
R=$GRAFT_REPO_ROOT
for v in BASE NOSTORE NOREFILL "NOSTORE -DPFN_EXP_NOREFILL"; do
  d=/tmp/exp_abl_$(echo $v | tr -d ' -'); rm -rf $d; mkdir -p $d; cp -r $R/poweflownet_amd $R/include $d/; bash $R/tools/ubench/apply_experiments.sh $d/poweflownet_amd/csrc
  ( cd $d/poweflownet_amd/csrc && rm -f *.o libpfn_hip.so && make -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DNT_EXP_TS2 -DPFN_EXP_$v" > /dev/null ) || exit 1
  cd $R/tools/ubench
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DNT_EXP_TS2 gemm_nt_bench.hip -L$d/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$d/poweflownet_amd/csrc -o $d/bench || exit 1
  echo "== $v"
  for cfg in "414080 129 129 1 1" "65536 129 129 1 1" "414080 129 129 4 1" "65536 129 129 4 1"; do PFN_NT_TINY_MAX_TILES=0 $d/bench $cfg 20 | grep -v "bad element\|waves:"; done
done
