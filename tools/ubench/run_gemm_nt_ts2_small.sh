#!/bin/bash
export ASYNC_CHECK=${GRAFT_REPO_ROOT:-/root/repo}/tools/check_async_fragments.py   # (csrc/Makefile checks the ISA of the async-fragment objects it links)
# per-wave cycle accounting of the small-M (one tile per wave) gemm_nt launches of config 2: who ends last?
R=$GRAFT_REPO_ROOT; d=/tmp/exp_nt_ts2; rm -rf $d; mkdir -p $d; cp -r $R/poweflownet_amd $R/include $d/; bash $R/tools/ubench/apply_experiments.sh $d/poweflownet_amd/csrc
( cd $d/poweflownet_amd/csrc && rm -f *.o libpfn_hip.so && make -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DNT_EXP_TS2 $NT_EXTRA" > /dev/null ) || exit 1
cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DNT_EXP_TS2 gemm_nt_bench.hip -L$d/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$d/poweflownet_amd/csrc -o /tmp/gemm_nt_bench_ts2 || exit 1
for cfg in "15104 129 129 4 1" "15104 129 129 2 2" "15104 129 129 1 1"; do PFN_TS2_GX=59 PFN_NT_TINY_MAX_TILES=0 /tmp/gemm_nt_bench_ts2 $cfg 20 | grep -v "bad element\|by XCD"; done
