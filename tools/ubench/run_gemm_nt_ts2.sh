#!/bin/bash
export ASYNC_CHECK=${GRAFT_REPO_ROOT:-/root/repo}/tools/check_async_fragments.py   # (csrc/Makefile checks the ISA of the async-fragment objects it links)
# per-wave shader-cycle accounting of gemm_nt_kernel's round loop (NT_EXP_TS2 build in /tmp), through the ubench harness
R=$GRAFT_REPO_ROOT; d=/tmp/exp_nt_ts2; rm -rf $d; mkdir -p $d; cp -r $R/poweflownet_amd $R/include $d/; bash $R/tools/ubench/apply_experiments.sh $d/poweflownet_amd/csrc
( cd $d/poweflownet_amd/csrc && rm -f *.o libpfn_hip.so && make -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DNT_EXP_TS2 $NT_EXTRA" > /dev/null ) || exit 1
cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DNT_EXP_TS2 gemm_nt_bench.hip -L$d/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$d/poweflownet_amd/csrc -o /tmp/gemm_nt_bench_ts2 || exit 1
for cfg in "414080 129 129 1 1" "414080 129 129 4 1" "241664 129 129 4 1"; do PFN_NT_TINY_MAX_TILES=0 /tmp/gemm_nt_bench_ts2 $cfg 20 | grep -v "bad element"; done
