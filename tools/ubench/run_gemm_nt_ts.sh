#!/bin/bash
# phase timestamps of gemm_nt_kernel (NT_EXP_TS build in /tmp): per workgroup, wall clock 100 MHz, through the ubench harness
R=$GRAFT_REPO_ROOT; d=/tmp/exp_nt_ts; mkdir -p $d; cp -r $R/poweflownet_amd $R/include $d/; bash $R/tools/ubench/apply_experiments.sh $d/poweflownet_amd/csrc
( cd $d/poweflownet_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DNT_EXP_TS -c gemm_nt.hip -o gemm_nt.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC graph.o edge.o gemm.o gemm_nt.o front.o ea_seg.o seg_lin_hops.o model.o physics.o prof.o -o libpfn_hip.so ) || exit 1
cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DNT_EXP_TS gemm_nt_bench.hip -L$d/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$d/poweflownet_amd/csrc -o /tmp/gemm_nt_bench_ts || exit 1
for cfg in "15104 129 129 1 1" "15104 129 129 4 1" "15104 129 129 2 1"; do PFN_NT_TINY_MAX_TILES=0 /tmp/gemm_nt_bench_ts $cfg 20 | grep -v "bad element"; done
