#!/bin/bash
# builds and runs the gemm_nt harness on the GPU box (gpurun -- tools/ubench/run_gemm_nt.sh)
R=$GRAFT_REPO_ROOT; cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w gemm_nt_bench.hip -L$R/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$R/poweflownet_amd/csrc -o /tmp/gemm_nt_bench || exit 1
PFN_NT_TIMING=1 /tmp/gemm_nt_bench 414080 129 129 4 1 5
for d in 16 32 48 64 112 113; do echo dbg=$d; PFN_GEMM_DBG=$d PFN_NT_TIMING=1 /tmp/gemm_nt_bench 414080 129 129 4 1 5 | sed -n '1p;4,5p'; done
for cfg in "15104 129 129 1 1" "15104 129 129 4 1" "414080 129 129 4 1" "414080 129 129 4 4"; do /tmp/gemm_nt_bench $cfg 20; done
