#!/bin/bash
# builds and runs the gemm_nt harness on the GPU box (gpurun -- tools/ubench/run_gemm_nt.sh)
R=$GRAFT_REPO_ROOT; cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w gemm_nt_bench.hip -L$R/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$R/poweflownet_amd/csrc -o /tmp/gemm_nt_bench || exit 1
for cfg in "15104 129 129 1 1" "15104 129 129 2 2" "15104 129 129 4 1" "15104 129 129 4 4" "15104 129 4 1 1" "15104 4 129 2 2" "414080 129 129 1 1" "414080 129 129 2 2" "414080 129 129 4 1" "414080 129 129 4 4" "414080 128 128 4 1" "414080 129 4 1 1" "414080 4 129 2 2" "100000 300 200 2 1" "100000 600 64 4 1"; do /tmp/gemm_nt_bench $cfg 20; done
