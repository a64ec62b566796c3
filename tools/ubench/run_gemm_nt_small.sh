#!/bin/bash
# gemm_nt at tiny M: the serial chain of ONE tile (launch, weight DMA, A fragment, MFMAs, flush) vs the metric size
R=$GRAFT_REPO_ROOT; cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w gemm_nt_bench.hip -L$R/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$R/poweflownet_amd/csrc -o /tmp/gemm_nt_bench || exit 1
for cfg in "32 129 129 1 1" "256 129 129 1 1" "2048 129 129 1 1" "8192 129 129 1 1" "15104 129 129 1 1" "32 129 129 4 1" "2048 129 129 4 1" "8192 129 129 4 1" "15104 129 129 4 1" "32 129 32 1 1" "15104 129 32 1 1"; do /tmp/gemm_nt_bench $cfg 50 | grep -v "bad element"; done
