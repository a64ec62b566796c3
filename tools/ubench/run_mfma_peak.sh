#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w mfma_peak.hip -o /tmp/mfma_peak || exit 1
/tmp/mfma_peak
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w gemm_nt_bench.hip -L$R/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$R/poweflownet_amd/csrc -o /tmp/gemm_nt_bench || exit 1
for d in 0 1 125 61 ; do echo dbg=$d; PFN_GEMM_DBG=$d /tmp/gemm_nt_bench 414080 129 129 4 1 5; PFN_GEMM_DBG=$d /tmp/gemm_nt_bench 414080 128 128 4 1 5;  done
