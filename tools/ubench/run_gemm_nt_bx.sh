#!/bin/bash
# gemm_nt at large M: fp32 through the bf16 matrix cores (nine exact partial products, gemm_nt_bx_kernel; opt-in:
# PFN_NT_BX_MIN_TILES=2 = from 2 row tiles per wave) against the fp32-MFMA kernels (default), per launch shape of the big configs;
# the harness checks every 97th row against a float64 host dot product.  Run on the GPU box: bash tools/ubench/run_gemm_nt_bx.sh
R=$GRAFT_REPO_ROOT; C=$R/poweflownet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $R/tools/ubench/gemm_nt_bench.hip -L$C -lpfn_hip -Wl,-rpath,$C -o /tmp/nt_bench || exit 1
echo "-- correctness at ragged sizes (bx forced from one row tile per wave)"
for M in 65541 70001; do PFN_NT_BX_MIN_TILES=1 /tmp/nt_bench $M 129 129 4 1 3; PFN_NT_BX_MIN_TILES=1 /tmp/nt_bench $M 129 129 2 2 3; done
for M in 241664 414080; do
  for cfg in "129 129 1 1" "129 129 2 2" "129 129 2 1" "129 129 4 1"; do
    echo "-- M=$M $cfg"
    echo -n "bf16 x 9 : "; PFN_NT_BX_MIN_TILES=2 /tmp/nt_bench $M $cfg 20
    echo -n "fp32 mfma: "; /tmp/nt_bench $M $cfg 20
  done
done
