#!/bin/bash
# gemm_nt at large M: the weight-streaming full-row kernel (default from 2 row tiles per wave) against the weight-stationary one
# (PFN_NT_WS_MIN_TILES=0), per launch shape of the big-graph configs.  Run on the GPU box: tools/ubench/run_gemm_nt_ws.sh
R=$GRAFT_REPO_ROOT; C=$R/poweflownet_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $R/tools/ubench/gemm_nt_bench.hip -L$C -lpfn_hip -Wl,-rpath,$C -o /tmp/nt_bench || exit 1
for M in 414080 241664; do
  for cfg in "129 129 1 1" "129 129 2 2" "129 129 2 1" "129 129 4 1"; do
    echo "-- M=$M $cfg"
    echo -n "streaming : "; /tmp/nt_bench $M $cfg 20 | grep -v "bad element"
    echo -n "stationary: "; PFN_NT_WS_MIN_TILES=0 /tmp/nt_bench $M $cfg 20 | grep -v "bad element"
  done
done
