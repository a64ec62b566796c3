#!/bin/bash
# the interleaved flush against the plain one, back to back (harness) at the config-3 / config-4 row counts
R=$GRAFT_REPO_ROOT; cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w gemm_nt_bench.hip -L$R/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$R/poweflownet_amd/csrc -o /tmp/gemm_nt_bench || exit 1
for rep in 1 2; do
for cfg in "241664 129 129 1 1" "241664 129 129 2 2" "414080 129 129 1 1" "414080 129 129 2 2"; do
  echo "== $cfg ILF"; /tmp/gemm_nt_bench $cfg 30 | grep -v "bad element"
  echo "== $cfg plain"; PFN_NO_NT_ILF=1 /tmp/gemm_nt_bench $cfg 30 | grep -v "bad element"
done; done
