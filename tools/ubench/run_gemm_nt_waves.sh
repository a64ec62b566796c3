#!/bin/bash
export ASYNC_CHECK=${GRAFT_REPO_ROOT:-/root/repo}/tools/check_async_fragments.py   # (csrc/Makefile checks the ISA of the async-fragment objects it links)
# experiment: two waves per SIMD (512 threads) against three (768; PFN_EXP_NT_THREADS) in the stationary gemm_nt, K = N = 128 (the
# shape that fits 168 registers without spills), in SHADER CYCLES (the harness runs power-capped: times mean nothing).
# pipe cycles a SIMD needs: 12940 tiles x 4 quarters x terms x 64 MFMAs x 64 / 1024 = 207,040 x terms
R=$GRAFT_REPO_ROOT
for T in 512 768; do
  d=/tmp/exp_w$T; rm -rf $d; mkdir -p $d; cp -r $R/poweflownet_amd $R/include $d/; bash $R/tools/ubench/apply_experiments.sh $d/poweflownet_amd/csrc
  ( cd $d/poweflownet_amd/csrc && rm -f *.o libpfn_hip.so && make -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DNT_EXP_TS2 -DPFN_EXP_NT_THREADS=$T $NT_EXTRA" > /dev/null ) || exit 1
  cd $R/tools/ubench
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DNT_EXP_TS2 gemm_nt_bench.hip -L$d/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$d/poweflownet_amd/csrc -o $d/bench || exit 1
  for ct in 2 1; do
    echo "== $T threads, PFN_NT_CT=$ct"
    for cfg in "414080 128 128 1 1" "414080 128 128 4 1"; do PFN_NT_CT=$ct PFN_NT_TINY_MAX_TILES=0 $d/bench $cfg 10 | grep "TFLOP\|waves:\|wave slot\|percentiles" | cut -c1-300; done
  done
done
