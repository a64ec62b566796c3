#!/bin/bash
# gemm_nt at the metric's M = 15,104 (one row tile per wave) with the A refills re-reading ONE cache line (NOREFILL: wrong results,
# timing only): how much of a 4-term launch is the row-per-lane gather of the operand fragments?
R=$GRAFT_REPO_ROOT; C=$R/poweflownet_amd/csrc; cd $C
for v in ${NT_VARIANTS:-BASE NOREFILL}; do
  d=/tmp/exp_$(echo $v | tr -d ' -' ); mkdir -p $d
  for f in graph edge gemm gemm_nt front ea_seg seg_lin_hops model physics prof; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DPFN_EXP_$v -c $f.hip -o $d/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o $d/libpfn_hip.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $R/tools/ubench/gemm_nt_bench.hip -L$d -lpfn_hip -Wl,-rpath,$d -o $d/bench || exit 1
  echo "== $v"
  for cfg in "15104 129 129 1 1" "15104 129 129 2 1" "15104 129 129 4 1" "241664 129 129 4 1"; do PFN_NT_TINY_MAX_TILES=0 $d/bench $cfg 50 | grep -v "bad element"; done
done
