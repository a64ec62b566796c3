#!/bin/bash
# gemm_nt at the metric's M = 15,104 (one row tile per wave) with the A refills re-reading ONE cache line (NOREFILL: wrong results,
# timing only): how much of a 4-term launch is the row-per-lane gather of the operand fragments?
R=$GRAFT_REPO_ROOT; S=/tmp/exp_src; rm -rf $S; mkdir -p $S; cp -r $R/poweflownet_amd $R/include $S/; C=$S/poweflownet_amd/csrc; bash $R/tools/ubench/apply_experiments.sh $C; cd $C   # (the switches live in tools/ubench/*.patch.txt)
for v in ${NT_VARIANTS:-BASE NOREFILL}; do
  d=/tmp/exp_$(echo $v | tr -d ' -' ); mkdir -p $d
  for f in $(ls *.hip | sed "s/.hip//"); do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DPFN_EXP_$v -c $f.hip -o $d/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o $d/libpfn_hip.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $R/tools/ubench/gemm_nt_bench.hip -L$d -lpfn_hip -Wl,-rpath,$d -o $d/bench || exit 1
  echo "== $v"
  for cfg in "15104 129 129 1 1" "15104 129 129 2 1" "15104 129 129 4 1" "241664 129 129 4 1"; do PFN_NT_TINY_MAX_TILES=0 $d/bench $cfg 50 | grep -v "bad element"; done
done
