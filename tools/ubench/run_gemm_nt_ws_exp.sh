#!/bin/bash
# The weight-streaming gemm_nt with one ingredient removed at a time (compile-time switches; results are WRONG by design, only the
# timings mean something): what bounds it at M = 414,080?
R=$GRAFT_REPO_ROOT; S=/tmp/exp_src; rm -rf $S; mkdir -p $S; cp -r $R/poweflownet_amd $R/include $S/; C=$S/poweflownet_amd/csrc; bash $R/tools/ubench/apply_experiments.sh $C; cd $C   # (the switches live in tools/ubench/*.patch.txt)
for v in BASE NOREFILL NOSTORE NOLDS NOWAIT NOSCHEDBAR WS_NOBARRIER WS_NODMA "NOREFILL -DPFN_EXP_NOSTORE -DPFN_EXP_WS_NODMA -DPFN_EXP_WS_NOBARRIER" "NOREFILL -DPFN_EXP_NOSTORE -DPFN_EXP_WS_NODMA -DPFN_EXP_WS_NOBARRIER -DPFN_EXP_NOWAIT -DPFN_EXP_NOSCHEDBAR"; do
  d=/tmp/exp_$(echo $v | tr -d ' -' | cut -c1-40); mkdir -p $d
  for f in $(ls *.hip | sed "s/.hip//"); do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DPFN_EXP_$v -c $f.hip -o $d/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o $d/libpfn_hip.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $R/tools/ubench/gemm_nt_bench.hip -L$d -lpfn_hip -Wl,-rpath,$d -o $d/bench || exit 1
  echo "== $v"
  for cfg in "414080 129 129 4 1" "414080 129 129 1 1"; do $d/bench $cfg 20 | grep -v "bad element"; done
done
