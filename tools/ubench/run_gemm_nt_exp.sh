#!/bin/bash
# gemm_nt at large M with one ingredient removed at a time (compile-time switches in gemm_nt.hip; results are WRONG by design,
# only the timings mean something): what bounds the multiply phase?
R=$GRAFT_REPO_ROOT; S=/tmp/exp_src; rm -rf $S; mkdir -p $S; cp -r $R/poweflownet_amd $R/include $S/; C=$S/poweflownet_amd/csrc; bash $R/tools/ubench/apply_experiments.sh $C; cd $C   # (the switches live in tools/ubench/*.patch.txt)
for v in BASE NOREFILL NOLDS NOSTORE "NOREFILL -DPFN_EXP_NOLDS" "NOREFILL -DPFN_EXP_NOLDS -DPFN_EXP_NOSTORE"; do
  d=/tmp/exp_$(echo $v | tr -d ' -' ); mkdir -p $d
  for f in $(ls *.hip | sed "s/.hip//"); do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DPFN_EXP_$v -c $f.hip -o $d/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o $d/libpfn_hip.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $R/tools/ubench/gemm_nt_bench.hip -L$d -lpfn_hip -Wl,-rpath,$d -o $d/bench || exit 1
  echo "== $v"
  for cfg in "414080 129 129 1 1" "414080 129 129 2 2" "414080 129 129 4 1" "414080 128 128 4 1"; do $d/bench $cfg 20 | grep -v "bad element"; done
done
