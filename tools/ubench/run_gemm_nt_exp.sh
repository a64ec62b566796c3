#!/bin/bash
# gemm_nt at large M with one ingredient removed at a time (compile-time switches in gemm_nt.hip; results are WRONG by design,
# only the timings mean something): what bounds the multiply phase?
R=$GRAFT_REPO_ROOT; C=$R/poweflownet_amd/csrc; cd $C
for v in BASE NOREFILL NOLDS NOSTORE "NOREFILL -DPFN_EXP_NOLDS" "NOREFILL -DPFN_EXP_NOLDS -DPFN_EXP_NOSTORE"; do
  d=/tmp/exp_$(echo $v | tr -d ' -' ); mkdir -p $d
  for f in graph edge gemm gemm_nt front model physics prof; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DPFN_EXP_$v -c $f.hip -o $d/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o $d/libpfn_hip.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $R/tools/ubench/gemm_nt_bench.hip -L$d -lpfn_hip -Wl,-rpath,$d -o $d/bench || exit 1
  echo "== $v"
  for cfg in "414080 129 129 1 1" "414080 129 129 2 2" "414080 129 129 4 1" "414080 128 128 4 1"; do $d/bench $cfg 20 | grep -v "bad element"; done
done
