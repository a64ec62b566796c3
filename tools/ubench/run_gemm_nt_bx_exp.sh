#!/bin/bash
# gemm_nt_bx_kernel with one ingredient removed at a time (compile-time switches; results are WRONG by design, only the timings
# mean something): what bounds it?   bash tools/ubench/run_gemm_nt_bx_exp.sh
R=$GRAFT_REPO_ROOT; C=$R/poweflownet_amd/csrc; cd $C
for v in BASE BX_NOLDS BX_NOREFILL BX_NOMFMA "BX_NOLDS -DPFN_EXP_BX_NOREFILL" "BX_NOMFMA -DPFN_EXP_BX_NOLDS" "BX_NOMFMA -DPFN_EXP_BX_NOREFILL" "BX_NOMFMA -DPFN_EXP_BX_NOLDS -DPFN_EXP_BX_NOREFILL"; do
  d=/tmp/exp_$(echo $v | tr -d ' -' | cut -c1-60); mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DPFN_EXP_$v -c gemm_nt.hip -o $d/gemm_nt.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC graph.o edge.o gemm.o $d/gemm_nt.o front.o ea_seg.o model.o physics.o prof.o -o $d/libpfn_hip.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $R/tools/ubench/gemm_nt_bench.hip -L$d -lpfn_hip -Wl,-rpath,$d -o $d/bench || exit 1
  echo "== $v"
  for cfg in "241664 129 129 4 1"; do PFN_NT_BX_MIN_TILES=2 $d/bench $cfg 20 | grep -v "bad element"; done
done
