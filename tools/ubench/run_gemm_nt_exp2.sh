#!/bin/bash
R=$GRAFT_REPO_ROOT; S=/tmp/exp_src; rm -rf $S; mkdir -p $S; cp -r $R/poweflownet_amd $R/include $S/; C=$S/poweflownet_amd/csrc; bash $R/tools/ubench/apply_experiments.sh $C; cd $C   # (the switches live in tools/ubench/*.patch.txt)
ALL="-DPFN_EXP_NOREFILL -DPFN_EXP_NOLDS -DPFN_EXP_NOSTORE"
i=0
for v in "" "$ALL" "$ALL -DPFN_EXP_NOWAIT" "$ALL -DPFN_EXP_NOSCHEDBAR" "$ALL -DPFN_EXP_NOWAIT -DPFN_EXP_NOSCHEDBAR"; do
  i=$((i+1)); d=/tmp/exp2_$i; mkdir -p $d
  for f in $(ls *.hip | sed "s/.hip//"); do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $v -c $f.hip -o $d/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o $d/libpfn_hip.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $R/tools/ubench/gemm_nt_bench.hip -L$d -lpfn_hip -Wl,-rpath,$d -o $d/bench || exit 1
  for ct in 1 2; do
    echo "== [$v] PFN_NT_CT=$ct"
    for cfg in "414080 129 129 1 1" "414080 129 129 4 1" "414080 128 128 4 1"; do PFN_NT_CT=$ct $d/bench $cfg 20 | grep -v "bad element"; done
  done
done
