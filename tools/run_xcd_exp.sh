#!/bin/bash
# Experiment: hop_kernel and edge_fwd_kernel with an XCD-contiguous block order (-DPFN_EXP_XCD) against the product build:
# kernel times and fabric traffic (FETCH_SIZE / WRITE_SIZE) at case118v2 x 2048 inference and 6470rte x 64 training.
R=$GRAFT_REPO_ROOT; C=$R/poweflownet_amd/csrc; O=$R/gpurun_out/xcd; mkdir -p $O
cp $C/libpfn_hip.so /tmp/libpfn_product.so
cd $C; mkdir -p /tmp/xcd
for f in graph edge gemm gemm_nt front ea_seg model physics prof; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DPFN_EXP_XCD -c $f.hip -o /tmp/xcd/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/xcd/*.o -o /tmp/libpfn_xcd.so || exit 1
for v in product xcd; do
  cp /tmp/libpfn_$v.so $C/libpfn_hip.so
  BENCH_ARGS="--mode infer --batch 2048" bash $R/tools/profile_round.sh exp_${v}_c3 > /dev/null 2>&1
  BENCH_ARGS="--case 6470rte --batch 64 --steps 10 --warmup 3" bash $R/tools/profile_round.sh exp_${v}_c4 > /dev/null 2>&1
done
cp /tmp/libpfn_product.so $C/libpfn_hip.so
mv $R/gpurun_out/prof_round/exp_* $O/
