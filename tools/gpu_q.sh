#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/q; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for x in 0 1; do
PFN_XCD_MAP=$x python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4_x$x.json 2> $O/b4_x$x.err
PFN_XCD_MAP=$x python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 --hub-frac 0.2 > $O/b4h_x$x.json 2> $O/b4h_x$x.err
done
