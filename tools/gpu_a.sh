#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/t; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pr/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 3 --warmup 1 > $O/prof.out 2> $O/prof.err
find /tmp/pr/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/cold_stats.csv
