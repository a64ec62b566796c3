#!/bin/bash
# usage: tools/prof_graph.sh <tag> <n-last-kernels>: kernel timeline of the hipGraph-replayed default bench step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; n=${2:-90}
rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag $R/gpurun_out
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 $BENCH_ARGS > /tmp/prof_$tag/bench.out 2>&1
python $R/tools/ktimeline.py /tmp/prof_$tag/t_results.db $n > $R/gpurun_out/$tag.txt 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/prof_$tag/bench.out | head -1 >> $R/gpurun_out/$tag.txt
