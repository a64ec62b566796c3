#!/bin/bash
# usage: tools/prof_eager.sh <tag> <like-pattern> [ENV=val ...] -- runs the eager bench under rocprofv3 --kernel-trace on the
# GPU box and leaves only a text summary in gpurun_out/<tag>.txt (the rocpd database is too big to ship back).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; like=$2; shift; shift
rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag $R/gpurun_out
env "$@" timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o t -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-graph $BENCH_ARGS > /tmp/prof_$tag/bench.out 2>&1
python $R/tools/kstats.py /tmp/prof_$tag/t_results.db "$like" 11 > $R/gpurun_out/$tag.txt 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/prof_$tag/bench.out | head -1 >> $R/gpurun_out/$tag.txt
if [ -n "$KSEQ" ]; then python $R/tools/kseq.py /tmp/prof_$tag/t_results.db "$KSEQ" ${KSEQ_N:-25} >> $R/gpurun_out/$tag.txt; fi
