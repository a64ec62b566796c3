#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/k; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -f csv -d /tmp/kt -o t -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --profile-steps 0 > $O/bench.json 2> $O/bench.err
F=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/ktimeline_csv.py $F 3 > $O/timeline.txt 2>&1
python $R/tools/ktimeline_csv.py $F 5 > $O/timeline2.txt 2>&1
cd $R && bash tools/ubench/run_gemm_nt.sh > $O/ubench_nt.txt 2>&1
