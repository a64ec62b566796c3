#!/bin/bash
# kernel timeline of one REPLAYED config-2 step (rocprofv3 --kernel-trace): durations and the gaps between the chain's kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; rm -rf /tmp/tl; mkdir -p /tmp/tl $R/gpurun_out
timeout 300 rocprofv3 --kernel-trace -f csv -d /tmp/tl -o t -- python $R/bench.py --no-cpu-baseline --no-live-traffic --no-dp-overhead --no-other-configs --profile-steps 0 --steps 30 --warmup 5 > /tmp/tl/out.json 2> /tmp/tl/err.txt
F=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python $R/tools/ktimeline_csv.py $F 40 | tee $R/gpurun_out/r04_config2_step_timeline.txt | tail -40
