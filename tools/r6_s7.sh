#!/bin/bash
# round 6, GPU session 7: per-launch timelines of one replayed step at configs 3 and 4 (which gemm_nt launch costs what)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s7; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof3; mkdir -p /tmp/prof3
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof3 -o t -- python $GRAFT_REPO_ROOT/bench.py --mode infer --batch 2048 --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-live-traffic --no-other-configs > /tmp/prof3/bench.out 2>&1
python $GRAFT_REPO_ROOT/tools/ktimeline.py /tmp/prof3/t_results.db 40 > $GRAFT_REPO_ROOT/$O/timeline_c3.txt 2>&1
rm -rf /tmp/prof4; mkdir -p /tmp/prof4
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof4 -o t -- python $GRAFT_REPO_ROOT/bench.py --case 6470rte --batch 64 --steps 3 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-live-traffic --no-other-configs > /tmp/prof4/bench.out 2>&1
python $GRAFT_REPO_ROOT/tools/ktimeline.py /tmp/prof4/t_results.db 70 > $GRAFT_REPO_ROOT/$O/timeline_c4.txt 2>&1
ls /tmp/prof3 /tmp/prof4 > $GRAFT_REPO_ROOT/$O/ls.txt
