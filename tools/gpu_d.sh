#!/bin/bash
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
BENCH_ARGS="" bash tools/profile_round.sh r02a_case118_b128_train > gpurun_out/prof_d1.log 2>&1
BENCH_ARGS="--case 6470rte --batch 64 --steps 6 --warmup 2" bash tools/profile_round.sh r02a_case6470_b64_train > gpurun_out/prof_d2.log 2>&1
ls gpurun_out/prof_round
