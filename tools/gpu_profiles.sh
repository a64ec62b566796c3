#!/bin/bash
# the five tracked profile sets of a round, one after the other (gpurun -- tools/gpu_profiles.sh rNN)
R=${1:-r04}
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
BENCH_ARGS="" bash tools/profile_round.sh ${R}_case118_b128_train
BENCH_ARGS="--mode infer --batch 2048" bash tools/profile_round.sh ${R}_case118_b2048_infer
BENCH_ARGS="--case 6470rte --batch 64 --steps 10 --warmup 3" bash tools/profile_round.sh ${R}_case6470_b64_train
BENCH_ARGS="--config wide --case 6470rte --batch 64 --steps 6 --warmup 2" bash tools/profile_round.sh ${R}_case6470_b64_wide_train
BENCH_ARGS="--case 6470rte --batch 64 --steps 10 --warmup 3 --hub-frac 0.2" bash tools/profile_round.sh ${R}_case6470_b64_hub_train
BENCH_ARGS="--config large --steps 20 --warmup 5" bash tools/profile_round.sh ${R}_case118_b128_large_train
