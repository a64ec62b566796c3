#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/j; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt
python bench.py > $O/b2_full.json 2> $O/b2_full.err
BENCH_ARGS="" bash tools/profile_round.sh r02_case118_b128_train > $O/prof1.log 2>&1
BENCH_ARGS="--mode infer --batch 2048 --steps 20 --warmup 5" bash tools/profile_round.sh r02_case118_b2048_infer > $O/prof2.log 2>&1
BENCH_ARGS="--case 6470rte --batch 64 --steps 6 --warmup 2" bash tools/profile_round.sh r02_case6470_b64_train > $O/prof3.log 2>&1
BENCH_ARGS="--case 6470rte --batch 64 --config wide --steps 4 --warmup 2" bash tools/profile_round.sh r02_case6470_b64_wide_train > $O/prof4.log 2>&1
BENCH_ARGS="--case 6470rte --batch 64 --hub-frac 0.2 --steps 6 --warmup 2" bash tools/profile_round.sh r02_case6470_b64_hub_train > $O/prof5.log 2>&1
ls gpurun_out/prof_round | grep r02_ | head -50
