#!/bin/bash
# The UNPROFILED bench.py line of every tracked workload (full lines: live PMC traffic, cpu_baseline, DP overhead), as committed
# under profiles/<round>_<tag>_bench.json.   gpurun --timeout 2400 -- bash tools/gpu_bench_lines.sh r03
R=${1:-r04}
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/bench_lines; mkdir -p $O
python bench.py > $O/${R}_case118_b128_train_bench.json 2> $O/${R}_case118_b128_train_bench.err
python bench.py --mode infer --batch 2048 > $O/${R}_case118_b2048_infer_bench.json 2> $O/${R}_case118_b2048_infer_bench.err
python bench.py --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/${R}_case6470_b64_train_bench.json 2> $O/${R}_case6470_b64_train_bench.err
python bench.py --config wide --case 6470rte --batch 64 --steps 6 --warmup 2 --no-cpu-baseline > $O/${R}_case6470_b64_wide_train_bench.json 2> $O/${R}_case6470_b64_wide_train_bench.err
python bench.py --case 6470rte --batch 64 --steps 10 --warmup 3 --hub-frac 0.2 --no-cpu-baseline > $O/${R}_case6470_b64_hub_train_bench.json 2> $O/${R}_case6470_b64_hub_train_bench.err
python bench.py --mode infer --batch 1 > $O/${R}_case118_b1_infer_latency_bench.json 2> $O/${R}_case118_b1_infer_latency_bench.err
python bench.py --config large --steps 20 --warmup 5 --no-other-configs > $O/${R}_case118_b128_large_train_bench.json 2> $O/${R}_case118_b128_large_train_bench.err
