cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
BENCH_ARGS="" bash tools/profile_round.sh r05_case118_b128_train
cd $GRAFT_REPO_ROOT
O=gpurun_out/bench_lines; mkdir -p $O
python bench.py > $O/r05_case118_b128_train_bench.json 2> $O/r05_case118_b128_train_bench.err
python bench.py --warmup 5 --steps 20 --no-cpu-baseline > $O/r05_case118_b128_train_bench_driver_flags.json 2> $O/df.err
python bench.py --loss masked_l2 --no-cpu-baseline --no-other-configs > $O/r05_case118_b128_train_masked_l2_bench.json 2> $O/ml2.err
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/final_pytest.log 2>&1; echo "exit $?" >> gpurun_out/final_pytest.log; tail -4 gpurun_out/final_pytest.log
