cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/final
( time timeout 1800 python -m pytest tests -m gpu -q ) > gpurun_out/final/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/final/pytest.log
cp gpurun_out/parity_report.json gpurun_out/final/ 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver_flags.json 2> gpurun_out/final/bench_driver_flags.err
bash tools/gpu_bench_lines.sh r05 > gpurun_out/final/bench_lines.log 2>&1
bash tools/gpu_profiles.sh r05 > gpurun_out/final/profiles.log 2>&1
