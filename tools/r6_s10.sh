cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/ab_ref.sh 2 > gpurun_out/ab_xcd.txt 2>&1
python bench.py --no-cpu-baseline --no-other-configs --no-dp-overhead > gpurun_out/b2_xcd.json 2> gpurun_out/b2_xcd.err
python bench.py --mode infer --batch 2048 --no-cpu-baseline --no-other-configs > gpurun_out/b3_xcd.json 2> gpurun_out/b3_xcd.err
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "interleaved_flush or config3_size or config2_full or fused_linear_hops" > gpurun_out/xcd_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/xcd_pytest.log
