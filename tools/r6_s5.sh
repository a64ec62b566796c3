#!/bin/bash
# round 6, GPU session 5: the repaired loss-tail test; phase timestamps of gemm_nt at config-2 size
cd $GRAFT_REPO_ROOT
O=gpurun_out/s5; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_mse_tail.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
bash tools/ubench/run_gemm_nt_ts.sh > $O/nt_ts.txt 2>&1
bash tools/ubench/run_gemm_nt_ts2_small.sh > $O/nt_ts2_small.txt 2>&1
