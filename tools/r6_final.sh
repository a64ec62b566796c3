#!/bin/bash
# round 6: the whole -m gpu suite, the driver-flag bench line and the six tracked profile sets on ONE tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
cp gpurun_out/parity_report.json $O/parity_report.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/b2_driver.json 2> $O/b2_driver.err
bash tools/gpu_profiles.sh r06 > $O/profiles.log 2>&1
