#!/bin/bash
# round 6, GPU session 3 (re-entry baseline): the whole -m gpu suite on the HEAD tree, the driver-flag bench line, the config-2 profile set
cd $GRAFT_REPO_ROOT
O=gpurun_out/s3; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/b2_driver.json 2> $O/b2_driver.err
bash tools/profile_round.sh r06_case118_b128_train > $O/prof.log 2>&1
