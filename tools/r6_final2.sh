#!/bin/bash
# round 6: the whole -m gpu suite + smoke + the driver-flag bench line on the final tree (XCD-paired slices)
cd $GRAFT_REPO_ROOT
O=gpurun_out/final2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
cp gpurun_out/parity_report.json $O/parity_report.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/b2_driver.json 2> $O/b2_driver.err
