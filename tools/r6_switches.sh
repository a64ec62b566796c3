#!/bin/bash
# the parity tests with each of round 6's A/B switches thrown (the fallback paths must stay green)
cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/switches; mkdir -p $O
for sw in PFN_NO_SERPENTINE PFN_NO_NT_ILF PFN_NO_NT_PAIR; do
  env $sw=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/$sw.log 2>&1; echo "$sw exit $?" >> $O/summary.txt; tail -1 $O/$sw.log >> $O/summary.txt
done
