#!/bin/bash
# One GPU session: the -m gpu tests, then the benches of configs 2 (also with the generic kernels forced), 3, 4 and 4-hub.
#   gpurun --timeout 2400 -- bash tools/gpu_check.sh     (results under gpurun_out/t/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/t; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
python bench.py --no-cpu-baseline > $O/b2.json 2> $O/b2.err
PFN_NO_SEG_EA=1 python bench.py --no-cpu-baseline > $O/b2_generic.json 2> $O/b2_generic.err
python bench.py --no-cpu-baseline --mode infer --batch 2048 > $O/b3.json 2> $O/b3.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4.json 2> $O/b4.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 --hub-frac 0.2 > $O/b4h.json 2> $O/b4h.err
