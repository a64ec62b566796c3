#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/m; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt
python bench.py --no-cpu-baseline > $O/b2.json 2> $O/b2.err
PFN_NO_FUSED_EDGE=1 python bench.py --no-cpu-baseline > $O/b2_noedge.json 2> $O/b2_noedge.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4.json 2> $O/b4.err
python bench.py --no-cpu-baseline --case 14 --batch 32 > $O/b1.json 2> $O/b1.err
