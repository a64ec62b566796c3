#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/o; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest.txt
for c in 512 1024 2048; do
PFN_EB_BLOCKS=$c python bench.py --no-cpu-baseline > $O/b2_$c.json 2> $O/b2_$c.err
PFN_EB_BLOCKS=$c python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4_$c.json 2> $O/b4_$c.err
done
