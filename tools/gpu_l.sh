#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/l; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt
python bench.py --no-cpu-baseline > $O/b2.json 2> $O/b2.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4.json 2> $O/b4.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --config wide --steps 6 --warmup 2 > $O/b4w.json 2> $O/b4w.err
