#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/t; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
python bench.py --no-cpu-baseline > $O/b2.json 2> $O/b2.err
python bench.py --no-cpu-baseline --mode infer --batch 2048 > $O/b3.json 2> $O/b3.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4.json 2> $O/b4.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 --hub-frac 0.2 > $O/b4h.json 2> $O/b4h.err
