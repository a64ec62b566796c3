#!/bin/bash
# GPU session B (round 2): full -m gpu suite on the deferred weight-gradient path, benches of configs 2/3/4.
cd $GRAFT_REPO_ROOT
O=gpurun_out/b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2000 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/pytest.txt
python bench.py --no-cpu-baseline > $O/b2.json 2> $O/b2.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4.json 2> $O/b4.err
python bench.py --no-cpu-baseline --mode infer --batch 2048 --steps 20 --warmup 5 > $O/b3.json 2> $O/b3.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --config wide --steps 6 --warmup 2 > $O/b4w.json 2> $O/b4w.err
for nb in 128 384 512; do PFN_TN_BLOCKS=$nb python bench.py --no-cpu-baseline --profile-steps 3 > $O/b2_tn$nb.json 2> $O/b2_tn$nb.err; done
PFN_TN_BLOCKS=512 python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4_tn512.json 2> $O/b4_tn512.err
ls -la $O
