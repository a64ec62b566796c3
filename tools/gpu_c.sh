#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt
python bench.py --no-cpu-baseline > $O/b2.json 2> $O/b2.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4.json 2> $O/b4.err
for nb in 384 512; do
  PFN_TN_BLOCKS=$nb python bench.py --no-cpu-baseline --profile-steps 3 > $O/b2_tn$nb.json 2> $O/b2_tn$nb.err
  PFN_TN_BLOCKS=$nb python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4_tn$nb.json 2> $O/b4_tn$nb.err
done
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --config wide --steps 6 --warmup 2 > $O/b4w.json 2> $O/b4w.err
python tools/exp_two_streams.py 128 2 > $O/two.txt 2>&1
python tools/exp_two_streams.py 128 4 >> $O/two.txt 2>&1
python tools/exp_two_streams.py 256 2 >> $O/two.txt 2>&1
ls $O
