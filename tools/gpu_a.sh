#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/t; mkdir -p $O
for w in 4 64; do
PFN_SEG_EA_PER_CU=$w python bench.py --no-cpu-baseline --mode infer --batch 2048 > $O/b3_seg$w.json 2> $O/b3_seg$w.err
PFN_SEG_EA_PER_CU=$w python bench.py --no-cpu-baseline --mode infer --batch 512 > $O/b3b_seg$w.json 2> $O/b3b_seg$w.err
done
