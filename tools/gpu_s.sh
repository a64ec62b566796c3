#!/bin/bash
# config 3 (inference, 2048 graphs) with / without the graph-resident kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/s; mkdir -p $O
for v in all notag noea none; do
  unset PFN_NO_SEG_TAG PFN_NO_SEG_EA
  [ $v = notag ] && export PFN_NO_SEG_TAG=1
  [ $v = noea ] && export PFN_NO_SEG_EA=1
  [ $v = none ] && export PFN_NO_SEG_TAG=1 PFN_NO_SEG_EA=1
  python bench.py --no-cpu-baseline --mode infer --batch 2048 > $O/b3_$v.json 2> $O/b3_$v.err
  python bench.py --no-cpu-baseline --batch 512 > $O/b2x4_$v.json 2> $O/b2x4_$v.err
done
