#!/bin/bash
# last-layer fusions (edge_fwd_out, ds_row): tests, A/B benches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
for v in fused nofuse; do unset PFN_NO_SEG_EA;
  if [ $v = nofuse ]; then export PFN_NO_SEG_EA=1; else unset PFN_NO_SEG_EA; fi
  python bench.py --no-cpu-baseline > $O/b2_$v.json 2> $O/b2_$v.err
  python bench.py --no-cpu-baseline --mode infer --batch 2048 > $O/b3_$v.json 2> $O/b3_$v.err
done
unset PFN_NO_SEG_EA
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pr -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /tmp/pr.out 2>/tmp/pr.err; cp /tmp/pr/*kernel_stats.csv $GRAFT_REPO_ROOT/$O/ 2>/dev/null
