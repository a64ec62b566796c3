#!/bin/bash
# A/B of the large-M GEMMs: fp32 through the bf16 matrix cores (opt-in: PFN_NT_BX_MIN_TILES=2) against the default fp32-MFMA kernels
#   gpurun --timeout 1500 -- bash tools/bx_check.sh      (results under gpurun_out/bx/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/bx; mkdir -p $O
for v in "PFN_NT_BX_MIN_TILES=2" ""; do
  t=fp32; [ -n "$v" ] && t=bx
  env $v python bench.py --no-cpu-baseline --no-live-traffic --mode infer --batch 2048 > $O/b3_$t.json 2> $O/b3_$t.err
  env $v python bench.py --no-cpu-baseline --no-live-traffic --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4_$t.json 2> $O/b4_$t.err
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bf16_split" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
