#!/bin/bash
# A/B of the layer-0 paths beyond the latency regime: P | Q formed from x0 (default) against stored rows (PFN_NO_L0_FLY=1), the
# one-row-per-thread fronts (default) against the block kernel (PFN_FRONT_NO_THREAD_ROWS=1): tests, then configs 3 and 4.
#   gpurun --timeout 1500 -- bash tools/fly_check.sh [tests-only]      (results under gpurun_out/fly/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/fly; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "first_layer_pq or fused_front or inference_forward_equals or config4" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
[ "$1" = "tests-only" ] && exit 0
for v in "" "PFN_FRONT_NO_THREAD_ROWS=1" "PFN_NO_L0_FLY=1"; do
  t=default; [ "$v" = "PFN_FRONT_NO_THREAD_ROWS=1" ] && t=blockfront; [ "$v" = "PFN_NO_L0_FLY=1" ] && t=stored
  env $v python bench.py --no-cpu-baseline --no-live-traffic --mode infer --batch 2048 > $O/b3_$t.json 2> $O/b3_$t.err
  env $v python bench.py --no-cpu-baseline --no-live-traffic --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4_$t.json 2> $O/b4_$t.err
done
