#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_linear_hops or more_edges_than or g4_whole or train_mode_matches or dropout_mask_stat" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for v in fused; do
  timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-dp-overhead --steps 200 --warmup 20 > $O/bench_$v.json 2> $O/bench_$v.err
  python - $O/bench_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d.get("median_ms_per_step"), d["value"])
for k,v in sorted(d.get("kernels",{}).items(), key=lambda kv:-kv[1]["ms_per_step"]): print("  %-20s %5.1f x %8.2f us = %7.4f ms"%(k,v["launches_per_step"],v["avg_us"],v["ms_per_step"]))
PY
done
bash tools/ubench/run_slh_ts.sh 2>&1 | grep -v amdgpu.ids | head -14
