#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4large; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_json or odd_shapes or g2_edge or g3_tag or other_edge_feature or seeded or wide_k6 or k6_big" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 600 python bench.py --config large --no-cpu-baseline --no-live-traffic --no-dp-overhead --steps 20 --warmup 5 > $O/bench_large.json 2> $O/bench_large.err
python - $O/bench_large.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d.get("median_ms_per_step"), d["value"], d["step_mfma_frac"])
for k,v in sorted(d.get("kernels",{}).items(), key=lambda kv:-kv[1]["ms_per_step"]): print("  %-20s %5.1f x %8.2f us = %7.4f ms  %s"%(k,v["launches_per_step"],v["avg_us"],v["ms_per_step"], v.get("frac")))
PY
