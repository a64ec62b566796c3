#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4hub; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config4 or big_graph or k6_big" -rs > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
for v in plain hub; do
  H=""; [ $v = hub ] && H="--hub-frac 0.2"
  timeout 600 python bench.py --case 6470rte --batch 64 --steps 10 --warmup 3 $H --no-cpu-baseline --no-live-traffic --no-dp-overhead > $O/bench_$v.json 2> $O/bench_$v.err
  python - $O/bench_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d.get("median_ms_per_step"), d["value"], d["step_mfma_frac"])
for k,v in sorted(d.get("kernels",{}).items(), key=lambda kv:-kv[1]["ms_per_step"])[:9]: print("  %-20s %5.1f x %8.2f us = %7.4f ms  %s"%(k,v["launches_per_step"],v["avg_us"],v["ms_per_step"], v.get("frac")))
PY
done
