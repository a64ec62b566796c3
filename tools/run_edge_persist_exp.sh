#!/bin/bash
# The persistent generic forward edge walk: workgroups per CU (PFN_EDGE_FWD_BPC; 100000 = one workgroup per 256 items, the old shape)
#   gpurun --timeout 1200 -- bash tools/run_edge_persist_exp.sh
cd $GRAFT_REPO_ROOT; O=gpurun_out/edge_persist; mkdir -p $O
for b in 0 4 7 10 14 100000; do
  v=""; [ $b != 0 ] && v="PFN_EDGE_FWD_BPC=$b"
  env $v python bench.py --no-cpu-baseline --no-live-traffic --no-dp-overhead --case 6470rte --batch 64 --steps 6 --warmup 2 > $O/b4_$b.json 2> $O/b4_$b.err
  env $v python bench.py --no-cpu-baseline --no-live-traffic --no-dp-overhead > $O/b2_$b.json 2> $O/b2_$b.err
done
python - <<'PY'
import json
for c in ("b4", "b2"):
    for b in (0, 4, 7, 10, 14, 100000):
        try:
            d = json.loads(open(f"gpurun_out/edge_persist/{c}_{b}.json").read().strip().splitlines()[-1])
            print(c, b, d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items() if k == "edge_fwd"})
        except Exception as e:
            print(c, b, "ERR", e)
PY
