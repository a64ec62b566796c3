#!/bin/bash
# Round profile (run on the GPU box through gpurun): rocprofv3 kernel-trace stats of the default bench command, plus
# two separate PMC passes (FETCH_SIZE / WRITE_SIZE) on a short eager run.  Summaries land in gpurun_out/prof_round/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_round; rm -rf $O /tmp/pr; mkdir -p $O /tmp/pr
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pr/trace -o bench -- python $R/bench.py --no-cpu-baseline $BENCH_ARGS > $O/bench_under_rocprof.json 2> /tmp/pr/trace.err
find /tmp/pr/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find /tmp/pr/trace -name "*domain_stats.csv" -exec cp {} $O/domain_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -f csv -d /tmp/pr/$c -o pmc -- python $R/bench.py --no-cpu-baseline --no-graph --steps 3 --warmup 1 --profile-steps 0 $BENCH_ARGS > /tmp/pr/$c.out 2> /tmp/pr/$c.err
  F=$(find /tmp/pr/$c -name "*counter_collection.csv" | head -1)
  python - "$F" "$c" > $O/pmc_$c.txt <<'PY'
import csv, sys, collections
path, ctr = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
with open(path) as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") != ctr: continue
        k = row["Kernel_Name"].split("(")[0][:60]
        agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
print(f"# {ctr}: per-kernel launches, total counter value, value per launch (raw counter units as reported by rocprofv3)")
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:60s} n={n:5d} total={v:16.1f} per_launch={v / n:14.1f}")
PY
done
ls -la $O
