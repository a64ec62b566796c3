#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/g; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt
python bench.py --no-cpu-baseline > $O/b2.json 2> $O/b2.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4.json 2> $O/b4.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters.txt 2>&1
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace -f csv -d /tmp/sq -o pmc -- python $R/bench.py --no-cpu-baseline --no-graph --steps 3 --warmup 1 --profile-steps 0 --case 6470rte --batch 64 > /tmp/sq.out 2> /tmp/sq.err
F=$(find /tmp/sq -name "*counter_collection.csv" | head -1)
python - "$F" > $R/$O/sq_6470.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][:50]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    n = max(cnt[k], 1)
    print(k, "launches", n, {c: round(v / n) for c, v in d.items()})
PY
tail -3 /tmp/sq.err >> $R/$O/sq_6470.txt
ls $R/$O
