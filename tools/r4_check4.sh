#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c4; mkdir -p $O
timeout 600 python bench.py --case 6470rte --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --no-dp-overhead > $O/b4.json 2> $O/b4.err
python -c "
import json; d=json.loads(open('$O/b4.json').read().strip().splitlines()[-1]); print('b4', d['ms_per_step'], d['median_ms_per_step'], d['value'])
for k,v in sorted(d.get('kernels',{}).items(), key=lambda kv:-kv[1]['ms_per_step'])[:8]: print('  %-20s %5.1f x %8.2f us = %7.4f ms  %s'%(k,v['launches_per_step'],v['avg_us'],v['ms_per_step'], v.get('frac')))"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config4 and 16 or big_graph or k6_big" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
