#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c3; mkdir -p $O
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-dp-overhead --no-other-configs --steps 200 --warmup 20 > $O/b2_$i.json 2> $O/b2_$i.err
python -c "
import json; d=json.loads(open('$O/b2_$i.json').read().strip().splitlines()[-1]); print('config2 run $i', d['ms_per_step'], d['median_ms_per_step'], d['value'])"
done
timeout 300 python bench.py --mode infer --batch 2048 --no-cpu-baseline --no-live-traffic > $O/b3.json 2> $O/b3.err
timeout 600 python bench.py --case 6470rte --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --no-dp-overhead > $O/b4.json 2> $O/b4.err
for f in b3 b4; do python -c "
import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['median_ms_per_step'], d['value'])
for k,v in sorted(d.get('kernels',{}).items(), key=lambda kv:-kv[1]['ms_per_step'])[:7]: print('  %-20s %5.1f x %8.2f us = %7.4f ms  %s'%(k,v['launches_per_step'],v['avg_us'],v['ms_per_step'], v.get('frac')))"; done
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_linear or g4_whole or g6 or config2_full or seeded or graphed_train_step_equals or inference_forward_equals" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
