#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_front_and_first or fused_linear or g4_whole or more_edges_than or float_mask or inference_forward_equals or train_mode_matches" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-live-traffic --no-dp-overhead --no-other-configs --steps 200 --warmup 20 > $O/b2_$i.json 2> $O/b2_$i.err
python -c "
import json; d=json.loads(open('$O/b2_$i.json').read().strip().splitlines()[-1]); print('config2 run $i', d['ms_per_step'], d['median_ms_per_step'], d['value'])
for k,v in sorted(d.get('kernels',{}).items(), key=lambda kv:-kv[1]['ms_per_step']): print('  %-20s %5.1f x %8.2f us = %7.4f ms'%(k,v['launches_per_step'],v['avg_us'],v['ms_per_step']))" | head -$((i==1?16:1))
done
