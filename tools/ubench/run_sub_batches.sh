cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0
for b in 2048 1024 683 512 256; do python bench.py --mode infer --batch $b --no-cpu-baseline --no-live-traffic --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b ms', d['ms_per_step'], 'graphs/s', d['value'], {k:(v['launches_per_step'],v['avg_us']) for k,v in d['kernels'].items()})"; done
for b in 64 32 16; do python bench.py --case 6470rte --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('6470 batch $b ms', d['ms_per_step'], 'graphs/s', d['value'])"; done
