cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
run() { python bench.py $1 --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --profile-steps 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k=j['kernels']; print('$2', 'ms_per_step', j['ms_per_step'], 'median', j.get('median_ms_per_step'), 'min', j.get('min_ms_per_step'), {n:(v.get('avg_us'), v.get('frac')) for n,v in k.items()})
" >> $O/bench.txt; }
run "--case 118v2 --batch 2048 --mode infer --steps 40 --warmup 5" cfg3
run "--case 118v2 --batch 2048 --mode infer --steps 40 --warmup 5" cfg3
run "--steps 20 --warmup 5" cfg2_driverflags
run "" cfg2
run "--case 6470rte --batch 64 --mode train --steps 12 --warmup 3" cfg4
