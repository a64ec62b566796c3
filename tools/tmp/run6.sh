cd $GRAFT_REPO_ROOT
O=gpurun_out/s6; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "quarter_by_quarter or config4 or config3 or inference" > $O/pytest_seq.log 2>&1; echo "exit $?" >> $O/pytest_seq.log
run() { env $1 python bench.py $2 --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --profile-steps 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k=j['kernels']; print('$1', '$3', 'ms_per_step', j['ms_per_step'], 'min', j.get('min_ms_per_step'), 'gemm_nt', k.get('gemm_nt',{}).get('avg_us'), k.get('gemm_nt',{}).get('frac'))
" >> $O/ab.txt; }
for rep in 1 2; do
for v in X=0 PFN_NO_NT_SEQ=1; do
run $v "--case 118v2 --batch 2048 --mode infer --steps 40 --warmup 5" cfg3
run $v "--case 6470rte --batch 64 --mode train --steps 12 --warmup 3" cfg4
done; done
run X=0 "" cfg2
