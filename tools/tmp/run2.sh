cd $GRAFT_REPO_ROOT
O=gpurun_out/s2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python tools/tmp/ab_hops.py > $O/ab_bits.txt 2>&1
B="python bench.py --case 6470rte --batch 64 --mode train --steps 12 --warmup 3 --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --profile-steps 3"
for rep in 1 2; do
for v in "X=0" "PFN_BIG_HOPS_V1=1" "PFN_BIG_HOPS_WPG=3" "PFN_BIG_HOPS_WPG=8" "PFN_BIG_HOPS_WPG=2"; do
  env $v $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k=j['kernels']; print('$v', 'ms_per_step', j['ms_per_step'], 'min', j.get('min_ms_per_step'), 'hops_fwd', k.get('fused_hops_fwd',{}).get('avg_us'), k.get('fused_hops_fwd',{}).get('frac'), 'hops_bwd', k.get('fused_hops_bwd',{}).get('avg_us'))
" >> $O/ab_time.txt
done; done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "config4 or big_graph or wide" > $O/pytest_big.log 2>&1; echo "exit $?" >> $O/pytest_big.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --profile-steps 0 2>/dev/null | head -c 600 > $O/bench_ramp.txt
