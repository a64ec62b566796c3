cd $GRAFT_REPO_ROOT
O=gpurun_out/s3; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_torch_ops.py -q -x > $O/pytest_torch_ops.log 2>&1; echo "exit $?" >> $O/pytest_torch_ops.log
B="python bench.py --case 6470rte --batch 64 --mode train --steps 12 --warmup 3 --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --profile-steps 3"
run() { env $1 $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k=j['kernels']; print('$2', 'ms_per_step', j['ms_per_step'], 'min', j.get('min_ms_per_step'), 'hops_fwd', k.get('fused_hops_fwd',{}).get('avg_us'), k.get('fused_hops_fwd',{}).get('frac'), 'hops_bwd', k.get('fused_hops_bwd',{}).get('avg_us'))
" >> $O/ab_time.txt; }
run X=0 pred; run X=0 pred
python tools/tmp/ab_hops.py > $O/ab_bits.txt 2>&1
cd poweflownet_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DBH_NOPRED -c edge.hip -o edge.o && make libpfn_hip.so > /dev/null 2>&1; cd ../..
run X=0 nopred; run X=0 nopred
run PFN_BIG_HOPS_V1=1 v1
