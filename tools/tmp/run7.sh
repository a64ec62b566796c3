cd $GRAFT_REPO_ROOT
O=gpurun_out/s7; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --case 6470rte --batch 64 --mode train --steps 12 --warmup 3 --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --profile-steps 3"
run() { $B $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k=j['kernels']; print('$1', 'ms_per_step', j['ms_per_step'], 'min', j.get('min_ms_per_step'), 'hops_fwd', k.get('fused_hops_fwd',{}).get('avg_us'), k.get('fused_hops_fwd',{}).get('frac'), 'hops_bwd', k.get('fused_hops_bwd',{}).get('avg_us'))
" >> $O/ab.txt; }
python tools/tmp/save_out.py /tmp/new.pt > $O/save_new.log 2>&1
run new ""; run new ""; run new_hub "--hub-frac 0.2"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "config4 or big_graph or wide" > $O/pytest_big.log 2>&1; echo "exit $?" >> $O/pytest_big.log
cd poweflownet_amd/csrc && cp ../../tools/tmp/edge_old.hip.txt edge_old.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -c edge_old.hip -o edge.o && make libpfn_hip.so > /dev/null 2>&1; cd ../..
python tools/tmp/save_out.py /tmp/old.pt > $O/save_old.log 2>&1
run old ""; run old ""; run old_hub "--hub-frac 0.2"
python - > $O/bits.txt 2>&1 <<'PY'
import torch
a, b = torch.load("/tmp/new.pt"), torch.load("/tmp/old.pt")
for k in a:
    print(k, "out identical", torch.equal(a[k][0], b[k][0]), "grads identical", torch.equal(a[k][1], b[k][1]))
PY
