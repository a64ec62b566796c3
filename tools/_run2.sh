cd $GRAFT_REPO_ROOT
O=gpurun_out/m2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_mse_tail.py tests/test_torch_ops.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -30 $O/pytest.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --warmup 5 --steps 20 --loss masked_l2 > $O/b2m_tail_$i.json 2> $O/b2m_tail_$i.err
PFN_NO_MSE_TAIL=1 python bench.py --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --warmup 5 --steps 20 --loss masked_l2 > $O/b2m_plain_$i.json 2> $O/b2m_plain_$i.err
done
python bench.py --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --warmup 5 --steps 20 > $O/b2_tail.json 2> $O/b2_tail.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/m2/b2*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d.get('median_ms_per_step'), {k:v['avg_us'] for k,v in d['kernels'].items() if 'ea_seg_bwd' in k or 'lin_out' in k or 'front' in k})
    except Exception as e: print(f, 'ERR', e)
P
