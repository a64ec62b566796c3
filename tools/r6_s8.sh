#!/bin/bash
# round 6, GPU session 8: the interleaved flush (ILF): bit-identity + parity tests, bench A/B at configs 3 and 4
cd $GRAFT_REPO_ROOT
O=gpurun_out/s8; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "interleaved_flush or config3 or inference_forward_equals or config4 or rows_kernels" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
for rep in 1 2; do
python bench.py --mode infer --batch 2048 --no-cpu-baseline --no-live-traffic --no-other-configs > $O/b3_ilf_$rep.json 2> $O/b3_ilf_$rep.err
PFN_NO_NT_ILF=1 python bench.py --mode infer --batch 2048 --no-cpu-baseline --no-live-traffic --no-other-configs > $O/b3_plain_$rep.json 2> $O/b3_plain_$rep.err
done
python bench.py --case 6470rte --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead > $O/b4_ilf.json 2> $O/b4_ilf.err
PFN_NO_NT_ILF=1 python bench.py --case 6470rte --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead > $O/b4_plain.json 2> $O/b4_plain.err
