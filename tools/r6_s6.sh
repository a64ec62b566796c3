#!/bin/bash
# round 6, GPU session 6: per-term pairing in gemm_nt (CT = 1): whole GPU suite, parity reports with / without, bench A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
cp gpurun_out/parity_report.json $O/parity_report_pair.json
PFN_NO_NT_PAIR=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "config2 or config3_size or hidden512 or wide" > $O/pytest_nopair.log 2>&1; echo "pytest exit $?" >> $O/pytest_nopair.log
cp gpurun_out/parity_report.json $O/parity_report_nopair.json
for rep in 1 2; do
python bench.py --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead > $O/b2_pair_$rep.json 2> $O/b2_pair_$rep.err
PFN_NO_NT_PAIR=1 python bench.py --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead > $O/b2_nopair_$rep.json 2> $O/b2_nopair_$rep.err
done
