#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/t; mkdir -p $O
bash tools/ubench/run_gemm_nt_ts.sh > $O/nt_ts.txt 2>&1
bash tools/ubench/run_gemm_nt_small.sh > $O/nt_small.txt 2>&1
PFN_NT_NOPIPE=1 bash tools/ubench/run_gemm_nt_small.sh > $O/nt_small_nopipe.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
python bench.py --no-cpu-baseline > $O/b2.json 2> $O/b2.err
PFN_NT_NOPIPE=1 python bench.py --no-cpu-baseline > $O/b2_nopipe.json 2> $O/b2_nopipe.err
python bench.py --no-cpu-baseline --mode infer --batch 2048 > $O/b3.json 2> $O/b3.err
python bench.py --no-cpu-baseline --case 6470rte --batch 64 --steps 10 --warmup 3 > $O/b4.json 2> $O/b4.err
