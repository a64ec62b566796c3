#!/bin/bash
# round 6, GPU session 9: the one-wave-per-SIMD interleaved flush (ILF4): bit-identity, parity, A/B against ILF2 and the plain flush
cd $GRAFT_REPO_ROOT
O=gpurun_out/s9; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "interleaved_flush or config3 or inference_forward_equals or config4" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
R=$GRAFT_REPO_ROOT
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w gemm_nt_bench.hip -L$R/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$R/poweflownet_amd/csrc -o /tmp/gemm_nt_bench )
for rep in 1 2; do for cfg in "241664 129 129 1 1" "241664 129 129 2 2" "414080 129 129 1 1"; do
  echo "== $cfg ILF4"; /tmp/gemm_nt_bench $cfg 30 | grep -v "bad element"
  echo "== $cfg ILF2"; PFN_NO_NT_ILF4=1 /tmp/gemm_nt_bench $cfg 30 | grep -v "bad element"
  echo "== $cfg plain"; PFN_NO_NT_ILF=1 /tmp/gemm_nt_bench $cfg 30 | grep -v "bad element"
done; done > $O/harness.txt 2>&1
for rep in 1 2; do
python bench.py --mode infer --batch 2048 --no-cpu-baseline --no-live-traffic --no-other-configs > $O/b3_ilf4_$rep.json 2> $O/b3_ilf4_$rep.err
PFN_NO_NT_ILF4=1 python bench.py --mode infer --batch 2048 --no-cpu-baseline --no-live-traffic --no-other-configs > $O/b3_ilf2_$rep.json 2> $O/b3_ilf2_$rep.err
PFN_NO_NT_ILF=1 python bench.py --mode infer --batch 2048 --no-cpu-baseline --no-live-traffic --no-other-configs > $O/b3_plain_$rep.json 2> $O/b3_plain_$rep.err
done
python bench.py --case 6470rte --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead > $O/b4_ilf4.json 2> $O/b4_ilf4.err
PFN_NO_NT_ILF=1 python bench.py --case 6470rte --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead > $O/b4_plain.json 2> $O/b4_plain.err
