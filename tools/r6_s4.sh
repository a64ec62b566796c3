#!/bin/bash
# round 6, GPU session 4: gemm_nt piece-by-piece publication (PUB) -- harness A/B, parity tests, bench A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w gemm_nt_bench.hip -L$R/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$R/poweflownet_amd/csrc -o /tmp/gemm_nt_bench ) || exit 1
for rep in 1 2; do
for cfg in "15104 129 129 4 1" "15104 129 129 2 1" "7552 129 129 4 1" "15104 129 129 4 2"; do
  echo "== $cfg PUB"; PFN_NT_TINY_MAX_TILES=0 /tmp/gemm_nt_bench $cfg 200 | grep -v "bad element"
  echo "== $cfg NO_PUB"; PFN_NO_NT_PUB=1 PFN_NT_TINY_MAX_TILES=0 /tmp/gemm_nt_bench $cfg 200 | grep -v "bad element"
done; done > $O/harness.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_mse_tail.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
for rep in 1 2; do
python bench.py --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead > $O/b2_pub_$rep.json 2> $O/b2_pub_$rep.err
PFN_NO_NT_PUB=1 python bench.py --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead > $O/b2_nopub_$rep.json 2> $O/b2_nopub_$rep.err
done
