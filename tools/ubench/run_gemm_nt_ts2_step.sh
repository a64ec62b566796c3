#!/bin/bash
export ASYNC_CHECK=${GRAFT_REPO_ROOT:-/root/repo}/tools/check_async_fragments.py   # (csrc/Makefile checks the ISA of the async-fragment objects it links)
# gemm_nt's per-wave cycle accounting (NT_EXP_TS2 build in /tmp) INSIDE the replayed step of configs 3 and 4, and back to back in
# the harness: pipe occupancy in shader cycles and the shader clock (s_memtime / wall_clock64) the launches actually ran at
R=$GRAFT_REPO_ROOT; d=/tmp/exp_ts2r; rm -rf $d; mkdir -p $d
cp -r $R/poweflownet_amd $R/include $R/bench.py $R/configs $R/oracle $R/BASELINE.json $d/ 2>/dev/null; bash $R/tools/ubench/apply_experiments.sh $d/poweflownet_amd/csrc
( cd $d/poweflownet_amd/csrc && rm -f *.o libpfn_hip.so && make -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DNT_EXP_TS2" > /dev/null ) || exit 1
cd $d
for cfgargs in "--case 118v2 --batch 2048 --mode infer --steps 30 --warmup 5" "--case 6470rte --batch 64 --mode train --steps 12 --warmup 3"; do
  echo "== in the step: $cfgargs $EXTRA"
  python - $cfgargs $EXTRA --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --profile-steps 0 <<'P'
import sys, runpy, ctypes
sys.argv = ["bench.py"] + sys.argv[1:]
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
sys.stdout.flush()
from poweflownet_amd import _lib
ctypes.CDLL(_lib.LIB_PATH).pfn_debug_nt_ts2_dump()
P
done 2>&1 | cut -c1-400
cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DNT_EXP_TS2 gemm_nt_bench.hip -L$d/poweflownet_amd/csrc -lpfn_hip -Wl,-rpath,$d/poweflownet_amd/csrc -o /tmp/gemm_nt_bench_ts2 || exit 1
echo "== back to back (harness)"
for cfg in "414080 129 129 1 1" "414080 129 129 4 1" "241664 129 129 4 1"; do PFN_NT_TINY_MAX_TILES=0 /tmp/gemm_nt_bench_ts2 $cfg 20 | grep -v "bad element\|waves:"; done
