#!/bin/bash
export ASYNC_CHECK=${GRAFT_REPO_ROOT:-/root/repo}/tools/check_async_fragments.py   # (csrc/Makefile checks the ISA of the async-fragment objects it links)
# gemm_tn's per-wave cycle stamps (T3_EXP_TS build in /tmp) for the LAST launch of an eager training step: where do the cycles
# between "kernel start" and "partials published" go?  tools/ubench/run_gemm_tn_ts.sh ["bench args"]
R=$GRAFT_REPO_ROOT; d=/tmp/exp_t3ts; rm -rf $d; mkdir -p $d
cp -r $R/poweflownet_amd $R/include $R/bench.py $R/configs $R/oracle $R/BASELINE.json $d/ 2>/dev/null; bash $R/tools/ubench/apply_experiments.sh $d/poweflownet_amd/csrc
( cd $d/poweflownet_amd/csrc && rm -f *.o libpfn_hip.so && make -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DT3_EXP_TS $T3_EXTRA" > /dev/null ) || exit 1
cd $d
for cfgargs in "${1:---case 118v2 --batch 128 --mode train --steps 20 --warmup 5}"; do
  echo "== $cfgargs"
  python - $cfgargs --no-graph --no-cpu-baseline --no-live-traffic --no-other-configs --no-dp-overhead --profile-steps 0 <<'P'
import sys, runpy, ctypes
import numpy as np
sys.argv = ["bench.py"] + sys.argv[1:]
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
sys.stdout.flush()
from poweflownet_amd import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (4096 * 8))()
lib.pfn_debug_t3_ts(buf, 4096 * 8)
t = np.array(buf[:], dtype=np.float64).reshape(4096, 8)
t = t[t[:, 0] > 0]
kind = (t[:, 7] // 1e6).astype(int); nsplit = (t[:, 7] % 1e6).astype(int)
wave = np.arange(len(t)) % 8
w0 = t[:, 5].min()
clk = np.nanmedian((t[:, 4] - t[:, 0])[t[:, 4] > 0] / ((t[:, 6] - t[:, 5])[t[:, 4] > 0] * 10.0))
print(f"waves {len(t)}, shader clock over the body ~{clk:.3f} GHz (waves that publish)")
print("kernel start spread (wall, us): mean %.2f max %.2f" % (((t[:, 5] - w0) * 0.01).mean(), ((t[:, 5] - w0) * 0.01).max()))
for k in sorted(set(kind)):
    m = kind == k
    fill = (t[m, 1] - t[m, 0]); loop = (t[m, 2] - t[m, 1]); tree = (t[m, 3] - t[m, 2])
    pub = (t[m, 4] - t[m, 3])[t[m, 4] > 0]
    print(f"shape TA/TB {k:02d}: {m.sum()} waves, splits {sorted(set(nsplit[m]))} | cycles: ring fill {fill.mean():.0f} | loop {loop.mean():.0f} (min {loop.min():.0f} max {loop.max():.0f}) | tree {tree.mean():.0f} | publish {pub.mean() if len(pub) else 0:.0f}")
    for half in (0, 1):
        mm = m & ((wave >= 4) == bool(half))
        if mm.any():
            print(f"    waves {'4-7' if half else '0-3'}: loop {(t[mm, 2] - t[mm, 1]).mean():.0f}, loop end at {(t[mm, 2] - t[mm, 0]).mean():.0f} cycles after the wave's start")
end_wall = (t[:, 6][t[:, 6] > 0] - w0) * 0.01
print("publish done (wall us after the first wave's start): p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(end_wall, [10, 50, 90, 100])))
P
done 2>&1 | grep -v "^{\|amdgpu.ids"
