#!/bin/bash
export ASYNC_CHECK=${GRAFT_REPO_ROOT:-/root/repo}/tools/check_async_fragments.py   # (csrc/Makefile checks the ISA of the async-fragment objects it links)
# phase timestamps of seg_lin_hops_kernel (SLH_EXP_TS build in /tmp; per workgroup, wall clock 100 MHz), optionally with the A
# fragments loaded from contiguous addresses (SLH_EXP_COAL: wrong results, timing only)
R=$GRAFT_REPO_ROOT
IFS=";" read -ra VARS <<< "${SLH_VARIANTS:-;-DSLH_EXP_COAL}"; for v in "${VARS[@]}"; do
d=/tmp/exp_slh_ts; rm -rf $d; mkdir -p $d; cp -r $R/poweflownet_amd $R/bench.py $R/oracle $R/include $R/BASELINE.json $d/; bash $R/tools/ubench/apply_experiments.sh $d/poweflownet_amd/csrc
( cd $d/poweflownet_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DSLH_EXP_TS $v -c seg_lin_hops.hip -o seg_lin_hops.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC graph.o edge.o gemm.o gemm_nt.o front.o ea_seg.o seg_lin_hops.o model.o physics.o prof.o -o libpfn_hip.so ) || exit 1
echo "######## variant: ${v:-BASE}"
cd $d && python - <<'PY'
import ctypes as C, torch, numpy as np, sys
sys.path.insert(0, ".")
from poweflownet_amd.networks.MPN import MaskEmbdMultiMPN
from poweflownet_amd.synth import make_batch
from poweflownet_amd import _lib as L
torch.manual_seed(0)
m = MaskEmbdMultiMPN(4, 2, 4, 129, 4, 3, 0.2).to("cuda:0").train()
d = make_batch("118v2", 128).to("cuda:0")
lib = L.load()
lib.pfn_debug_slh_ts.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
n = 640 * 8
for _ in range(5):
    m.zero_grad(); out = m(d); torch.cuda.synchronize()
    fbuf = (C.c_ulonglong * n)(); lib.pfn_debug_slh_ts(fbuf, n)     # the LAST forward launch
    torch.nn.MSELoss()(out, d.y).backward()
torch.cuda.synchronize()
bbuf = (C.c_ulonglong * n)(); lib.pfn_debug_slh_ts(bbuf, n)         # the LAST backward launch
def show(buf, title):
    t = np.array(buf[:], dtype=np.int64).reshape(-1, 8)[:512, :5]
    rel = (t - t[:, 0].min()) * 10.0 / 1000.0
    print("==", title)
    for q in range(4):                      # blockIdx.y = q; the kernel maps it to quarter nq - 1 - q
        sel = rel[q * 128:(q + 1) * 128]; dq = np.diff(sel, axis=1)
        print(f"blockIdx.y {q}: start {sel[:,0].mean():5.2f} | staging {dq[:,0].mean():5.2f} (max {dq[:,0].max():5.2f}) | mfma+tile {dq[:,1].mean():5.2f} | epilogue {dq[:,2].mean():5.2f} | hops {dq[:,3].mean():5.2f} | end mean {sel[:,4].mean():6.2f} max {sel[:,4].max():6.2f}")
show(fbuf, "seg_lin_hops_kernel<1> (last forward launch)")
show(bbuf, "seg_lin_hops_kernel<2> (last backward launch)")
PY
done
