#!/bin/bash
export ASYNC_CHECK=${GRAFT_REPO_ROOT:-/root/repo}/tools/check_async_fragments.py   # (csrc/Makefile checks the ISA of the async-fragment objects it links)
# phase timestamps of ea_seg_bwd_kernel<false> (SG_EXP_TS build in /tmp): per workgroup, wall clock 100 MHz
R=$GRAFT_REPO_ROOT; d=/tmp/exp_ts; mkdir -p $d; cp -r $R/poweflownet_amd $R/bench.py $R/oracle $R/include $R/BASELINE.json $d/; bash $R/tools/ubench/apply_experiments.sh $d/poweflownet_amd/csrc
( cd $d/poweflownet_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DSG_EXP_TS -c ea_seg.hip -o ea_seg.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC graph.o edge.o gemm.o gemm_nt.o front.o ea_seg.o seg_lin_hops.o model.o physics.o prof.o -o libpfn_hip.so ) || exit 1
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
lib.pfn_debug_seg_ts.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for _ in range(5):
    m.zero_grad(); out = m(d); torch.cuda.synchronize()
    fbuf = (C.c_ulonglong * (640 * 8))(); lib.pfn_debug_seg_ts(fbuf, 640 * 8)     # the LAST forward launch (layer 3's P | Q + walk)
    torch.nn.MSELoss()(out, d.y).backward()
torch.cuda.synchronize()
def show(buf, title):
    t = np.array(buf[:], dtype=np.int64).reshape(-1, 8)[:512, :5]
    rel = (t - t[:, 0].min()) * 10.0 / 1000.0
    print("==", title)
    for q in range(4):                      # blockIdx.y = q; the kernel maps it to quarter nq - 1 - q
        sel = rel[q * 128:(q + 1) * 128]; dq = np.diff(sel, axis=1)
        print(f"blockIdx.y {q}: start {sel[:,0].mean():5.2f} | staging {dq[:,0].mean():5.2f} | mfma+tile {dq[:,1].mean():5.2f} | walks {dq[:,2].mean():5.2f} | reduce {dq[:,3].mean():5.2f} | end mean {sel[:,4].mean():6.2f} max {sel[:,4].max():6.2f}")
show(fbuf, "ea_seg_fwd (last forward launch)")
n = 640 * 8
buf = (C.c_ulonglong * n)()
print("rc", lib.pfn_debug_seg_ts(buf, n))
show(buf, "ea_seg_bwd<false> (last backward launch)")
t = np.array(buf[:], dtype=np.int64).reshape(-1, 8)[:512, :5]      # the LAST bwd launch = layer 0 (dS by MFMA): 128 x 4 blocks
t0 = t[:, 0].min()
rel = (t - t0) * 10.0 / 1000.0                                      # us
print("blocks", len(rel))
names = ["start", "after staging+DMA barrier", "after MFMA+tile+rem barrier", "after walks", "end"]
for i, nm in enumerate(names):
    print(f"{nm:32s} min {rel[:, i].min():7.2f}  mean {rel[:, i].mean():7.2f}  max {rel[:, i].max():7.2f} us")
dur = np.diff(rel, axis=1)
for i, nm in enumerate(["staging", "mfma+tile", "walks", "dwe reduce"]):
    print(f"phase {nm:12s} mean {dur[:, i].mean():6.2f}  max {dur[:, i].max():6.2f} us")
for q in range(4):
    sel = rel[q * 128:(q + 1) * 128]
    dq = np.diff(sel, axis=1)
    print(f"quarter {q}: start {sel[:,0].mean():5.2f} | staging {dq[:,0].mean():5.2f} | mfma+tile {dq[:,1].mean():5.2f} | walks {dq[:,2].mean():5.2f} | reduce {dq[:,3].mean():5.2f} | end mean {sel[:,4].mean():6.2f} max {sel[:,4].max():6.2f}")
PY
