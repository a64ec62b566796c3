#!/bin/bash
export ASYNC_CHECK=${GRAFT_REPO_ROOT:-/root/repo}/tools/check_async_fragments.py   # (csrc/Makefile checks the ISA of the async-fragment objects it links)
# phase timestamps of seg_chain_fwd_kernel (the instrumented build -- tools/ubench/seg_chain_timestamps.patch.txt, -DCH_EXP_TS -- in /tmp; per workgroup and stage, wall clock 100 MHz)
R=$GRAFT_REPO_ROOT
export PFN_SEG_CHAIN=1
d=/tmp/exp_chain_ts; rm -rf $d; mkdir -p $d; cp -r $R/poweflownet_amd $R/bench.py $R/oracle $R/include $R/BASELINE.json $d/; bash $R/tools/ubench/apply_experiments.sh $d/poweflownet_amd/csrc
( cd $d/poweflownet_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DCH_EXP_TS $CH_DEFS -c seg_chain.hip -o seg_chain.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC graph.o edge.o gemm.o gemm_nt.o front.o ea_seg.o seg_lin_hops.o seg_chain.o model.o physics.o prof.o -o libpfn_hip.so ) || exit 1
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
lib.pfn_debug_chain_ts.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
n = 512 * 8 * 16
for _ in range(5):
    m.zero_grad(); out = m(d); torch.cuda.synchronize()
    fbuf = (C.c_ulonglong * n)(); lib.pfn_debug_chain_ts(fbuf, n)
    torch.nn.MSELoss()(out, d.y).backward()
torch.cuda.synchronize()
t = np.array(fbuf[:], dtype=np.int64).reshape(512, 8, 16)[:, :3, :12]
t0 = t[:, 0, 0].min()
rel = (t - t0) / 100.0          # us
names = ["waitS", "A.mfma", "A.epi", "A.hops", "A.drain", "waitX", "B.mfma", "B.epi+drain", "waitH", "C.mfma", "C.walk"]
bt = np.arange(512) >> 3
quarter = 3 - (bt % 4)
for q in (3, 0):
    sel = rel[quarter == q]
    print(f"== quarter {q} ({'with' if q == 3 else 'without'} the trailing column): mean over workgroups, us")
    for s in range(3):
        dq = np.diff(sel[:, s, :], axis=1)
        print(f" stage {s}: start {sel[:, s, 0].mean():7.2f} end {sel[:, s, 11].mean():7.2f} (max {sel[:, s, 11].max():7.2f}) | " + " ".join(f"{nm} {dq[:, i].mean():5.2f}" for i, nm in enumerate(names)))
PY
