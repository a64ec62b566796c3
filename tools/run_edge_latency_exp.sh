#!/bin/bash
# Is the forward edge walk at 6470rte x 64 bound by its chain of dependent loads (row pointers -> indices -> rows)?  A build whose
# walk takes neighbour / edge ids WITHOUT loading them (results wrong by design) against the product build, same bench.
#   gpurun --timeout 900 -- bash tools/run_edge_latency_exp.sh
cd $GRAFT_REPO_ROOT; C=poweflownet_amd/csrc; O=gpurun_out/edge_exp; mkdir -p $O
python bench.py --no-cpu-baseline --no-live-traffic --no-dp-overhead --case 6470rte --batch 64 --steps 6 --warmup 2 > $O/base.json 2> $O/base.err
cp $C/libpfn_hip.so /tmp/libpfn_base.so
(cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DPFN_EXP_EDGE_NOL2 -c edge.hip -o /tmp/edge_exp.o && \
 /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC graph.o /tmp/edge_exp.o gemm.o gemm_nt.o front.o ea_seg.o model.o physics.o prof.o -o libpfn_hip.so) || exit 1
python bench.py --no-cpu-baseline --no-live-traffic --no-dp-overhead --case 6470rte --batch 64 --steps 6 --warmup 2 > $O/nol2.json 2> $O/nol2.err
cp /tmp/libpfn_base.so $C/libpfn_hip.so
python - <<'PY'
import json
for t in ("base", "nol2"):
    d = json.loads(open(f"gpurun_out/edge_exp/{t}.json").read().strip().splitlines()[-1])
    print(t, d["ms_per_step"], {k: v["avg_us"] for k, v in d["kernels"].items() if "edge" in k})
PY
