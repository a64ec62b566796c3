#!/bin/bash
# big_graph_hops_kernel (edge.hip) with one ingredient removed at a time (compile-time switches; results WRONG by design, only the
# timings of the fused_hops_* classes in bench.py's kernel table mean something): what bounds it at 6470rte x 64?
R=$GRAFT_REPO_ROOT; S=/tmp/exp_src; rm -rf $S; mkdir -p $S; cp -r $R/poweflownet_amd $R/include $R/bench.py $R/oracle $R/configs $R/BASELINE.json $S/; C=$S/poweflownet_amd/csrc; bash $R/tools/ubench/apply_experiments.sh $C; cd $C   # (the switches live in tools/ubench/*.patch.txt)
for v in BASE NOGATHER NOSTORE NOLOADX NOCSR "NOGATHER -DBH_EXP_NOSTORE -DBH_EXP_NOLOADX -DBH_EXP_NOCSR" "NOSTORE -DBH_EXP_NOLOADX"; do
  d=/tmp/bh_$(echo $v | tr -d ' -' | cut -c1-30); mkdir -p $d
  for f in $(ls *.hip | sed "s/.hip//"); do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DBH_EXP_$v -c $f.hip -o $d/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o $d/libpfn_hip.so
  cp $C/libpfn_hip.so /tmp/libpfn_keep.so; cp $d/libpfn_hip.so $C/libpfn_hip.so
  echo "== $v"
  (cd $S && python bench.py --case 6470rte --batch 64 --steps 3 --warmup 1 --profile-steps 3 --no-cpu-baseline --no-live-traffic --no-dp-overhead 2>/dev/null) | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: v['avg_us'] for k, v in d['kernels'].items() if 'hops' in k})"
  cp /tmp/libpfn_keep.so $C/libpfn_hip.so
done
