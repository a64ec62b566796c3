#!/bin/bash
# ea_seg kernels with one ingredient removed at a time (compile-time switches in ea_seg.hip; results are WRONG by design, only
# the timings mean something): where do the microseconds of a launch go?   gpurun -- tools/ubench/run_ea_seg_exp.sh
R=$GRAFT_REPO_ROOT; C=$R/poweflownet_amd/csrc; cd $C
export TMPDIR=/tmp
for v in BASE NOMFMA NOLOAD NOWALK NOSTORE NOSTAGE NOPQ "NOWALK -DSG_EXP_NOSTORE -DSG_EXP_NOPQ -DSG_EXP_NOSTAGE" "NOWALK -DSG_EXP_NOSTORE -DSG_EXP_NOPQ -DSG_EXP_NOSTAGE -DSG_EXP_NOMFMA -DSG_EXP_NOLOAD"; do
  d=/tmp/exp_$(echo $v | tr -d ' -' ); mkdir -p $d
  cp -r $R/poweflownet_amd $R/bench.py $R/oracle $R/include $R/BASELINE.json $d/ 2>/dev/null; bash $R/tools/ubench/apply_experiments.sh $d/poweflownet_amd/csrc
  ( cd $d/poweflownet_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DSG_EXP_$v -c ea_seg.hip -o ea_seg.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC graph.o edge.o gemm.o gemm_nt.o front.o ea_seg.o model.o physics.o prof.o -o libpfn_hip.so ) || exit 1
  echo "== $v"
  ( cd $d && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $d/prof -o b -- python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $d/bench.json 2> $d/bench.err )
  python3 - $d <<'PY'
import csv, glob, sys, json
d = sys.argv[1]
f = glob.glob(d + "/prof/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "ea_seg" in r["Name"]:
        print("   ", r["Name"][:48].ljust(48), r["Calls"], round(float(r["AverageNs"]) / 1000, 2), "us")
try:
    print("    step ms", json.loads(open(d + "/bench.json").read().strip().splitlines()[-1])["ms_per_step"])
except Exception as e:
    print("    bench failed", e)
PY
done
