R=$GRAFT_REPO_ROOT; C=$R/poweflownet_amd/csrc
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $R/tools/ubench/gemm_nt_bench.hip -L$C -lpfn_hip -Wl,-rpath,$C -o /tmp/nt_bench || exit 1
mkdir -p $R/gpurun_out/bxpmc
for mode in bx fp32; do
  v=2; [ $mode = fp32 ] && v=0   # (0 = the default: fp32 MFMA kernels)
  PFN_NT_BX_MIN_TILES=$v rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmc_$mode -o out --output-format csv -- /tmp/nt_bench 241664 129 129 4 1 5 > /tmp/pmc_$mode.log 2>&1
  f=$(find /tmp/pmc_$mode -name "*counter_collection.csv" | head -1)
  echo "== $mode $f"
  python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if 'gemm_nt' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']:
        agg[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k, len(v), sum(v)/len(v))
PY
  cp $f $R/gpurun_out/bxpmc/$mode.csv
  f2=$(find /tmp/pmc_$mode -name "*kernel_trace.csv" | head -1)
  python3 - "$f2" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'gemm_nt' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print('durations us', [round(x,1) for x in d])
PY
done
