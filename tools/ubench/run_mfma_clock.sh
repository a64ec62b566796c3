#!/bin/bash
# what s_memtime counts and the clock under sustained fp32 MFMA (gpurun -- tools/ubench/run_mfma_clock.sh)
R=$GRAFT_REPO_ROOT; cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w mfma_clock.hip -o /tmp/mfma_clock || exit 1
/tmp/mfma_clock
cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/clk -o clk --output-format csv -- /tmp/mfma_clock > /dev/null 2>&1
python3 - <<'P'
import csv, glob
f = glob.glob('/tmp/clk/**/*counter_collection.csv', recursive=True)
kt = glob.glob('/tmp/clk/**/*kernel_trace.csv', recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])): dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
agg = {}
for r in csv.DictReader(open(f[0])): agg.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
for d, c in sorted(agg.items(), key=lambda x: int(x[0])):
    g = c.get('GRBM_GUI_ACTIVE', 0); b = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
    print("dispatch %s: %.1f us  GRBM_GUI_ACTIVE %.0f -> /8/dur = %.3f GHz   MFMA_BUSY %.0f" % (d, dur.get(d, 0), g, g / 8 / max(dur.get(d, 1), 1e-9) / 1e3, b))
P
