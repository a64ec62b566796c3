#!/bin/bash
# sustained fp32 MFMA rate of the chip vs waves per SIMD / accumulators per wave (gpurun -- tools/ubench/run_mfma_peak.sh)
R=$GRAFT_REPO_ROOT; cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w mfma_peak.hip -o /tmp/mfma_peak || exit 1
/tmp/mfma_peak
