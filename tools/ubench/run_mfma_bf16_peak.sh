#!/bin/bash
# sustained bf16 vs fp32 MFMA rate with nonzero data (gpurun -- bash tools/ubench/run_mfma_bf16_peak.sh)
R=$GRAFT_REPO_ROOT; cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w mfma_bf16_peak.hip -o /tmp/mfma_bf16_peak || exit 1
/tmp/mfma_bf16_peak
