#!/bin/bash
# does pacing an fp32 MFMA stream with s_nop free the vector issue port for the SIMD partner? (gpurun -- tools/ubench/run_mfma_pace.sh)
R=$GRAFT_REPO_ROOT; cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w mfma_pace.hip -o /tmp/mfma_pace || exit 1
/tmp/mfma_pace
