#!/bin/bash
# what the second wave of a SIMD costs a wave that streams fp32 MFMAs, by instruction kind (gpurun -- tools/ubench/run_mfma_partner.sh)
R=$GRAFT_REPO_ROOT; cd $R/tools/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w mfma_partner.hip -o /tmp/mfma_partner || exit 1
/tmp/mfma_partner
