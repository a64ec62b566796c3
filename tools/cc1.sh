#!/bin/bash
# compile ONE translation unit of libpfn_hip.so and print its kernels' register / spill figures:  tools/cc1.sh gemm_nt
cd /root/repo/poweflownet_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -c $1.hip -o $1.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -i "error\|warning\|Function Name\|spill\|scratch\|VGPRs:\|SGPRs:" | sed 's/.*remark: //' 
