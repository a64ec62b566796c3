#!/bin/bash
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $GRAFT_REPO_ROOT/tools/ubench/mfma_interleave.hip -o /tmp/mfma_interleave && /tmp/mfma_interleave
