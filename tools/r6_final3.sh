#!/bin/bash
# round 6: the six profile sets and the seven unprofiled bench lines on the final tree (XCD-paired slices, serpentine sweeps)
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_profiles.sh r06 > gpurun_out/profiles3.log 2>&1
bash tools/gpu_bench_lines.sh r06 > gpurun_out/benchlines3.log 2>&1
