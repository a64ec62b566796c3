#!/bin/bash
# cycle accounting with ONE wave per SIMD (PFN_EXP_HALF: the waves 4-7 leave after the barrier; half the tiles are never computed)
R=$GRAFT_REPO_ROOT; NT_EXTRA="-DPFN_EXP_HALF" $R/tools/ubench/run_gemm_nt_ts2.sh
