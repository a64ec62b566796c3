#!/bin/bash
# The product sources carry no experiment switches: the timestamp / ablation instrumentation the scripts in this
# directory build with (-DSG_EXP_*, -DNT_EXP_*, -DT3_EXP_*, -DSLH_EXP_*, -DBH_EXP_*, -DPFN_EXP_*, -DCH_EXP_TS) lives in the
# *.patch.txt files next to this script.  Usage: apply_experiments.sh <a COPY of poweflownet_amd/csrc>
set -e
here=$(cd "$(dirname "$0")" && pwd)
cd "$1"
for f in ea_seg.hip edge.hip gemm.hip gemm_nt.hip seg_lin_hops.hip pfn_internal.hpp seg_tile.hpp; do
    patch -s -p0 "$f" < "$here/experiments_${f%.*}.patch.txt"
done
