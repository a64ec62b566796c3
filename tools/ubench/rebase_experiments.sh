#!/bin/bash
# Re-derives every tools/ubench/*.patch.txt against the CURRENT product sources: applies them (offsets / fuzz allowed) to a scratch
# copy of poweflownet_amd/csrc and writes each patch back as an exact `diff -u`.  Fails if any hunk is rejected (fix that hunk by
# hand first).  Run after every kernel edit, before committing: tests/test_abi.py requires the patches to apply.
set -e
here=$(cd "$(dirname "$0")" && pwd)
root=$(cd "$here/../.." && pwd)
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
mkdir "$tmp/csrc"
cp "$root"/poweflownet_amd/csrc/*.hip "$root"/poweflownet_amd/csrc/*.hpp "$tmp/csrc/"
bash "$here/apply_experiments.sh" "$tmp/csrc"
if ls "$tmp"/csrc/*.rej >/dev/null 2>&1; then echo "rejected hunks:"; ls "$tmp"/csrc/*.rej; exit 1; fi
for f in ea_seg.hip edge.hip gemm.hip gemm_nt.hip seg_lin_hops.hip pfn_internal.hpp seg_tile.hpp; do
    out="$here/experiments_${f%.*}.patch.txt"
    (cd "$root" && diff -u --label "poweflownet_amd/csrc/$f" --label "$f (with the experiment switches)" "poweflownet_amd/csrc/$f" "$tmp/csrc/$f" > "$out") || true
done
echo "experiment patches rebased"
