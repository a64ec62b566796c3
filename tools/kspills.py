#!/usr/bin/env python3
"""Where a kernel spills: tools/kspills.py <source.hip> <mangled-name substring>  -- scratch loads / stores of the kernel with the
count of MFMAs in front of them (file order), from a -save-temps build of the source."""
import os, subprocess, sys, tempfile, collections
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "poweflownet_amd", "csrc")
d = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-save-temps=obj", "-c", sys.argv[1], "-o", d + "/x.o"],
               cwd=csrc, capture_output=True)
asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
s = open(os.path.join(d, asm)).read()
for blk in s.split(".Lfunc_end")[:-1]:
    lines = blk.splitlines()
    names = [l.split(":")[0] for l in lines if l.startswith("_Z") and ":" in l]
    if not names or sys.argv[2] not in names[-1]: continue
    i0 = max(k for k, l in enumerate(lines) if l.startswith(names[-1] + ":"))
    body = lines[i0:]
    mf = 0
    print(names[-1], len(body), "lines")
    for k, l in enumerate(body):
        if "v_mfma" in l: mf += 1
        if "scratch_" in l or "v_writelane" in l or "v_readlane" in l: print(f"  line {k:5d} after {mf:4d} MFMAs: {l.strip()}")
    print("  total MFMAs", mf)
if len(sys.argv) > 3: print("asm kept in", d)
