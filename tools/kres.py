#!/usr/bin/env python3
"""Register / spill table of every kernel of one csrc source:  tools/kres.py gemm_nt.hip [name filter]"""
import os, re, subprocess, sys
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "poweflownet_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", sys.argv[1], "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], cwd=csrc, capture_output=True, text=True).stderr
cur, rows = None, []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None: cur[m.group(1).strip()] = int(m.group(2))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt not in name: continue
    g = lambda k: r.get(k, -1)
    print(f"{name[:70]:70s} VGPR {g('VGPRs'):4d} AGPR {g('AGPRs'):4d} vspill {g('VGPRs Spill'):4d} sspill {g('SGPRs Spill'):4d} scratch {g('ScratchSize'):5d} occ {g('Occupancy')}")
