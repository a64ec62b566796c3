#!/usr/bin/env python3
"""One replayed training step from a rocprofv3 `--kernel-trace -f csv` run: for each kernel its start offset, duration and the
gap since the previous kernel ended (us) -- the launch gaps of the hipGraph chain can be read off.
    ktimeline_csv.py <kernel_trace.csv> [which_step_from_the_end=3]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
if len(ends) <= back + 1:
    sys.exit("not enough steps in the trace")
lo, hi = ends[-back - 2] + 1, ends[-back - 1]
t0 = rows[lo][0]
prev_end = rows[lo][0]
busy = gaps = 0.0
for st, en, name in rows[lo:hi + 1]:
    short = name.split("(")[0].replace("void pfn::", "").replace("pfn::", "")[:36]
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:7.1f} gap {(st - prev_end) / 1e3:6.1f}  {short}")
    busy += (en - st) / 1e3
    gaps += max(0.0, (st - prev_end) / 1e3)
    prev_end = max(prev_end, en)
print(f"{hi - lo + 1} kernels, span {(prev_end - t0) / 1e3:.1f} us, sum of kernel durations {busy:.1f} us, sum of gaps {gaps:.1f} us")
