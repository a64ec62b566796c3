#!/usr/bin/env python3
"""Print the last N launches of kernels matching a pattern, in launch order, with durations (us)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
like, n = sys.argv[2], int(sys.argv[3])
rows = db.execute("select name, grid_x, workgroup_x, (end-start)/1e3 from kernels where name like ? order by start desc limit ?",
                  (like, n)).fetchall()[::-1]
print(" ".join(f"{r[3]:.0f}" for r in rows))
