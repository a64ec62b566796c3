#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database: per kernel (and launch shape) count / avg / min duration in us."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
like = sys.argv[2] if len(sys.argv) > 2 else "%"
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = db.execute("select name, grid_x, grid_y, workgroup_x, count(*), avg(end-start)/1e3, min(end-start)/1e3, sum(end-start)/1e3 "
                  "from kernels where name like ? group by name, grid_x, grid_y order by 8 desc", (like,)).fetchall()
tot = sum(r[7] for r in rows)
print(f"total {tot:.1f} us over all launches; {tot / steps:.1f} us per step ({steps:g} steps)")
for r in rows:
    name = r[0].replace("void ", "").split("(")[0][:48]
    print(f"{name:48s} blocks=({r[1] // max(r[3], 1):6d},{r[2]:3d}) n={r[4]:5d} avg={r[5]:8.2f} min={r[6]:8.2f} per_step={r[7] / steps:8.1f}")
