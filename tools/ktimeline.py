#!/usr/bin/env python3
"""Timeline of the last N kernel launches of a rocprofv3 (rocpd) trace: start offset, duration, gap since the previous
kernel ended (all in us), so the launch gaps and the overlap of the two streams of a replayed step can be read off."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 90
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = db.execute(f"select name, start, end, {q} from kernels order by start desc limit ?", (n,)).fetchall()[::-1]
t0 = rows[0][1]
prev_end = rows[0][1]
busy = 0.0
for name, st, en, qid in rows:
    short = name.split("(")[0].replace("void pfn::", "").replace("pfn::", "")[:34]
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:7.1f} gap {(st - prev_end) / 1e3:6.1f} q{qid} {short}")
    busy += (en - st) / 1e3
    prev_end = max(prev_end, en)
print(f"span {(prev_end - t0) / 1e3:.1f} us, sum of kernel durations {busy:.1f} us")
