#!/usr/bin/env python3
"""Per-launch durations of k_extend_queue from a rocprofv3 --kernel-trace database, grouped by position within a step
(the trip-capped extend stage is up to three launches per step).  Usage: tools/phase_times.py <results.db> <launches-per-step>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1]); per = int(sys.argv[2])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t][0] if any("kernel_dispatch" in t for t in tabs) else None
view = "kernels" if "kernels" in tabs else kd
cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
print("# view", view, cols)
rows = list(cur.execute(f"select name, start, end from {view} order by start"))
ext = [(s, e) for n, s, e in rows if n.startswith("void k_extend_queue") or n.startswith("k_extend_queue")]
print("# extend launches", len(ext))
import statistics
for ph in range(per):
    d = [(e - s) / 1e3 for i, (s, e) in enumerate(ext) if i % per == ph]
    mid = d[len(d) // 4: len(d) // 2]       # the steady part of the first render
    print(f"phase {ph}: n {len(d)}  mean {statistics.mean(d):.1f} us  steady-part mean {statistics.mean(mid):.1f} us  max {max(d):.1f}")
