#!/usr/bin/env python3
"""Per-dispatch kernel durations in launch order from a rocprofv3 --kernel-trace database (rocpd sqlite):
    tools/dispatch_times.py <results.db> <kernel-name-substring> [first] [last]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel_dispatch" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
name_col = "name" if "name" in cols else "kernel_name"
rows = list(cur.execute(f"select {name_col}, start, end from {view} order by start"))
sel = [(n, s, e) for n, s, e in rows if sys.argv[2] in n]
a = int(sys.argv[3]) if len(sys.argv) > 3 else 0
b = int(sys.argv[4]) if len(sys.argv) > 4 else len(sel)
for i, (n, s, e) in enumerate(sel[a:b], a):
    print(i, round((e - s) / 1e3, 1))
