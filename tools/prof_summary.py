#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite database written by `rocprofv3 --kernel-trace --stats [--pmc ...]`)
as a small text table for profiles/.  Usage: tools/prof_summary.py <results.db> [--pmc] [--by-grid] > profiles/<name>.txt
--by-grid adds a table per (kernel, launch grid): bench.py's exclusive roofline pass launches the extend kernel over one pool of
another size than the timed region's pools, and its launches are told apart by their grid."""
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    name = name[:cut] if cut > 0 else name
    return name if len(name) < 70 else name[:67] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    print(f"# source: {sys.argv[1]}")
    print(f"{'kernel':<72}{'calls':>8}{'total_us':>14}{'avg_us':>12}{'pct':>8}")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{short(name):<72}{calls:>8}{total:>14.1f}{avg:>12.3f}{pct:>8.2f}")
    if "--by-grid" in sys.argv:
        print("\n# per kernel and launch grid (work-items): calls, mean / min / max duration in us")
        q = "select name, grid_x, count(*), avg(duration), min(duration), max(duration) from kernels group by name, grid_x having count(*) >= 8 order by sum(duration) desc"
        for name, grid, n, avg, lo, hi in cur.execute(q):
            print(f"{short(name):<72}{grid:>10}{n:>8}{avg / 1e3:>12.3f}{lo / 1e3:>12.3f}{hi / 1e3:>12.3f}")
    if "--pmc" in sys.argv:
        print("\n# PMC counters per kernel: dispatches, sum over dispatches, mean per dispatch")
        q = ("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
             "group by kernel_name, counter_name order by kernel_name, counter_name")
        for kname, cname, n, tot in cur.execute(q):
            print(f"{short(kname):<72} {cname:<28}{n:>8}{tot:>22.1f}{tot / n:>18.1f}")


if __name__ == "__main__":
    main()
