#!/usr/bin/env python3
"""Turn the `ea` PMC pass (TCC_EA0_RDREQ_sum / _32B_sum / _128B_sum: the L2's memory-side read requests by size)
into profiles/traffic_<config>.json for bench.py's roofline.traffic.
    tools/pmc_traffic.py <config>_<pools>x<paths> gpurun_out/pmc_<tag>_ea.txt <kernel-name-prefix> [gpurun_out/pmc_<tag>_tcc2.txt]
Read bytes = 32*n32 + 128*n128 + 64*(total - n32 - n128); write bytes = WRITE_SIZE (KB, exact for this chip per the
guide) of the tcc2 pass when that file is given.  (FETCH_SIZE is not used: MI355X_MICROARCH.md notes it tallies
128-B requests at 64 B for wide streaming reads and is uncalibrated for other widths; the per-size request counters are exact.)"""
import json
import os
import re
import sys

config, path, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
vals = {}
for f in [path] + sys.argv[4:5]:
    for line in open(f):
        m = re.search(r"(TCC_EA0\w+|WRITE_SIZE)\s", line)
        if line.startswith(prefix) and m:
            vals[m.group(1)] = float(line.split()[-1])          # mean per dispatch
tot, n32, n128 = vals["TCC_EA0_RDREQ_sum"], vals["TCC_EA0_RDREQ_32B_sum"], vals["TCC_EA0_RDREQ_128B_sum"]
read_bytes = 32 * n32 + 128 * n128 + 64 * (tot - n32 - n128)
write_bytes = vals["WRITE_SIZE"] * 1024 if "WRITE_SIZE" in vals else None
out = {"kernel": prefix, "hbm_read_bytes_per_launch": round(read_bytes, 1), "hbm_write_bytes_per_launch": write_bytes,
       "hbm_bytes_per_launch": round(read_bytes + (write_bytes or 0), 1), "rdreq": tot, "rdreq_32B": n32, "rdreq_128B": n128,
       "wrreq": vals.get("TCC_EA0_WRREQ_sum"), "source": [os.path.basename(f) for f in [path] + sys.argv[4:5]]}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(root, "profiles", f"traffic_{config}.json"), "w"), indent=1)
print(out)
