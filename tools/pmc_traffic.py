#!/usr/bin/env python3
"""Turn the `ea` PMC pass (TCC_EA0_RDREQ_sum / _32B_sum / _128B_sum: the L2's memory-side read requests by size)
into profiles/traffic_<config>.json for bench.py's roofline.traffic.
    tools/pmc_traffic.py <config> gpurun_out/pmc_<tag>_ea.txt <kernel-name-prefix>
Bytes = 32*n32 + 128*n128 + 64*(total - n32 - n128).  (FETCH_SIZE is not used: MI355X_MICROARCH.md notes it tallies
128-B requests at 64 B for wide streaming reads and is uncalibrated for other widths; the per-size request counters are exact.)"""
import json
import os
import sys

config, path, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
vals = {}
for line in open(path):
    if line.startswith(prefix) and "TCC_EA0" in line:
        parts = line.split()
        name = [p for p in parts if p.startswith("TCC_EA0")][0]
        vals[name] = float(parts[-1])          # mean per dispatch
tot, n32, n128 = vals["TCC_EA0_RDREQ_sum"], vals["TCC_EA0_RDREQ_32B_sum"], vals["TCC_EA0_RDREQ_128B_sum"]
read_bytes = 32 * n32 + 128 * n128 + 64 * (tot - n32 - n128)
out = {"kernel": prefix, "hbm_read_bytes_per_launch": round(read_bytes, 1), "rdreq": tot, "rdreq_32B": n32, "rdreq_128B": n128,
       "wrreq": vals.get("TCC_EA0_WRREQ_sum"), "source": os.path.basename(path)}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(root, "profiles", f"traffic_{config}.json"), "w"), indent=1)
print(out)
