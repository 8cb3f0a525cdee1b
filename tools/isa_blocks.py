#!/usr/bin/env python3
"""Per-basic-block instruction counts (vector / scalar / memory, scratch flagged) of one kernel in a gfx950 assembly listing.

    hipcc --offload-arch=gfx950 <the Makefile's FLAGS> -S --cuda-device-only pt_kernels_extend.hip -o /tmp/ext.s
    tools/isa_blocks.py /tmp/ext.s '<mangled-name prefix>'

Shows which blocks of the traversal loop hold the bodies' instructions and whether a register spill sits inside a loop."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ": " in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blk, stats, order = "entry", {"entry": {"v": 0, "s": 0, "mem": 0, "hdr": ""}}, ["entry"]
for i in range(start + 1, end):
    l = lines[i]
    m = re.match(r"(\.LBB\d+_\d+):", l)
    if m:
        blk = m.group(1); stats[blk] = {"v": 0, "s": 0, "mem": 0, "hdr": ""}; order.append(blk)
        continue
    if "Loop Header" in l or "Parent Loop" in l or "in Loop" in l:
        stats[blk]["hdr"] += re.sub(r"\s+", " ", l.strip())[:60] + " | "
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m:
        continue
    op = m.group(1)
    if op.startswith("v_"):
        stats[blk]["v"] += 1
    elif op.startswith("s_"):
        stats[blk]["s"] += 1
    elif op.startswith(("global_", "ds_", "scratch_", "flat_", "buffer_")):
        stats[blk]["mem"] += 1
        if op.startswith("scratch_"):
            stats[blk]["hdr"] += "SCRATCH "
print(f"{'block':<12}{'valu':>6}{'salu':>6}{'mem':>5}  loop")
for b in order:
    s = stats[b]
    if s["v"] + s["s"] + s["mem"] > 0:
        print(f"{b:<12}{s['v']:>6}{s['s']:>6}{s['mem']:>5}  {s['hdr'][:150]}")
