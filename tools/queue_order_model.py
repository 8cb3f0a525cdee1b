#!/usr/bin/env python3
"""What would ordering the ray queue buy the kind-scheduled walk?  Model on recorded rays (tools/sched_model_record.py):
vector instructions per wave (tools/sched_model.py's pass costs, weights 6:9:5) when the rays of each class are ordered by a key
inside blocks of `block` consecutive queue entries (what a compaction workgroup could do without a global sort) or over the
whole queue.  Keys use only what the material / camera stage knows before the walk.

    tools/queue_order_model.py <rec.npz> [block]"""
import sys

import numpy as np

from sched_model import micro_ops, run, weights

z = np.load(sys.argv[1]); kinds, ln, ncam, o, d, prev = z["kinds"], z["ln"], int(z["ncam"]), z["o"], z["d"], z["prev_obj"]
block = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rng = np.random.default_rng(1)


def cost(idx, n_waves=120):
    nw = len(idx) // 64; ws = rng.choice(nw, size=min(n_waves, nw), replace=False)
    tot = use = 0
    for w in ws:
        ii = idx[w * 64:(w + 1) * 64]
        t, p, u = run(micro_ops(kinds[ii], ln[ii]), weights(6, 9, 5)); tot += t; use += u
    return tot / len(ws), use / 64 / tot


def ordered(lo, hi, key, blk):
    idx = np.arange(lo, hi)
    if key is None: return idx
    if blk is None: return idx[np.argsort(key[idx], kind="stable")]
    out = []
    for b in range(0, len(idx), blk):
        part = idx[b:b + blk]; out.append(part[np.argsort(key[part], kind="stable")])
    return np.concatenate(out)


octant = (d[:, 0] > 0).astype(int) | ((d[:, 1] > 0).astype(int) << 1) | ((d[:, 2] > 0).astype(int) << 2)
up = (d[:, 2] > 0).astype(int)
keys = {"queue order": None, "previous object": prev, "previous object, up/down": prev * 2 + up, "previous object, octant": prev * 8 + octant,
        "direction z in 8 steps": np.floor((d[:, 2] + 1) * 3.999).astype(int), "previous object, direction z in 8 steps": prev * 8 + np.floor((d[:, 2] + 1) * 3.999).astype(int),
        "trip count (oracle knowledge: upper bound)": ln}
for cls, lo, hi in (("camera", 0, ncam), ("bounce", ncam, len(ln))):
    print(f"# {cls} rays: {hi - lo}, mean trips {ln[lo:hi].mean():.1f}")
    for name, key in keys.items():
        for blk in ((None,) if key is None else (block, None)):
            c, u = cost(ordered(lo, hi, key, blk))
            print(f"{name:<46} {'inside blocks of %d' % blk if blk else 'whole queue':<24} {c:8.0f} instructions per wave, lane use {u:.3f}")
