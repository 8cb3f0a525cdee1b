#!/usr/bin/env python3
"""Would repacking the unfinished rays of a 256-thread workgroup into fewer waves (through LDS, at a workgroup barrier every P passes)
pay on top of the kind-scheduled walk?  Model on recorded rays (tools/sched_model_record.py), pass costs of tools/sched_model.py.

    tools/repack_model.py <rec.npz> [groups]

Counts issued vector instructions only: a repack costs `move` instructions per wave that takes part (ray state + LDS stack column
out and in); waves that reach the barrier early wait, which costs time the model does not see."""
import sys

import numpy as np

from sched_model import COST, LOOP, micro_ops

z = np.load(sys.argv[1]); kinds, ln = z["kinds"], z["ln"]
n_groups = int(sys.argv[2]) if len(sys.argv) > 2 else 60
W = np.array([6, 9, 5])


def pick(cnt):
    v = cnt * W
    return 0 if v[0] >= v[1] and v[0] >= v[2] else (1 if v[1] >= v[2] else 2)


def run_wave(seqs, max_passes):
    """seqs: list (<= 64) of op lists; runs up to max_passes passes; returns (instructions, remaining op lists of unfinished lanes)."""
    seqs = [s for s in seqs]
    pos = [0] * len(seqs); tot = 0; passes = 0
    while passes < max_passes:
        cnt = np.zeros(3, int)
        for l, s in enumerate(seqs):
            if pos[l] < len(s): cnt[s[pos[l]]] += 1
        if cnt.sum() == 0: break
        op = pick(cnt)
        for l, s in enumerate(seqs):
            if pos[l] < len(s) and s[pos[l]] == op: pos[l] += 1
        tot += COST[op] + LOOP; passes += 1
    return tot, [s[pos[l]:] for l, s in enumerate(seqs) if pos[l] < len(s)]


def workgroup(rays, period, move):
    waves = [rays[i:i + 64] for i in range(0, len(rays), 64)]
    tot = 0
    if period is None:
        return sum(run_wave(w, 10 ** 9)[0] for w in waves)
    while True:
        left = []
        for w in waves:
            t, rest = run_wave(w, period); tot += t; left += rest
        if not left: return tot
        new = [left[i:i + 64] for i in range(0, len(left), 64)]
        if len(new) < len(waves): tot += move * (len(waves) + len(new))       # repack only when a wave is saved: old waves write, new waves read
        else: new = [w2 for w2 in ([r for r in w if True] for w in waves)] and new
        waves = new


rng = np.random.default_rng(0)
ng = len(ln) // 256
groups = rng.choice(ng, size=min(n_groups, ng), replace=False)
sets = [micro_ops(kinds[g * 256:(g + 1) * 256], ln[g * 256:(g + 1) * 256]) for g in groups]
base = sum(workgroup(r, None, 0) for r in sets)
print(f"{len(groups)} workgroups of 256 rays; no repacking: {base / len(groups):.0f} vector instructions per workgroup")
for period in (8, 16, 24, 32, 48):
    for move in (100, 200):
        t = sum(workgroup(r, period, move) for r in sets)
        print(f"repack every {period:>2} passes, {move} instructions per wave and repack: {t / len(groups):8.0f}  ({t / base:.3f})")
