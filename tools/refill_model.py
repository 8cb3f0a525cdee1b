#!/usr/bin/env python3
"""Would persistent waves with lane refill pay on top of the kind-scheduled walk?  Model on recorded rays
(tools/sched_model_record.py): a wave owns 64 * R consecutive queue entries (no shared counter), lane l walks entries l, l + 64, ...;
finishing a ray and fetching the next is one more body ("retire": finish + logic + count records + next ray, `cost` vector
instructions) that runs when `thresh` lanes wait for it or nothing else is left.  Compared with R one-ray-per-lane waves.

    tools/refill_model.py <rec.npz> <R> <retire cost> [groups]"""
import sys, numpy as np
from sched_model import micro_ops, COST, LOOP
z = np.load(sys.argv[1]); kinds, ln = z["kinds"], z["ln"]
R = int(sys.argv[2]); RET = int(sys.argv[3]); n_groups = int(sys.argv[4]) if len(sys.argv) > 4 else 60
HT = RET  # head+tail cost per batch in the non-persistent kernel
W = np.array([6, 9, 5])
def pick_w(cnt):
    v = cnt[:3] * W
    return 0 if v[0] >= v[1] and v[0] >= v[2] else (1 if v[1] >= v[2] else 2)
def run_wave(seqs):   # non-persistent: one ray per lane
    L = max(len(s) for s in seqs) + 1
    ops = np.full((64, L), 4, np.int8)
    for l, s in enumerate(seqs): ops[l, :len(s)] = s
    pos = np.zeros(64, np.int64); lanes = np.arange(64); tot = 0
    while True:
        cur = ops[lanes, pos]; cnt = np.bincount(cur, minlength=5)
        if cnt[:3].sum() == 0: break
        op = pick_w(cnt); pos += cur == op; tot += COST[op] + LOOP
    return tot + HT
def run_refill(seqs_per_lane, thresh):   # seqs_per_lane[l] = list of R op lists
    full = []
    for l in range(64):
        s = []
        for k, ray in enumerate(seqs_per_lane[l]):
            s += ray + [3]          # 3 = retire (tail of this ray + head of the next)
        full.append(s)
    L = max(len(s) for s in full) + 1
    ops = np.full((64, L), 4, np.int8)
    for l, s in enumerate(full): ops[l, :len(s)] = s
    pos = np.zeros(64, np.int64); lanes = np.arange(64); tot = HT // 2   # the first head
    while True:
        cur = ops[lanes, pos]; cnt = np.bincount(cur, minlength=5)
        if cnt[:4].sum() == 0: break
        if cnt[3] >= thresh or cnt[:3].sum() == 0: op = 3; c = RET
        else:
            if cnt[:3].sum() == 0: op = 3; c = RET
            else: op = pick_w(cnt); c = COST[op]
        pos += cur == op; tot += c + LOOP
    return tot
rng = np.random.default_rng(0)
ngr = len(ln) // (64 * R)
groups = rng.choice(ngr, size=min(n_groups, ngr), replace=False)
base = 0; ref = {t: 0 for t in (8, 16, 24, 32, 48)}
for g in groups:
    lo = g * 64 * R
    rays = micro_ops(kinds[lo:lo + 64 * R], ln[lo:lo + 64 * R])
    for b in range(R): base += run_wave(rays[b * 64:(b + 1) * 64])
    per_lane = [[rays[b * 64 + l] for b in range(R)] for l in range(64)]
    for t in ref: ref[t] += run_refill(per_lane, t)
print(f"R={R} retire cost {RET}: non-persistent {base / len(groups) / R:.0f} per 64 rays")
for t, v in ref.items(): print(f"  refill when >= {t} lanes wait: {v / len(groups) / R:.0f} per 64 rays ({v / base:.3f})")
