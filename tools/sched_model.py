#!/usr/bin/env python3
"""Model of the kind-scheduled walk (extend variant 4) on recorded per-ray operation sequences.

    tools/sched_model.py <kinds.npz> [n_waves]

<kinds.npz> holds, for the rays of one extend launch in queue order, the kind of every trip of the oracle's walk
(orc_trace_kinds: bit 0 inner node, bit 1 triangle, bit 2 enter an object, bit 3 "this trip first leaves an object"); it is written
by tools/sched_model_record.py.  A wave = 64 consecutive rays.  Every pass the wave runs ONE body -- inner, triangle or switch
(entering and leaving share a body) -- for the lanes whose next operation it is; the model adds up the vector instructions of the
passes (body sizes from the kernel's ISA: 116 / 112 / 182, + 12 per pass for the choice) under different rules for choosing the
body, next to the unscheduled loop (every body some lane needs, every trip: 95 / 88 / 131 / 134).  It counts issued instructions only;
on the GPU about half of a change in instruction count shows up as time (profiles/r02_valu_sensitivity_sphere.txt)."""
import sys

import numpy as np

COST = np.array([116, 112, 182]); LOOP = 12
BASE = {"inner": 95, "leaf": 88, "enter": 131, "leave": 134, "loop": 10}


def micro_ops(kinds, ln):
    """-> per ray the list of operations: 0 inner, 1 triangle, 2 switch (enter or leave)."""
    out = []
    for i in range(len(ln)):
        ops = []
        for k in kinds[i, :min(ln[i], kinds.shape[1])]:
            if k & 8: ops.append(2)
            ops.append(0 if k & 1 else 1 if k & 2 else 2)
        out.append(ops)
    return out


def weights(wi, wl, ws):
    w = np.array([wi, wl, ws])
    def pick(cnt, state):
        v = cnt * w
        return 0 if v[0] >= v[1] and v[0] >= v[2] else (1 if v[1] >= v[2] else 2)      # the kernel's tie rule
    return pick


def sticky(wi, wl, ws, keep):
    """as weights, but the body of the last pass keeps going while it still serves `keep` lanes"""
    base = weights(wi, wl, ws)
    def pick(cnt, state):
        last = state.get("last")
        op = last if last is not None and cnt[last] >= keep else base(cnt, state)
        state["last"] = op
        return op
    return pick


def waiting(wi, wl, ws, age_w):
    """weights + the passes the lanes of a kind have been waiting"""
    w = np.array([wi, wl, ws], float)
    def pick(cnt, state):
        age = state.setdefault("age", np.zeros(3))
        v = cnt * w + age_w * age * (cnt > 0)
        op = int(np.argmax(v))
        age += cnt; age[op] = 0
        return op
    return pick


def run(seqs, pick):
    L = max(len(s) for s in seqs) + 1
    ops = np.full((64, L), 3, np.int8)
    for l, s in enumerate(seqs): ops[l, :len(s)] = s
    pos = np.zeros(64, np.int64); lanes = np.arange(64); tot = 0; passes = 0; useful = 0; state = {}
    while True:
        cur = ops[lanes, pos]
        cnt = np.bincount(cur, minlength=4)[:3]
        if cnt.sum() == 0: break
        op = pick(cnt, state)
        pos += cur == op
        tot += COST[op] + LOOP; passes += 1; useful += cnt[op] * COST[op]
    return tot, passes, useful


def unscheduled(kinds, ln):
    tot = 0
    for t in range(min(ln.max(), kinds.shape[1])):
        k = kinds[ln > t, t]
        tot += BASE["loop"] + BASE["leave"] * bool((k & 8).any()) + BASE["enter"] * bool((k & 4).any()) + BASE["leaf"] * bool((k & 2).any()) + BASE["inner"] * bool((k & 1).any())
    return tot


def main():
    z = np.load(sys.argv[1]); kinds, ln = z["kinds"], z["ln"]
    nw = len(ln) // 64; n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    ws = np.random.default_rng(0).choice(nw, size=min(n, nw), replace=False)
    per_wave = [micro_ops(kinds[w * 64:(w + 1) * 64], ln[w * 64:(w + 1) * 64]) for w in ws]
    base = sum(unscheduled(kinds[w * 64:(w + 1) * 64], ln[w * 64:(w + 1) * 64]) for w in ws)
    print(f"{len(ws)} waves of {sys.argv[1]}; unscheduled loop: {base / len(ws):.0f} vector instructions per wave")
    rules = [("weights 12:9:5 (by cost)", weights(12, 9, 5)), ("weights 6:9:5 (shipped)", weights(6, 9, 5)), ("weights 4:9:5", weights(4, 9, 5)),
             ("weights 3:9:5", weights(3, 9, 5)), ("weights 8:12:6", weights(8, 12, 6)), ("weights 1:1:1 (most lanes)", weights(1, 1, 1)),
             ("6:9:5, body keeps going while >= 24 lanes", sticky(6, 9, 5, 24)), ("6:9:5, keeps going while >= 40 lanes", sticky(6, 9, 5, 40)),
             ("12:9:5 + 0.25 x waiting", waiting(12, 9, 5, 0.25)), ("12:9:5 + 1 x waiting", waiting(12, 9, 5, 1.0)), ("6:9:5 + 0.25 x waiting", waiting(6, 9, 5, 0.25))]
    for name, rule in rules:
        r = [run(s, rule) for s in per_wave]
        tot = sum(x[0] for x in r); passes = sum(x[1] for x in r); useful = sum(x[2] for x in r)
        print(f"{name:<48} {tot / len(ws):>8.0f} per wave ({tot / base:5.3f} of unscheduled), {passes / len(ws):6.1f} passes, lane use {useful / 64 / (tot - LOOP * passes):.3f}")


if __name__ == "__main__":
    main()
