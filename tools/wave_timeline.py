#!/usr/bin/env python3
"""Per-wave timeline of one extend launch (pt_extend_wave_log): when each wave started and ended, on which XCD / CU,
its ray class and trip count.  Prints occupancy over time, the tail, and per-class / per-XCD summaries.

    python tools/wave_timeline.py --config cornell --warm-steps 12 --variants 1,2 [--npz out.npz]"""
import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def summarize(tag, log, bins=20):
    log = log[log[:, 1] != 0]
    t0 = log[:, 0].astype(np.int64); t1 = log[:, 1].astype(np.int64)
    base = t0.min(); s = (t0 - base) / 100.0; e = (t1 - base) / 100.0           # us
    xcc = (log[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
    hw = log[:, 2].astype(np.int64) & 0xffffffff
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7
    cls = (log[:, 3] >> np.uint64(24)).astype(np.int64) & 0xff
    trips = log[:, 3].astype(np.int64) & 0xffffff
    blk = (log[:, 3] >> np.uint64(32)).astype(np.int64)
    total = e.max()
    print(f"## {tag}: {len(log)} waves, launch spans {total:.1f} us; last wave starts at {s.max():.1f} us")
    dur = e - s
    for c, name in ((0, "camera"), (1, "bounce")):
        m = cls == c
        if m.any():
            print(f"   {name:<6} waves {m.sum():>6}  duration mean {dur[m].mean():6.1f} p50 {np.median(dur[m]):6.1f} p99 {np.percentile(dur[m], 99):6.1f} max {dur[m].max():6.1f} us"
                  f"   trips mean {trips[m].mean():6.1f} max {trips[m].max()}   start mean {s[m].mean():6.1f}  end max {e[m].max():6.1f}")
    edges = np.linspace(0, total, bins + 1)
    occ = []
    for a, b in zip(edges[:-1], edges[1:]):
        overlap = np.clip(np.minimum(e, b) - np.maximum(s, a), 0, None).sum() / (b - a)
        occ.append(overlap)
    print("   resident waves over time (mean per bin of %.1f us): " % (total / bins) + " ".join(f"{o:.0f}" for o in occ))
    print("   per XCD: waves / last end (us): " + "  ".join(f"{x}:{(xcc == x).sum()}/{e[xcc == x].max():.0f}" for x in sorted(set(xcc.tolist()))))
    order = np.argsort(e)[-5:]
    print("   last 5 waves to end: " + "; ".join(f"blk {blk[i]} cls {cls[i]} trips {trips[i]} {s[i]:.0f}->{e[i]:.0f}us xcd{xcc[i]} se{se[i]} cu{cu[i]}" for i in order))
    return dict(s=s, e=e, xcc=xcc, cu=cu, se=se, cls=cls, trips=trips, blk=blk)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cornell")
    ap.add_argument("--warm-steps", type=int, default=12)
    ap.add_argument("--variants", default="1,2")
    ap.add_argument("--paths", type=int, default=1_000_000)
    ap.add_argument("--npz", default="")
    ap.add_argument("--fast", action="store_true", help="tolerance-mode traversal (the 4-wide collapse)")
    args = ap.parse_args()
    pta = importlib.import_module("path-tracer_amd")
    factory, kwargs, w, h, spp, _ = bench.CONFIGS[args.config]
    desc = getattr(pta.scenes, factory)(**kwargs)
    dev = pta.request_device(0)
    if args.fast:
        dev.call("pt_set_traversal_mode", pta.capi.PT_TRAVERSAL_WIDE)
    scene = pta.scenes.populate(pta.init_scene(dev), desc); scene.build()
    pt = pta.init_path_tracer({"device": dev, "scene": scene, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": {"samples": spp, "numPaths": args.paths}})
    for _ in range(args.warm_steps):
        st = pt.step()
    print(f"# {args.config}: queue after {args.warm_steps} steps: camera {st.n_camera} material {st.n_material}")
    keep = {}
    for v in [int(x) for x in args.variants.split(",")]:
        us = C.c_double()
        dev.call("pt_extend_replay", v, 2, C.byref(us))          # warm the caches the same way for every variant
        cap = (args.paths + 63) // 64 + 8
        buf = np.zeros((cap, 4), np.uint64); n = C.c_int64()
        dev.call("pt_extend_wave_log", v, buf.ctypes.data_as(C.c_void_p), cap, C.byref(n))
        r = summarize(f"variant {v} (replay mean {us.value:.1f} us)", buf[: n.value])
        keep.update({f"v{v}_{k}": a for k, a in r.items()})
    if args.npz:
        np.savez_compressed(args.npz, **keep)
    dev.destroy()


if __name__ == "__main__":
    main()
