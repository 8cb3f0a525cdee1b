#!/usr/bin/env python3
"""A/B of the extend-kernel variants on identical work inside one process (interleaved rounds, median and min).

    python tools/extend_ab.py --config grid --warm-steps 12 --rounds 5 --reps 3 [--variants 0,1,2]

Takes `warm-steps` wavefront iterations of the bench workload so that the ray queue holds the steady-state mix of
camera and bounce rays, then replays the extend kernel of every variant on that same queue (pt_extend_replay) and
checks that all variants leave identical hit_obj / hit_tri."""
import argparse
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="grid")
    ap.add_argument("--warm-steps", type=int, default=12)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--variants", default="")
    ap.add_argument("--paths", type=int, default=1_000_000)
    args = ap.parse_args()
    pta = importlib.import_module("path-tracer_amd")
    factory, kwargs, w, h, spp, _ = bench.CONFIGS[args.config]
    desc = getattr(pta.scenes, factory)(**kwargs)
    dev = pta.request_device(0)
    scene = pta.scenes.populate(pta.init_scene(dev), desc); scene.build()
    pt = pta.init_path_tracer({"device": dev, "scene": scene, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": {"samples": spp, "numPaths": args.paths}})
    for _ in range(args.warm_steps):
        st = pt.step()
    name = C.create_string_buffer(200)
    nvar = dev.lib.pt_extend_variants(dev.ctx, 0, name, 200)
    variants = [int(v) for v in args.variants.split(",")] if args.variants else list(range(nvar))
    times = {v: [] for v in variants}
    ref = None
    for rnd in range(args.rounds):
        for v in variants:
            us = C.c_double()
            dev.call("pt_extend_replay", v, args.reps, C.byref(us))
            times[v].append(us.value)
            if rnd == 0:
                got = (pt.field("hit_obj").copy(), pt.field("hit_tri").copy())
                if ref is None:
                    ref = got
                assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1]), f"variant {v} changes results"
    print(f"# {args.config}: {scene.info().n_tris} tris, queue after {args.warm_steps} steps: camera {st.n_camera} material {st.n_material} rays {st.n_rays}")
    rows = []
    for v in variants:
        dev.lib.pt_extend_variants(dev.ctx, v, name, 200)
        t = np.array(times[v])
        rows.append({"variant": v, "name": name.value.decode(), "median_us": round(float(np.median(t)), 2), "min_us": round(float(t.min()), 2),
                     "Mrays_s": round(st.n_rays / np.median(t), 1)})
        print(f"{v:>3} {rows[-1]['name'][:60]:<60} median {rows[-1]['median_us']:>9.2f} us   min {rows[-1]['min_us']:>9.2f} us   {rows[-1]['Mrays_s']:>8.1f} Mrays/s")
    print(json.dumps({"config": args.config, "rays": int(st.n_rays), "rows": rows}))
    dev.destroy()


if __name__ == "__main__":
    main()
