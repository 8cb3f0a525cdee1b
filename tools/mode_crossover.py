#!/usr/bin/env python3
"""Where the tolerance-mode (4-wide) walk starts to pay: the displaced grid of configs[3] at several sizes, 3840x2160, both traversal
modes, the hosts' default pools.  PT_TRAVERSAL_AUTO's threshold (256 MB of binary nodes = 4.2 M triangles) comes from this table.

    python tools/mode_crossover.py [--spp 16]"""
import argparse
import importlib
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--sides", default="500,707,1000,1414,1600,2236")
args = ap.parse_args()
pta = importlib.import_module("path-tracer_amd")
w, h = 3840, 2160
pools, paths = pta.host.default_pool_shape(w, h)
print(f"# displaced grid of n x n quads at {w}x{h}, {args.spp} spp, {pools} pools of {paths} slots; Mrays/s")
print(f"{'n':>6} {'triangles':>10} {'node MB':>8} {'reference':>10} {'tolerance':>10} {'ratio':>6}")
for n in [int(x) for x in args.sides.split(",")]:
    desc = pta.scenes.grid_scene(n, n)
    res = {}
    for mode in (pta.capi.PT_TRAVERSAL_REFERENCE, pta.capi.PT_TRAVERSAL_WIDE):
        dev = pta.request_device(0)
        dev.call("pt_set_traversal_mode", mode)
        scene = pta.scenes.populate(pta.init_scene(dev), desc); scene.build()
        pt = pta.init_path_tracer({"device": dev, "scene": scene, "image": {"width": w, "height": h}, "camera": desc["camera"],
                                   "settings": {"samples": args.spp, "numPaths": paths, "pools": pools}})
        pt.run()
        best = 0.0
        for _ in range(2):
            pt.reset()
            dev.call("pt_sync"); t0 = time.perf_counter()
            rs = pt.run()
            dt = time.perf_counter() - t0
            best = max(best, rs.total_rays / dt / 1e6)
        res[mode] = best
        tris = scene.info().n_tris
        pt.destroy(); dev.destroy()
    a, b = res[pta.capi.PT_TRAVERSAL_REFERENCE], res[pta.capi.PT_TRAVERSAL_WIDE]
    print(f"{n:>6} {tris:>10} {tris * 64 / 2**20:>8.0f} {a:>10.0f} {b:>10.0f} {b / a:>6.2f}", flush=True)
