#!/usr/bin/env python3
"""Builds one bench scene (scene.build(): per-mesh LBVH, pack, TLAS, objects) and exits: run under
`rocprofv3 --kernel-trace --stats` to see where the build's device time goes (tools/prof_summary.py prints the table).
    python tools/build_trace.py --config grid [--repeat 3]"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="grid")
ap.add_argument("--repeat", type=int, default=3)
args = ap.parse_args()
pta = importlib.import_module("path-tracer_amd")
factory, kwargs, w, h, spp, _ = bench.CONFIGS[args.config]
desc = getattr(pta.scenes, factory)(**kwargs)
for r in range(args.repeat):
    dev = pta.request_device(0)
    t0 = time.perf_counter()
    scene = pta.scenes.populate(pta.init_scene(dev), desc)
    t1 = time.perf_counter()
    scene.build()
    t2 = time.perf_counter()
    print(f"{args.config}: {scene.info().n_tris} triangles, upload {1e3 * (t1 - t0):.1f} ms, build {1e3 * (t2 - t1):.1f} ms")
    dev.destroy()
