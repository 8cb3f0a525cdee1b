#!/usr/bin/env python3
"""Lane use of the kind-scheduled traversal loop: one counting render of a bench config (one pool), passes per body and lane visits.

    python tools/pass_counts.py --config grid [--fast] [--paths 2000000] [--spp 8]"""
import argparse
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="grid"); ap.add_argument("--fast", action="store_true")
ap.add_argument("--paths", type=int, default=2_000_000); ap.add_argument("--spp", type=int, default=8)
args = ap.parse_args()
pta = importlib.import_module("path-tracer_amd")
factory, kwargs, w, h, spp, _ = bench.CONFIGS[args.config]
desc = getattr(pta.scenes, factory)(**kwargs)
dev = pta.request_device(0)
if args.fast:
    dev.call("pt_set_traversal_mode", pta.capi.PT_TRAVERSAL_WIDE)
scene = pta.scenes.populate(pta.init_scene(dev), desc); scene.build()
pt = pta.init_path_tracer({"device": dev, "scene": scene, "image": {"width": w, "height": h}, "camera": desc["camera"],
                           "settings": dict({"samples": args.spp, "numPaths": args.paths}, **desc.get("settings", {}))})
rs = pt.run(count_bytes=True)
out = (C.c_int64 * 12)()
dev.call("pt_extend_pass_counts", out)
p_i, p_w, p_l, p_s, v_n, v_l, v_e, v_x = list(out)[:8]
hist = list(out)[8:]
waves = rs.total_rays / 64
print(f"# {args.config}{' tolerance mode' if args.fast else ''}: {rs.total_rays} rays in {rs.steps} steps, 1 pool of {args.paths}")
print(f"lane visits per ray: nodes {v_n / rs.total_rays:.2f}  triangles {v_l / rs.total_rays:.2f}  enter {v_e / rs.total_rays:.3f}  leave {v_x / rs.total_rays:.3f}")
print(f"passes per wave:     inner {p_i / waves:.1f}  wide {p_w / waves:.1f}  triangle {p_l / waves:.1f}  switch {p_s / waves:.1f}")
node_p = p_i + p_w
print(f"lane use: node bodies {v_n / (64 * max(node_p, 1)):.3f}   triangle body {v_l / (64 * max(p_l, 1)):.3f}   switch body {(v_e + v_x) / (64 * max(p_s, 1)):.3f}")
print("node-body passes by lanes served (<=16, <=32, <=48, <=64):", [round(x / max(sum(hist), 1), 3) for x in hist])
dev.destroy()
