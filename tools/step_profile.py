#!/usr/bin/env python3
"""Per-step queue sizes and (host-synchronised) step times over one render of the bench configuration: shows the
ramp-down at the end of a render, when the pixel tickets have run out and the pool drains.
    python tools/step_profile.py > profiles/<name>.txt      (each time includes ~25 us of host synchronisation)"""
import importlib, sys, time
sys.path.insert(0, '/root/repo')
import torch
pta = importlib.import_module("path-tracer_amd")
desc = pta.scenes.cornell_box()
dev = pta.request_device(0)
scene = pta.scenes.populate(pta.init_scene(dev), desc); scene.build()
pt = pta.init_path_tracer({"device": dev, "scene": scene, "image": {"width": 1920, "height": 1080}, "camera": desc["camera"], "settings": {"samples": 64, "numPaths": 2_000_000}})
rows = []
for k in range(160):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = pt.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rows.append((k, s.n_camera, s.n_material, dt * 1e6))
    if s.n_camera + s.n_material == 0: break
for r in rows:
    if r[0] < 6 or r[0] % 10 == 0 or r[0] > len(rows) - 45: print("%3d cam %8d mat %8d  %.0f us" % r)
