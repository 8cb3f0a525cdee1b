#!/usr/bin/env python3
"""Distribution of traversal trips per ray and per wave (max over 64 consecutive queue entries) for the steady-state
ray queue of a bench config: how much of a wave's time is spent waiting for its longest ray."""
import argparse, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="grid"); ap.add_argument("--warm-steps", type=int, default=200)
a = ap.parse_args()
pta = importlib.import_module("path-tracer_amd")
factory, kwargs, w, h, spp, _ = bench.CONFIGS[a.config]
desc = getattr(pta.scenes, factory)(**kwargs)
dev = pta.request_device(0)
scene = pta.scenes.populate(pta.init_scene(dev), desc); scene.build()
pt = pta.init_path_tracer({"device": dev, "scene": scene, "image": {"width": w, "height": h}, "camera": desc["camera"], "settings": {"samples": spp}})
for _ in range(a.warm_steps):
    pt.step(False)
st = pt.step()
q = np.concatenate([pt.queue(0), pt.queue(1)])
o = pt.field("path_o")[q, :3]; d = pt.field("path_d")[q, :3]
_, _, _, cn = scene.trace_rays(o, d, counts=True)
trips = cn.sum(1)
n = len(trips) // 64 * 64
wmax = trips[:n].reshape(-1, 64).max(1); wmean = trips[:n].reshape(-1, 64).mean(1)
pc = [50, 90, 99, 99.9, 100]
print(f"{a.config}: rays {len(trips)} trips/ray mean {trips.mean():.1f} pct{pc} {np.percentile(trips, pc)}")
print(f"per wave: mean of max {wmax.mean():.1f} pct{pc} {np.percentile(wmax, pc)}  lane utilisation {wmean.sum() / wmax.sum():.3f}")
print(f"sum of wave max {wmax.sum()}  longest wave / mean wave {wmax.max() / wmax.mean():.1f}")
dev.destroy()
