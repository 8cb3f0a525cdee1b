#!/usr/bin/env python3
"""Experiment: K independent path pools on one GPU, each on its own stream and host thread, each taking 1/K of the pixel
tickets (multigpu.strong_share), so that one pool's compaction launches fill the tail of another pool's extend launch.
    python tools/two_pool.py --pools 2 --paths 1000000"""
import argparse, importlib, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pta = importlib.import_module("path-tracer_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--pools", type=int, default=2)
ap.add_argument("--paths", type=int, default=1_000_000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--spp", type=int, default=64)
args = ap.parse_args()
w, h, spp = 1920, 1080, args.spp
desc = pta.scenes.cornell_box()
pools = []
for k in range(args.pools):
    st = torch.cuda.Stream()
    dev = pta.request_device(0, stream=st.cuda_stream)
    scene = pta.scenes.populate(pta.init_scene(dev), desc); scene.build()
    image = torch.zeros(h * w * 4, dtype=torch.float32, device="cuda")
    settings = pta.multigpu.strong_share(k, args.pools, w, h, spp, args.paths)
    settings["imageDev"] = image.data_ptr()
    pt = pta.init_path_tracer({"device": dev, "scene": scene, "image": {"width": w, "height": h}, "camera": desc["camera"], "settings": settings})
    pools.append((st, dev, scene, image, pt))

def render():
    res = [None] * len(pools)
    def work(i):
        pools[i][4].reset(); res[i] = pools[i][4].run()
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(pools))]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return res
render()
t0 = time.perf_counter()
rays = 0
for _ in range(args.reps):
    rays += sum(r.total_rays for r in render())
dt = time.perf_counter() - t0
tot = sum(p[3] for p in pools).cpu().reshape(h, w, 4)
assert bool((tot[..., 3] == spp).all())
print(f"pools {args.pools} x {args.paths} slots: {rays / dt / 1e6:.1f} Mrays/s, {dt / args.reps * 1e3:.2f} ms per render")
