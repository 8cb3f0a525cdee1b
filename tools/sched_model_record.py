#!/usr/bin/env python3
"""Record, with the CPU oracle, the rays of one extend launch of a bench config and the kind of every trip of their walks
(input of tools/sched_model.py and tools/queue_order_model.py).

    tools/sched_model_record.py <sphere|cornell|grid|instanced> <out.npz> [slots] [steps]

Test infrastructure: uses tests/orc.py (the oracle)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import orc  # noqa: E402

pta = importlib.import_module("path-tracer_amd")
name, out = sys.argv[1], sys.argv[2]
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
desc, w, h, spp = {"sphere": (pta.scenes.bunny_scale_scene, 1920, 1080, 256), "cornell": (pta.scenes.cornell_box, 1920, 1080, 64),
                   "grid": (pta.scenes.grid_scene, 3840, 2160, 32), "instanced": (pta.scenes.instanced_scene, 3840, 2160, 32)}[name]
desc = desc()
s = orc.OrcScene(desc)
tb = int(0.42 * w * h)                                    # tickets from the middle of the image
r = orc.OrcRender(s, w, h, desc["camera"], spp, slots, ticket_begin=tb, ticket_end=tb + 50_000_000)
for _ in range(steps):
    r.step_like_device()
r.phase(r.QUEUES_CAMERA); r.phase(r.MATERIAL)
cam, mat = r.queue(0), r.queue(1)
q = np.concatenate([cam, mat])
o = r.field("path_o")[q, :3]; d = r.field("path_d")[q, :3]
kinds, ln = s.trace_kinds(o, d, cap=1024)
np.savez_compressed(out, kinds=kinds[:, :min(1024, ln.max())], ln=ln, ncam=len(cam), slot=q, o=o, d=d, prev_obj=r.field("hit_obj")[q],
                    pixel=r.field("pixel_index")[q])
print(name, "camera", len(cam), "bounce", len(mat), "mean trips", ln.mean(), "max", ln.max())
