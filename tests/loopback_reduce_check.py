"""Run by test_reduce_semantics_at_world_size_two in a process of its own with PT_COMM_LOOPBACK=1 (the backend is chosen once per process):
two contexts on the one GPU as ranks 0 and 1 of a pt_comm, each with its own scene replica and half of the pixel tickets (SURVEY 8e's
partition).  A reduce in the middle of the render, then one at the end: the root must end with exactly spp samples in every pixel and
the sum of the two shares' radiance (nothing counted twice), the other rank with an empty image."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
assert os.environ.get("PT_COMM_LOOPBACK") == "1"
pta = importlib.import_module("path-tracer_amd")
import orc  # noqa: E402

desc = pta.scenes.cornell_box()
w, h, spp, paths, world = 64, 36, 32, 3000, 2
devs, pts = [], []
for rank in range(world):
    dev = pta.request_device(0)
    scene = pta.scenes.populate(pta.init_scene(dev), desc); scene.build()
    st = dict(pta.multigpu.strong_share(rank, world, w, h, spp, paths))
    pts.append(pta.init_path_tracer({"device": dev, "scene": scene, "image": {"width": w, "height": h}, "camera": desc["camera"], "settings": st}))
    devs.append(dev)
comm = pta.multigpu.LibComm(devs)
for pt in pts:
    pt.run(max_steps=5)                       # part of the render; open tickets are flushed by the reduce
comm.reduce_image(0)
mid0, mid1 = pts[0].read_image().copy(), pts[1].read_image().copy()
assert 0 < mid0[..., 3].sum() < w * h * spp and not mid1.any(), "after a reduce the samples live on the root only"
for pt in pts:
    pt.run()
comm.reduce_image(0)
img0, img1 = pts[0].read_image().astype(np.float64), pts[1].read_image()
assert not img1.any()
np.testing.assert_array_equal(img0[..., 3], np.full((h, w), spp))            # a second in-place sum of intact buffers would give more than spp
o = orc.OrcScene(desc)
ref = np.zeros((h, w, 4), np.float64)
for rank in range(world):
    st = pta.multigpu.strong_share(rank, world, w, h, spp, paths)
    orr = orc.OrcRender(o, w, h, desc["camera"], spp, paths, slot_base=st["slotBase"], ticket_begin=st["ticketBegin"], ticket_end=st["ticketEnd"]); orr.run()
    ref += orr.image()
np.testing.assert_array_equal(img0[..., 3], ref[..., 3])
np.testing.assert_allclose(img0[..., :3], ref[..., :3], rtol=1e-5, atol=1e-6)
comm.reduce_image(0)                                                              # and a third one changes nothing
np.testing.assert_array_equal(pts[0].read_image().astype(np.float64), img0)
comm.destroy()
for d in devs:
    d.destroy()
print("loopback reduce ok")
