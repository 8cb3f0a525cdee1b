
import importlib, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
pta = importlib.import_module("path-tracer_amd")
dev = pta.request_device(0)
for name, desc in (("cornell", pta.scenes.cornell_box()), ("sphere", pta.scenes.bunny_scale_scene()), ("instanced", pta.scenes.instanced_scene()), ("grid", pta.scenes.grid_scene())):
    s = pta.scenes.populate(pta.init_scene(dev), desc); s.build()
    rng = np.random.default_rng(1)
    n = 400000
    cam = np.asarray(desc["camera"]["position"], np.float32)
    look = np.asarray(desc["camera"]["lookAt"], np.float32)
    d = (look - cam)[None, :] + rng.normal(size=(n, 3)).astype(np.float32) * np.linalg.norm(look - cam) * 0.4
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.broadcast_to(cam, (n, 3)).copy()
    dev.call("pt_set_traversal_mode", 0)
    t0 = time.time(); a = s.trace_rays(o, d, counts=True); t_a = time.time() - t0
    dev.call("pt_set_traversal_mode", 1)
    t0 = time.time(); b = s.trace_rays(o, d, counts=True); t_b = time.time() - t0
    same = (a[0] == b[0]) & (a[1] == b[1])
    hit = a[0] >= 0
    rel = np.abs(a[2] - b[2]) / np.maximum(np.abs(a[2]), 1e-6)
    print(name, "hit frac", hit.mean(), "identical", same.mean(), "max rel dist diff", rel.max(), "mismatch dist diff", (rel[~same].max() if (~same).any() else 0),
          "inner visits exact", a[3][:, 0].mean(), "wide visits", b[3][:, 0].mean() / 2, "leaf", a[3][:, 1].mean(), b[3][:, 1].mean())
    dev.call("pt_set_traversal_mode", 0)
dev.destroy()
