"""The Node.js host (path-tracer_amd/js): host logic against the reference's golden vectors, addon loading,
and -- on the GPU box -- the same render through node + N-API as through the Python host."""
import json
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, "path-tracer_amd", "js")
NODE = shutil.which("node")
needs_node = pytest.mark.skipif(NODE is None, reason="node is not installed")


@needs_node
def test_js_host_logic_matches_reference_fixtures():
    r = subprocess.run([NODE, "check-host.js"], cwd=JS, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr


@needs_node
def test_addon_loads_and_exports_the_boundary():
    if not os.path.exists(os.path.join(JS, "pt_amd.node")):
        pytest.skip("addon not built (run __graft_entry__.build())")
    code = "const a=require('./pt_amd.node');console.log(JSON.stringify(Object.keys(a).sort()))"
    r = subprocess.run([NODE, "-e", code], cwd=JS, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    names = json.loads(r.stdout)
    for n in ("create", "destroy", "meshUpload", "sceneBuild", "renderInit", "renderStep", "renderRun", "imageRead", "tonemap", "traceRays"):
        assert n in names


@needs_node
def test_addon_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available() or not os.path.exists(os.path.join(JS, "pt_amd.node")):
        pytest.skip("only meaningful on a box without a GPU")
    r = subprocess.run([NODE, "main.js", "--width", "8", "--height", "8"], cwd=JS, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no HIP device" in r.stderr


BOX = "\n".join(["v -0.5 -0.5 0", "v 0.5 -0.5 0", "v 0.5 0.5 0", "v -0.5 0.5 0", "v -0.5 -0.5 1.4", "v 0.5 -0.5 1.4", "v 0.5 0.5 1.4",
                 "v -0.5 0.5 1.4", "f 1 4 3 2", "f 5 6 7 8", "f 1 2 6 5", "f 3 4 8 7", "f 1 5 8 4", "f 2 3 7 6"])


EXT = {"maxNb": 8, "seed": 3, "materials": [{"type": "lambert", "albedo": [0.8, 0.35, 0.05]}, {"type": "mirror", "albedo": [0.9, 0.9, 0.9]},
                                            {"type": "emitter", "emission": [6, 4, 2]}]}


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("ext,pools,comm", [(False, 1, False), (True, 1, False), (False, 2, False), (False, 2, True)])
def test_node_render_equals_python_host_render(pta, device, ext, pools, comm):
    """main.js (JS host -> N-API -> C ABI) and the Python host drive the same library: identical frames, with the
    reference's rules, with the extensions (materials, bounce cut, seed) switched on, and with two path pools."""
    w, h, spp, paths, grid = 64, 48, 8, 3000, 3
    r = subprocess.run([NODE, "main.js", "--width", str(w), "--height", str(h), "--samples", str(spp), "--paths", str(paths),
                        "--pools", str(pools), "--grid", str(grid), "--json", "--out", "/tmp/node_frame.png"] + (["--ext"] if ext else []) +
                       (["--gpus", "1", "--force-comm"] if comm else []),     # the multi-GPU path at world size 1: one RCCL reduce onto rank 0
                       cwd=JS, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["gpus"] == 1 and rep["reduced"] == comm
    scene = pta.init_scene(device)
    box = scene.register_mesh({"file": BOX}); floor = scene.register_mesh({"file": pta.scenes.FLOOR_OBJ})
    lcg = 12345
    for i in range(grid * grid):
        lcg = (lcg * 1664525 + 1013904223) & 0xFFFFFFFF
        scene.instance_mesh(box, [(i % grid) * 2, (i // grid) * 2, -1.3], [0, 0, lcg / 4294967296 * math.pi * 2.0], [1, 1, 1], 1 + i % 4 if ext else 1)
    scene.instance_mesh(floor, [10, 10, -1.3], [0, 0, 0], [5, 5, 5], 0)
    scene.build()
    pt = pta.init_path_tracer({"device": device, "scene": scene, "image": {"width": w, "height": h},
                               "camera": {"lookAt": [10, 10, 0], "position": [15, 15, 5], "fov": 60},
                               "settings": dict({"samples": spp, "numPaths": paths, "pools": pools}, **(EXT if ext else {}))})
    rs = pt.run()
    img = pt.read_image()
    assert rep["steps"] == rs.steps and rep["rays"] == rs.total_rays
    assert rep["sampleCount"] == w * h * spp == img[..., 3].sum()
    assert rep["info"]["nTris"] == 14 and rep["info"]["nObjects"] == grid * grid + 1
    np.testing.assert_allclose(rep["radianceSum"], img[..., 0].astype(np.float64).sum(), rtol=1e-12)   # spp = 8: frames are bit-identical
    disp = pta.init_display({"device": device, "image": {"buffer": pt.get_image_buffer(), "width": w, "height": h}})
    assert rep["firstPixel"] == disp.draw()[0, 0].tolist()
    assert os.path.getsize("/tmp/node_frame.png") > 100


@needs_node
@pytest.mark.gpu
def test_node_debug_view_equals_python_debug_view(pta, device):
    """initDebugView (JS) and init_debug_view (Python) draw the same normal view of the same scene."""
    code = """
const pt = require('./pt-host.js')
const FLOOR = "v -10.0 -10.0 0.0\\nv  10.0 -10.0 0.0\\nv  10.0  10.0 0.0\\nv -10.0  10.0 0.0\\n\\nf 1 2 3\\nf 1 3 4"
const BOX = ["v -0.5 -0.5 0", "v 0.5 -0.5 0", "v 0.5 0.5 0", "v -0.5 0.5 0", "v -0.5 -0.5 1.4", "v 0.5 -0.5 1.4", "v 0.5 0.5 1.4", "v -0.5 0.5 1.4",
             "f 1 4 3 2", "f 5 6 7 8", "f 1 2 6 5", "f 3 4 8 7", "f 1 5 8 4", "f 2 3 7 6"].join("\\n")
;(async () => {
  const device = pt.requestDevice(0), scene = pt.initScene(device)
  const box = scene.registerMesh({file: BOX}), floor = scene.registerMesh({file: FLOOR})
  scene.instanceMesh(box, [9, 9, -1.3], [0, 0, 0.5], [2, 2, 2], 1)
  scene.instanceMesh(floor, [10, 10, -1.3], [0, 0, 0], [5, 5, 5], 0)
  await scene.build()
  const dv = pt.initDebugView({device, scene, image: {width: 40, height: 30}, camera: {position: [15, 15, 5], lookAt: [10, 10, 0]}})
  const rgba = dv.draw(), idx = dv.draw("index")
  let sum = 0, sum2 = 0; for (let i = 0; i < rgba.length; i++) { sum += rgba[i] * (1 + i % 7); sum2 += idx[i] * (1 + i % 5) }
  console.log(JSON.stringify({sum, sum2}))
  device.destroy()
})().catch(e => { console.error(e); process.exit(1) })
"""
    r = subprocess.run([NODE, "-e", code], cwd=JS, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    scene = pta.init_scene(device)
    box = scene.register_mesh({"file": BOX}); floor = scene.register_mesh({"file": pta.scenes.FLOOR_OBJ})
    scene.instance_mesh(box, [9, 9, -1.3], [0, 0, 0.5], [2, 2, 2], 1)
    scene.instance_mesh(floor, [10, 10, -1.3], [0, 0, 0], [5, 5, 5], 0)
    scene.build()
    dv = pta.init_debug_view({"device": device, "scene": scene, "image": {"width": 40, "height": 30},
                              "camera": {"position": [15, 15, 5], "lookAt": [10, 10, 0]}})
    a, b = dv.draw().reshape(-1).astype(np.int64), dv.draw("index").reshape(-1).astype(np.int64)
    i = np.arange(a.size)
    # rays are built in f32 on both hosts; a pixel whose ray grazes an edge may resolve differently in the last bit
    assert abs(rep["sum"] - int((a * (1 + i % 7)).sum())) <= 255 * 7 * 4
    assert abs(rep["sum2"] - int((b * (1 + i % 5)).sum())) <= 255 * 5 * 4


@needs_node
@pytest.mark.gpu
def test_node_tolerance_mode_and_check_stats():
    """requestDevice(id, {traversal: "tolerance", checkPeriod: 1}) (pt_set_traversal_mode / pt_set_traversal_check through the addon):
    run() reports every ray re-walked exactly, none beyond the tolerance, no reference hit lost; an unknown mode throws."""
    code = """
const pt = require('./pt-host.js')
const BOX = ["v -0.5 -0.5 0", "v 0.5 -0.5 0", "v 0.5 0.5 0", "v -0.5 0.5 0", "v -0.5 -0.5 1.4", "v 0.5 -0.5 1.4", "v 0.5 0.5 1.4", "v -0.5 0.5 1.4",
             "f 1 4 3 2", "f 5 6 7 8", "f 1 2 6 5", "f 3 4 8 7", "f 1 5 8 4", "f 2 3 7 6"].join("\\n")
const FLOOR = "v -10.0 -10.0 0.0\\nv  10.0 -10.0 0.0\\nv  10.0  10.0 0.0\\nv -10.0  10.0 0.0\\n\\nf 1 2 3\\nf 1 3 4"
;(async () => {
  let threw = false
  try { pt.requestDevice(0, {traversal: "fastest"}) } catch (e) { threw = true }
  const out = {threw}
  for (const traversal of ["reference", "tolerance"]) {
    const device = pt.requestDevice(0, {traversal, checkPeriod: 1}), scene = pt.initScene(device)
    const box = scene.registerMesh({file: BOX}), floor = scene.registerMesh({file: FLOOR})
    for (let i = 0; i < 9; i++) scene.instanceMesh(box, [8 + (i % 3), 8 + Math.floor(i / 3), -1.3], [0, 0, 0.3 * i], [0.6, 0.6, 0.6], 1)
    scene.instanceMesh(floor, [10, 10, -1.3], [0, 0, 0], [5, 5, 5], 0)
    await scene.build()
    const tracer = pt.initPathTracer({device, scene, image: {width: 64, height: 48}, camera: {position: [15, 15, 5], lookAt: [10, 10, 0], fov: 60}, settings: {samples: 16, numPaths: 3000}})
    const rs = await tracer.run()
    out[traversal] = {rays: rs.totalRays, samples: rs.totalSamples, checked: rs.checkedRays, mismatched: rs.mismatchedRays, beyond: rs.beyondToleranceRays, lost: rs.lostHitRays, rel: rs.maxRelDistDiff}
    device.destroy()
  }
  console.log(JSON.stringify(out))
})().catch(e => { console.error(e); process.exit(1) })
"""
    r = subprocess.run([NODE, "-e", code], cwd=JS, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["threw"] and rep["reference"]["checked"] == 0
    t = rep["tolerance"]
    assert t["samples"] == 64 * 48 * 16 and t["checked"] == t["rays"] > 0
    assert t["mismatched"] <= 1e-3 * t["rays"] and t["beyond"] == 0 and t["lost"] == 0 and t["rel"] <= 1e-6
