"""The Node-free C++ host (path-tracer_amd/cli): its OBJ parser / TRS matrices / transformed boxes against the golden vectors
produced by the reference's own JS (tests/golden/host_fixtures.json), and its render against the Python host's."""
import json
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "path-tracer_amd", "pt_cli")
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "host_fixtures.json")))


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "path-tracer_amd", "cli")])
    return CLI


def dump(cli, tmp_path, text, trs="0,0,0,0,0,0,1,1,1"):
    f = tmp_path / "m.obj"
    f.write_bytes(text.encode())
    r = subprocess.run([cli, "--dump-host", str(f), trs], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout)


def nan_list(v):
    return np.asarray([np.nan if x is None else x for x in v], dtype=np.float64)


@pytest.mark.parametrize("case", FIX["parseObj"], ids=[c["name"] for c in FIX["parseObj"]])
def test_cli_parse_obj_matches_reference(cli, tmp_path, case):
    got = dump(cli, tmp_path, case["text"])
    assert got["numTriangles"] == case["numTriangles"]
    np.testing.assert_array_equal(nan_list(got["triangleArray"]), nan_list(case["triangleArray"]))       # exact in float64
    np.testing.assert_array_equal(nan_list(got["bounds"]["min"]), nan_list(case["bounds"]["min"]))
    np.testing.assert_array_equal(nan_list(got["bounds"]["max"]), nan_list(case["bounds"]["max"]))


@pytest.mark.parametrize("case", FIX["transforms"], ids=[str(i) for i in range(len(FIX["transforms"]))])
def test_cli_transforms_match_reference(cli, tmp_path, case):
    t = case["transform"]
    trs = ",".join(repr(float(x)) for x in t["position"] + t["rotation"] + t["scale"])
    for box, want in zip(case["boxes"], case["transformedBoxes"]):
        text = "v %r %r %r\nv %r %r %r\n" % (*box["min"], *box["max"])     # a mesh whose bounds are exactly this box
        got = dump(cli, tmp_path, text, trs)
        for key in ("localToWorld", "worldToLocal"):
            g, w = np.asarray(got[key]), np.asarray(case[key])
            np.testing.assert_allclose(g, w, rtol=4e-16, atol=1e-300)       # V8 vs glibc sin/cos: last f64 bit
            np.testing.assert_array_equal(g.astype(np.float32), w.astype(np.float32))
        if np.array_equal(np.asarray(got["localToWorld"]), np.asarray(case["localToWorld"])):
            assert got["worldBox"]["min"] == want["min"] and got["worldBox"]["max"] == want["max"]
        else:
            np.testing.assert_allclose(got["worldBox"]["min"] + got["worldBox"]["max"], want["min"] + want["max"], rtol=1e-15)


def read_png(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(b):
        n = int.from_bytes(b[pos:pos + 4], "big"); typ = b[pos + 4:pos + 8]; data = b[pos + 8:pos + 8 + n]
        assert zlib.crc32(typ + data) == int.from_bytes(b[pos + 8 + n:pos + 12 + n], "big")
        chunks.setdefault(typ, b""); chunks[typ] += data; pos += 12 + n
    w, h = int.from_bytes(chunks[b"IHDR"][:4], "big"), int.from_bytes(chunks[b"IHDR"][4:8], "big")
    raw = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), np.uint8).reshape(h, 1 + 4 * w)
    assert np.all(raw[:, 0] == 0)
    return raw[:, 1:].reshape(h, w, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("pools,comm", [(1, False), (2, False), (2, True)])
def test_cli_render_equals_python_host_render(cli, pta, device, tmp_path, pools, comm):
    """pt_cli (C++ host -> C ABI) renders the reference's floor scene (main.js:33,42-46); the Python host renders the same:
    identical step / ray / sample counts; radiance to the float-add-order tolerance (--progress draws mid-render, which
    adds the unfinished tickets' partial sums early and so reorders the adds), PNG pixels within one 8-bit step.
    comm: the multi-GPU path at world size 1 (--gpus 1 with the communicator forced): pt_comm_create + one RCCL reduce of the
    accumulation buffer onto rank 0, which must leave the frame as it was."""
    w, h, spp, paths = 64, 48, 8, 3000
    out = tmp_path / "frame.png"
    r = subprocess.run([cli, "--width", str(w), "--height", str(h), "--samples", str(spp), "--paths", str(paths), "--pools", str(pools), "--json",
                        "--progress", "3", "--out", str(out), "--gpus", "1"], capture_output=True, text=True, timeout=240,
                       env=dict(os.environ, PT_CLI_FORCE_COMM="1") if comm else None)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["gpus"] == 1 and rep["reduced"] == comm
    desc = pta.scenes.floor_scene()
    s = pta.scenes.populate(pta.init_scene(device), desc); s.build()
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": {"samples": spp, "numPaths": paths, "pools": pools}})
    rs = pt.run()
    img = pt.read_image()
    assert (rep["steps"], rep["rays"], rep["nTris"], rep["nObjects"]) == (rs.steps, rs.total_rays, 2, 1)
    assert rep["sampleCount"] == w * h * spp == img[..., 3].sum() and rep["frames"] >= 2
    np.testing.assert_allclose(rep["radianceSum"], img[..., 0].astype(np.float64).sum(), rtol=1e-6)
    disp = pta.init_display({"device": device, "image": {"buffer": pt.get_image_buffer(), "width": w, "height": h}})
    assert np.abs(read_png(str(out)).astype(np.int32) - disp.draw().astype(np.int32)).max() <= 1


@pytest.mark.gpu
def test_cli_tolerance_mode_option(cli, tmp_path):
    """pt_cli --traversal tolerance / auto: the floor scene (two triangles) renders to the same counts as the reference walk -- the wide
    walk of a two-triangle mesh has nothing to tie -- and an unknown mode is refused."""
    reps = []
    for mode in ("reference", "tolerance", "auto"):
        r = subprocess.run([cli, "--width", "48", "--height", "32", "--samples", "8", "--paths", "2000", "--json", "--traversal", mode], capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout + r.stderr
        reps.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert all((x["steps"], x["rays"], x["sampleCount"]) == (reps[0]["steps"], reps[0]["rays"], 48 * 32 * 8) for x in reps)
    assert all(abs(x["radianceSum"] - reps[0]["radianceSum"]) <= 1e-6 * reps[0]["radianceSum"] for x in reps)
    r = subprocess.run([cli, "--traversal", "fastest"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2
