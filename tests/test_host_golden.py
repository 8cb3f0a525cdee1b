"""Host-side logic (OBJ parser, TRS matrices, transformed boxes) against golden vectors produced by
the reference's own JS (tests/golden/make_host_fixtures.js ran parse-obj.js / transform-utils.js
under Node 12 in the build container)."""
import json
import os

import numpy as np
import pytest

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "host_fixtures.json")))


@pytest.mark.parametrize("case", FIX["parseObj"], ids=[c["name"] for c in FIX["parseObj"]])
def test_parse_obj_matches_reference(pta, case):
    got = pta.parse_obj(case["text"])
    assert got["numTriangles"] == case["numTriangles"]
    want = np.asarray([np.nan if v is None else v for v in case["triangleArray"]], dtype=np.float64)
    np.testing.assert_array_equal(got["triangleArray"], want)          # exact in float64
    assert got["bounds"]["min"] == case["bounds"]["min"] and got["bounds"]["max"] == case["bounds"]["max"]


@pytest.mark.parametrize("case", FIX["transforms"], ids=[str(i) for i in range(len(FIX["transforms"]))])
def test_transform_matrices_match_reference(pta, case):
    t = case["transform"]
    m = pta.compute_transform_matrices(t["position"], t["rotation"], t["scale"])
    for key in ("localToWorld", "worldToLocal"):
        got, want = np.asarray(m[key]), np.asarray(case[key])
        # V8's and glibc's sin/cos may differ in the last f64 bit; what reaches the GPU is the f32 rounding
        np.testing.assert_allclose(got, want, rtol=4e-16, atol=1e-300)
        np.testing.assert_array_equal(got.astype(np.float32), want.astype(np.float32))
    for box, want in zip(case["boxes"], case["transformedBoxes"]):
        got = pta.get_transformed_bounding_box({"localToWorld": case["localToWorld"]}, box)
        assert got["min"] == want["min"] and got["max"] == want["max"]    # same inputs -> exact


def test_floor_instance_world_box(pta):
    """floor.obj bounds and the main.js:33 instance box, as executed in the survey session."""
    p = pta.parse_obj(pta.scenes.FLOOR_OBJ)
    assert p["numTriangles"] == 2 and p["bounds"] == {"min": [-10.0, -10.0, 0.0], "max": [10.0, 10.0, 0.0]}
    m = pta.compute_transform_matrices([10, 10, -1.3], [0, 0, 0], [5, 5, 5])
    b = pta.get_transformed_bounding_box(m, p["bounds"])
    assert b == {"min": [-40.0, -40.0, -1.3], "max": [60.0, 60.0, -1.3]}


def test_render_ext_packing_against_hand_written_bytes(pta):
    """host.render_ext turns settings.{maxNb, seed, materials} into the pt_render_ext record the library AND the oracle are fed with
    (tests/orc.py uses it for both), so a packing bug there would be invisible to every extension parity test.  Pinned here against
    bytes written out field by field from the layout in include/pt_amd.h / oracle/pt_oracle.h: {i32 max_nb, u32 seed, i32 n_materials,
    i32 pad, 14 x {f32 albedo[3], i32 type, f32 emission[3], i32 pad}}; types lambert 0, emitter 1, mirror 2, glossy 3; a glossy
    material's roughness travels in emission[0]."""
    import ctypes
    import struct
    import orc
    settings = {"maxNb": 16, "seed": 0x1_0000_0007,          # the seed is taken modulo 2^32
                "materials": [{"type": "lambert", "albedo": [0.8, 0.35, 0.05]},
                              {"type": "emitter", "emission": [17.0, 12.0, 4.0]},
                              {"type": "mirror", "albedo": [0.9, 0.9, 0.9]},
                              {"type": "glossy", "albedo": [0.95, 0.75, 0.35], "roughness": 0.15, "emission": [9.0, 9.0, 9.0]}]}
    want = struct.pack("<iIii", 16, 7, 4, 0)
    want += struct.pack("<3fi3fi", 0.8, 0.35, 0.05, 0, 0.0, 0.0, 0.0, 0)
    want += struct.pack("<3fi3fi", 0.0, 0.0, 0.0, 1, 17.0, 12.0, 4.0, 0)
    want += struct.pack("<3fi3fi", 0.9, 0.9, 0.9, 2, 0.0, 0.0, 0.0, 0)
    want += struct.pack("<3fi3fi", 0.95, 0.75, 0.35, 3, 0.15, 9.0, 9.0, 0)      # roughness replaces emission[0]
    want += bytes(32 * 10)
    assert len(want) == 16 + 14 * 32
    for cls in (pta.capi.PtRenderExt, orc.OrcRenderExt):
        ext = pta.host.render_ext(settings, cls)
        assert ctypes.sizeof(ext) == len(want) and bytes(ext) == want, cls.__name__
    assert pta.host.render_ext({"samples": 8}) is None              # none of the extension keys: the reference's rules, no call
    only_nb = pta.host.render_ext({"maxNb": 8})
    assert bytes(only_nb) == struct.pack("<iIii", 8, 0, 0, 0) + bytes(32 * 14)


def test_default_pool_shape(pta):
    """What the hosts use when the caller fixes neither numPaths nor pools (the measured optima, DESIGN.md "Pools"): three pools of a quarter
    of the frame's pixels each, clamped to 750,000 .. 2,000,000 slots; at 4K four pools of 1,500,000 when HIP has at least 8 hardware queues."""
    import os
    old = os.environ.get("GPU_MAX_HW_QUEUES")
    try:
        for queues, at_4k in (("4", (3, 2_000_000)), ("8", (4, 1_500_000)), ("16", (4, 1_500_000)), ("junk", (3, 2_000_000))):
            os.environ["GPU_MAX_HW_QUEUES"] = queues        # four pools only where every pool's stream gets a hardware queue of its own
            assert pta.host.default_pool_shape(1920, 1080) == (3, 750_000)
            assert pta.host.default_pool_shape(3840, 2160) == at_4k
            assert pta.host.default_pool_shape(512, 512) == (3, 750_000)
            assert pta.host.default_pool_shape(2560, 1440) == (3, 921_600)
    finally:
        if old is None:
            os.environ.pop("GPU_MAX_HW_QUEUES", None)
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = old
