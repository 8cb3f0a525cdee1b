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
