"""MI355X-native hot path of AddisonPrairie/Path-Tracer: LBVH build, wavefront path tracing, tonemap.

Layout: csrc/ (HIP kernels + C ABI -> libpt_amd.so), js/ (headless Node.js host + N-API addon),
host.py (Python twin of the JS host, used by tests and bench.py), scenes.py (seeded synthetic scenes)."""
from . import capi, host, multigpu, scenes  # noqa: F401
from .host import (init_debug_view, init_display, init_path_tracer, init_scene, parse_obj, request_device,  # noqa: F401
                   compute_transform_matrices, get_transformed_bounding_box)
