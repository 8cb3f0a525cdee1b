"""MI355X-native hot path of AddisonPrairie/Path-Tracer: LBVH build, wavefront path tracing, tonemap.

Layout: csrc/ (HIP kernels + C ABI -> libpt_amd.so), js/ (headless Node.js host + N-API addon),
host.py (Python twin of the JS host, used by tests and bench.py), scenes.py (seeded synthetic scenes)."""
import os as _os

# Four path pools at 4K need more hardware queues than HIP's default of four (streams beyond the count share a queue and their kernels
# serialise; host.default_pool_shape).  The variable is read when the HIP runtime starts, so it only takes effect if this package -- or
# the caller -- sets it before anything touches the GPU; a value the caller has set is kept.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import capi, host, multigpu, scenes  # noqa: F401,E402
from .host import (init_debug_view, init_display, init_path_tracer, init_scene, parse_obj, request_device,  # noqa: F401
                   compute_transform_matrices, get_transformed_bounding_box)
