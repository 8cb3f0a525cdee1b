"""ctypes binding of libpt_amd.so (the C ABI declared in include/pt_amd.h).

This module is plumbing only: it loads the in-tree shared library and declares argument types.
There is no fallback: if the HIP library is missing or no GPU is visible the calls raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_AMD_LIB") or os.path.join(_HERE, "libpt_amd.so")   # override: A/B of two builds

PT_RUN_TIME_EXTEND = 1
PT_RUN_COUNT_BYTES = 2
PT_RUN_TIME_EXTEND_ALL = 4


class PtInstance(C.Structure):
    _fields_ = [("l2w", C.c_float * 16), ("w2l", C.c_float * 16), ("aabb_min", C.c_float * 3),
                ("aabb_max", C.c_float * 3), ("mesh", C.c_int32), ("material", C.c_int32)]


class PtSceneInfo(C.Structure):
    _fields_ = [("n_tris", C.c_int64), ("n_objects", C.c_int32), ("n_meshes_used", C.c_int32)]


class PtRenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("cam_pos", C.c_float * 3),
                ("cam_look_at", C.c_float * 3), ("fov_deg", C.c_float), ("spp", C.c_int32),
                ("num_paths", C.c_int32), ("slot_base", C.c_int64), ("ticket_begin", C.c_int64),
                ("ticket_end", C.c_int64), ("image_dev", C.c_void_p)]


class PtStepStats(C.Structure):
    _fields_ = [("n_camera", C.c_int64), ("n_material", C.c_int64), ("n_rays", C.c_int64), ("step_index", C.c_int64)]


class PtRunStats(C.Structure):
    _fields_ = [("steps", C.c_int64), ("total_rays", C.c_int64), ("total_samples", C.c_int64),
                ("stack_overflows", C.c_int64), ("extend_ms", C.c_double), ("extend_launches", C.c_int64),
                ("traversal_bytes", C.c_int64), ("wasted_rays", C.c_int64), ("checked_rays", C.c_int64),
                ("mismatched_rays", C.c_int64), ("flipped_rays", C.c_int64), ("max_rel_dist_diff", C.c_double),
                ("beyond_tolerance_rays", C.c_int64), ("lost_hit_rays", C.c_int64)]


PT_TRAVERSAL_REFERENCE, PT_TRAVERSAL_WIDE, PT_TRAVERSAL_AUTO = 0, 1, 2
PT_TRACE_NOALLOC_MAX_RAYS = 4194304
PT_MAT_LAMBERT, PT_MAT_EMITTER, PT_MAT_MIRROR, PT_MAT_GLOSSY, PT_MAX_EXT_MATERIALS = 0, 1, 2, 3, 14


class PtMaterial(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("type", C.c_int32), ("emission", C.c_float * 3), ("pad", C.c_int32)]


class PtRenderExt(C.Structure):
    _fields_ = [("max_nb", C.c_int32), ("seed", C.c_uint32), ("n_materials", C.c_int32), ("pad", C.c_int32),
                ("materials", PtMaterial * PT_MAX_EXT_MATERIALS)]


# name -> (restype, argtypes); every symbol include/pt_amd.h declares
SIGNATURES = {
    "pt_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "pt_destroy": (C.c_int, [C.c_void_p]),
    "pt_last_error": (C.c_char_p, [C.c_void_p]),
    "pt_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pt_sync": (C.c_int, [C.c_void_p]),
    "pt_device_pci_bus_id": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "pt_mesh_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pt_scene_build": (C.c_int, [C.c_void_p, C.POINTER(PtInstance), C.c_int]),
    "pt_scene_get_info": (C.c_int, [C.c_void_p, C.POINTER(PtSceneInfo)]),
    "pt_get_blas_nodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pt_get_blas_tris": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pt_get_tlas_nodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pt_get_objects": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pt_get_tlas_perm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pt_get_mesh_perm": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "pt_scene_get_stack_bound": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "pt_trace_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_trace_last_guard_hits": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "pt_trace_rays_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_image_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pt_render_init": (C.c_int, [C.c_void_p, C.POINTER(PtRenderParams)]),
    "pt_render_reset": (C.c_int, [C.c_void_p]),
    "pt_render_set_extensions": (C.c_int, [C.c_void_p, C.POINTER(PtRenderExt)]),
    "pt_render_step": (C.c_int, [C.c_void_p, C.POINTER(PtStepStats)]),
    "pt_render_run": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.POINTER(PtRunStats)]),
    "pt_create_sharing_scene": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "pt_render_run_many": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int, C.POINTER(PtRunStats)]),
    "pt_image_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pt_image_dev_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "pt_image_flush": (C.c_int, [C.c_void_p]),
    "pt_comm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]),
    "pt_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pt_comm_create_rank": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "pt_comm_reduce_image": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_comm_destroy": (None, [C.c_void_p]),
    "pt_path_state_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "pt_queue_read": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int64)]),
    "pt_extend_variants": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]),
    "pt_set_extend_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_set_traversal_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_get_traversal_mode": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "pt_set_traversal_check": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_extend_replay": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "pt_extend_pass_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "pt_extend_wave_log": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "pt_tonemap_rgba8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
}

_lib = None


def load():
    """Load libpt_amd.so (built by `make -C path-tracer_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C path-tracer_amd/csrc` "
                               "(there is no CPU fallback for the hot path)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class PtError(RuntimeError):
    pass


def check(lib, ctx, rc, what):
    if rc < 0:
        msg = lib.pt_last_error(ctx)
        raise PtError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
    return rc
