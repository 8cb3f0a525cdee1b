"""The C-ABI library loads and exports every symbol include/pt_amd.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("pt_create", "pt_mesh_upload", "pt_scene_build", "pt_render_init", "pt_render_step", "pt_render_run",
                 "pt_image_read", "pt_trace_rays", "pt_tonemap_rgba8", "pt_destroy"):
        assert must in syms


def test_library_exports_every_declared_symbol(pta):
    lib = ctypes.CDLL(pta.capi.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    assert sorted(pta.capi.SIGNATURES) == declared_symbols()


def test_struct_sizes_match_header(pta):
    c = pta.capi
    assert ctypes.sizeof(c.PtInstance) == 160
    assert ctypes.sizeof(c.PtRenderParams) == 4 * 2 + 12 + 12 + 4 + 4 + 4 + 4 + 8 * 3 + 8   # with natural padding
    assert ctypes.sizeof(c.PtRunStats) == 112 and ctypes.sizeof(c.PtStepStats) == 32


def test_no_gpu_is_a_loud_error(pta):
    """Without a GPU the product must fail, not fall back (this container has none)."""
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    with pytest.raises(pta.capi.PtError):
        pta.request_device(0)
