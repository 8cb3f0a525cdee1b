"""pt_trace_rays_noalloc (include/pt_amd.h): the variant of pt_trace_rays for callers that cannot allocate.  Its spill and its
guard counter live in a file-scope device array of the library; inputs and outputs are the caller's device memory.
Checked through the C ABI against pt_trace_rays itself, bit for bit, on two calls in a row (the second smaller, so stale
contents of the shared array would show), on a scene whose rays push the traversal stack past its LDS part into that array."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def deep_chain_scene(pta):
    """30 boxes growing geometrically along the diagonal, 64 coincident slivers each: the Morton codes form a chain, the ties
    add six levels, and rays along the diagonal keep one pending entry per level (the oracle measures 29 > the 24 in LDS)."""
    vs = []
    for j in range(30):
        t = 1.35 ** j
        c, h = np.array([t, t, t]), 0.3 * t
        for _ in range(64):
            vs += [c + np.array([-h, -h, -h]), c + np.array([h, -h, -h * 0.99]), c + np.array([h, -h, h])]
    v = np.asarray(vs, np.float32)
    mesh = pta.scenes.mesh_from_vertices(v, np.arange(len(v)).reshape(-1, 3))
    return {"meshes": [mesh], "instances": [(0, [0, 0, 0], [0, 0, 0], [1, 1, 1], 0), (0, [5000, 0, 0], [0, 0, 0], [1, 1, 1], 0)], "camera": None}


def diagonal_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.2, 0.2, (n, 3)).astype(np.float32)
    d = (np.array([1, 1, 1]) + rng.normal(0, 0.05, (n, 3))).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def noalloc(dev, torch, o, d, counts=True, outputs=True):
    """One pt_trace_rays_noalloc call on torch-owned device buffers (allocated here, by the test: the entry point itself
    allocates nothing); returns the outputs as numpy arrays after pt_sync."""
    n = o.shape[0]
    to = torch.from_numpy(np.ascontiguousarray(o)).cuda(); td = torch.from_numpy(np.ascontiguousarray(d)).cuda()
    ho = torch.full((n,), -77, dtype=torch.int32, device="cuda"); ht = torch.full((n,), -77, dtype=torch.int32, device="cuda")
    hd = torch.full((n,), -7.0, dtype=torch.float32, device="cuda"); cn = torch.full((n, 4), -77, dtype=torch.int32, device="cuda")
    gh = torch.full((1,), 12345, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    p = lambda t: C.c_void_p(t.data_ptr())
    dev.call("pt_trace_rays_noalloc", p(to), p(td), n, p(ho) if outputs else None, p(ht) if outputs else None, p(hd),
             p(cn) if counts else None, p(gh))
    dev.call("pt_sync")
    return ho.cpu().numpy(), ht.cpu().numpy(), hd.cpu().numpy(), cn.cpu().numpy(), int(gh.cpu()[0])


def assert_same(a, b, n):
    for x, y in zip(a[:4], b[:4]):
        assert x.shape[0] == n
        np.testing.assert_array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)


def test_trace_rays_noalloc_equals_trace_rays(pta, orc):
    import torch
    dev = pta.request_device(0)      # its own context: the module-scoped scenes of the other test files stay untouched
    try:
        desc = deep_chain_scene(pta)
        s = pta.scenes.populate(pta.init_scene(dev), desc); s.build()
        o_scene = orc.OrcScene(desc)
        # first call: 3 full chunks of 65,536 rays and a ragged one; second call, straight after it: smaller (one chunk and a bit)
        for n, seed in ((200_000, 1), (70_001, 2)):
            o, d = diagonal_rays(n, seed)
            want = s.trace_rays(o, d, counts=True)
            assert s.last_guard_hits == 0
            got = noalloc(dev, torch, o, d)
            assert_same(got, want, n)
            assert got[4] == 0                                      # guard counter: cleared by the call, nothing hit it
            if seed == 1:                                           # the oracle agrees, and says the spill was in use
                ref = o_scene.trace_rays(o[:20000], d[:20000], counts=True)
                assert o_scene.last_max_depth > 24 and o_scene.last_overflow == 0
                assert_same(tuple(x[:20000] for x in got[:4]), ref, 20000)
            assert (want[0] >= 0).mean() > 0.2 and (want[0] < 0).mean() > 0.2
        # outputs are optional; untouched buffers keep the test's fill
        o, d = diagonal_rays(1000, 3)
        got = noalloc(dev, torch, o, d, counts=False, outputs=False)
        assert np.all(got[0] == -77) and np.all(got[1] == -77) and np.all(got[3] == -77)
        np.testing.assert_array_equal(got[2].view(np.uint32), s.trace_rays(o, d)[2].view(np.uint32))
        # n = 0 is a no-op; beyond the stated maximum is refused with a clear error, before anything is launched
        dev.call("pt_trace_rays_noalloc", C.c_void_p(8), C.c_void_p(8), 0, None, None, None, None, None)
        with pytest.raises(pta.capi.PtError, match="at most 4194304"):
            dev.call("pt_trace_rays_noalloc", C.c_void_p(8), C.c_void_p(8), pta.capi.PT_TRACE_NOALLOC_MAX_RAYS + 1, None, None, None, None, None)
        # a second context on the same GPU uses the same array (calls one after the other): a two-level scene with another camera
        desc2 = pta.scenes.bunny_scale_scene(level=3)
        dev2 = pta.request_device(0)
        try:
            s2 = pta.scenes.populate(pta.init_scene(dev2), desc2); s2.build()
            rng = np.random.default_rng(4)
            o2 = np.tile(np.float32(desc2["camera"]["position"]), (30_000, 1))
            d2 = (np.float32(desc2["camera"]["lookAt"]) + rng.normal(0, 1.0, (30_000, 3)).astype(np.float32)) - o2
            d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
            assert_same(noalloc(dev2, torch, o2, d2.astype(np.float32)), s2.trace_rays(o2, d2, counts=True), 30_000)
        finally:
            dev2.destroy()
    finally:
        dev.destroy()
