"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bars (north_star): BVH node layout and primitive order bit-exact (f_1/f_2 excluded: the reference's own
values are a race-dependent arrival counter / zero); per-ray closest-hit indices bit-exact;
path state bit-exact step by step (same strict-IEEE arithmetic contract on both sides); image sample
counts exact and radiance within 1e-5 relative (float atomic-add order is the only freedom, logic.js:222-233).
The oracle itself is "parity unpinned" against the reference (no fixtures exist there); its own pins
are in test_oracle_pins.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NODE_COLS = [c for c in range(16) if c not in (7, 15)]   # everything but f_1 / f_2


def scene_descs(pta):
    sc = pta.scenes
    return {
        "floor": sc.floor_scene(),
        "cornell": sc.cornell_box(),
        "cornell_single": sc.cornell_box(single_mesh=True),
        "soup5k": {"meshes": [sc.random_soup(5000, 1)], "instances": [(0, [0, 0, 0], [0.1, 0.2, 0.3], [1, 1, 1], 0)],
                   "camera": {"position": [0, -14, 3], "lookAt": [0, 0, 0], "fov": 25}},
        "soup_ties": {"meshes": [sc.random_soup(20000, 2, extent=0.02, size=0.01), {"file": sc.FLOOR_OBJ}],
                      "instances": [(0, [0, 0, 1], [0, 0, 0], [30, 30, 30], 1), (1, [0, 0, 0], [0, 0, 0], [1, 1, 1], 0),
                                    (0, [3, 1, 2], [1, 2, 3], [10, 20, 30], 0)],
                      "camera": {"position": [4, -6, 3], "lookAt": [0, 0, 1], "fov": 30}},
        "sphere": sc.bunny_scale_scene(level=4),
        "grid": sc.grid_scene(96, 80),
        "instanced": sc.instanced_scene(n_side=24, grid=4),
    }


@pytest.fixture(scope="module")
def built(pta, orc, device):
    cache = {}

    def get(name):
        if name not in cache:
            desc = scene_descs(pta)[name]
            s = pta.scenes.populate(pta.init_scene(device), desc)
            s.build()
            cache.clear()            # one GPU scene at a time per ctx (build frees the previous one)
            cache[name] = (desc, s, orc.OrcScene(desc))
        return cache[name]
    return get


NAMES = ["floor", "cornell", "cornell_single", "soup5k", "soup_ties", "sphere", "grid", "instanced"]


@pytest.mark.parametrize("name", NAMES)
def test_bvh_layout_bit_exact(built, name):
    desc, s, o = built(name)
    inf = s.info()
    assert (inf.n_tris, inf.n_objects) == (o.n_tris, o.n_objects)
    np.testing.assert_array_equal(s.tlas_perm(), o.tlas_perm())
    for mi in range(len(desc["meshes"])):
        if any(i[0] == mi for i in desc["instances"]):
            np.testing.assert_array_equal(s.mesh_perm(mi), o.mesh_perm(mi))
    np.testing.assert_array_equal(s.blas_tris().view(np.uint32), o.blas_tris().view(np.uint32))
    np.testing.assert_array_equal(s.blas_nodes()[:, NODE_COLS], o.blas_nodes()[:, NODE_COLS])
    np.testing.assert_array_equal(s.tlas_nodes()[:, NODE_COLS], o.tlas_nodes()[:, NODE_COLS])
    np.testing.assert_array_equal(s.objects_buffer(), o.objects_buffer())


@pytest.mark.parametrize("n_tris", [1, 2, 3, 63, 64, 65, 1023, 1025, 4095, 4096, 4097, 8193, 70001])
def test_sort_and_tree_at_tile_boundaries(pta, orc, n_tris):
    """a11-a13 on mesh sizes around the radix sort's wave group (64), wave share (1024) and tile (4096), on a soup squeezed into a
    few Morton cells so that most keys tie (a stable sort must keep tied keys in index order, radix-sort.js; the tree then
    splits them by index, radix-tree.js:172-185): permutation and nodes equal the oracle's bit for bit."""
    dev = pta.request_device(0)
    try:
        mesh = pta.scenes.random_soup(n_tris, 100 + n_tris, extent=0.05, size=0.04)
        big = pta.scenes.random_soup(max(n_tris // 3, 1), 7, extent=3.0, size=0.3)          # spread keys: every digit of every pass in use
        desc = {"meshes": [mesh, big], "instances": [(0, [0, 0, 0], [0, 0, 0], [1, 1, 1], 0), (1, [1, 0, 0], [0.1, 0, 0], [1, 1, 1], 1)], "camera": None}
        s = pta.scenes.populate(pta.init_scene(dev), desc); s.build()
        o = orc.OrcScene(desc)
        for mi in (0, 1):
            np.testing.assert_array_equal(s.mesh_perm(mi), o.mesh_perm(mi))
        np.testing.assert_array_equal(s.tlas_perm(), o.tlas_perm())
        np.testing.assert_array_equal(s.blas_nodes()[:, NODE_COLS], o.blas_nodes()[:, NODE_COLS])
        np.testing.assert_array_equal(s.blas_tris().view(np.uint32), o.blas_tris().view(np.uint32))
    finally:
        dev.destroy()


def rays_for(desc, n, seed):
    rng = np.random.default_rng(seed)
    cam = np.asarray(desc["camera"]["position"], np.float32)
    look = np.asarray(desc["camera"]["lookAt"], np.float32)
    k = n // 2
    o1 = np.tile(cam, (k, 1)); t1 = look + rng.normal(0, 1.5, (k, 3)).astype(np.float32)
    o2 = (look + rng.normal(0, 2.0, (n - k, 3))).astype(np.float32); t2 = rng.normal(0, 1, (n - k, 3)).astype(np.float32) + o2
    o = np.concatenate([o1, o2]); d = np.concatenate([t1, t2]) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o.astype(np.float32), d.astype(np.float32)


@pytest.mark.parametrize("name", NAMES)
def test_closest_hit_bit_exact(built, name):
    desc, s, o = built(name)
    ro, rd = rays_for(desc, 200_000, 7)
    g_obj, g_tri, g_dist, g_cnt = s.trace_rays(ro, rd, counts=True)
    c_obj, c_tri, c_dist, c_cnt = o.trace_rays(ro, rd, counts=True)
    assert o.last_overflow == 0 and s.last_guard_hits == 0
    np.testing.assert_array_equal(g_obj, c_obj)
    np.testing.assert_array_equal(g_tri, c_tri)
    np.testing.assert_array_equal(g_dist.view(np.uint32), c_dist.view(np.uint32))
    np.testing.assert_array_equal(g_cnt, c_cnt)       # identical traversal order => identical algorithmic bytes
    assert (g_obj >= 0).mean() > 0.02


def test_axis_aligned_and_degenerate_rays(built):
    """d components of exactly 0 (1/0 = inf, 0*inf = NaN in the slab test), rays inside boxes,
    rays starting on geometry."""
    desc, s, o = built("cornell")
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 1, 0], [0, 1, 1]], np.float32)
    axes /= np.linalg.norm(axes, axis=1, keepdims=True)
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.uniform(-1, 1, (500, 3)) + [0, 0, 1], np.array([[0, 0, 0], [1, 1, 2], [-1, 0, 1], [0, -4.6, 1]])]).astype(np.float32)
    ro = np.repeat(pts, len(axes), axis=0); rd = np.tile(axes, (len(pts), 1))
    g = s.trace_rays(ro, rd, counts=True); c = o.trace_rays(ro, rd, counts=True)
    for a, b in zip(g, c):
        np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)


def test_non_finite_and_empty_ray_batches(built):
    """Rays nobody should send but a host can: zero direction (1/0 = inf in every slab term), NaN and infinite components in origin or
    direction, unnormalised and denormal directions; and an empty batch.  Hit indices and visit counts equal the oracle's, hit distances
    bit for bit except that a NaN may differ in payload (min/max pass NaNs through on both sides; the WGSL leaves them undefined)."""
    desc, s, o = built("cornell")
    inf, nan = np.inf, np.nan
    ro = np.array([[0, -4, 1], [0, -4, 1], [0, -4, 1], [0, -4, 1], [nan, 0, 1], [0, inf, 1], [0, -4, 1], [0, -4, 1], [0, 0, 1], [0, -4, 1]], np.float32)
    rd = np.array([[0, 0, 0], [nan, 1, 0], [0, inf, 0], [0, 1e-40, 0], [0, 1, 0], [0, 1, 0], [0, 250, 0], [0, 1e-20, 1e-20], [-0.0, 1, -0.0], [0, 1, 0]], np.float32)
    g = s.trace_rays(ro, rd, counts=True); c = o.trace_rays(ro, rd, counts=True)
    np.testing.assert_array_equal(g[0], c[0]); np.testing.assert_array_equal(g[1], c[1]); np.testing.assert_array_equal(g[3], c[3])
    both_nan = np.isnan(g[2]) & np.isnan(c[2])
    np.testing.assert_array_equal(g[2].view(np.uint32)[~both_nan], c[2].view(np.uint32)[~both_nan])
    assert g[0][-1] >= 0                                                       # the last ray is an ordinary one and hits the box
    e = s.trace_rays(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), counts=True)
    assert all(len(x) == 0 for x in e)


FIELDS = ["pixel_index", "num_bounces", "random_seed", "path_throughput", "material_throughput_pdf", "path_o", "path_d",
          "hit_obj", "hit_tri", "flags", "samples_in_pixel"]


def assert_state_equal(pt, orr, step):
    for f in FIELDS:
        a, b = pt.field(f), orr.field(f)
        if a.ndim == 2:
            a, b = a[:, :3], b[:, :3]
        if a.dtype == np.float32:
            a, b = a.view(np.uint32), b.view(np.uint32)
        assert np.array_equal(a, b), f"step {step}: field {f} differs in {(a != b).sum()} elements"
    assert pt.field("pixel_counter")[0] == orr.field("pixel_counter")[0]


@pytest.mark.parametrize("name,w,h,spp,paths", [("floor", 64, 64, 16, 3000), ("cornell", 96, 54, 8, 5000),
                                                  ("instanced", 80, 60, 24, 4096), ("cornell_single", 40, 30, 16, 777)])
def test_path_state_bit_exact_step_by_step(pta, orc, device, built, name, w, h, spp, paths):
    desc, s, o = built(name)
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": {"samples": spp, "numPaths": paths}})
    orr = orc.OrcRender(o, w, h, desc["camera"], spp, paths)
    for step in range(30):
        gs = pt.step(); cs = orr.step_like_device()
        assert (gs.n_camera, gs.n_material, gs.n_rays) == (cs.n_camera, cs.n_material, cs.n_rays), step
        np.testing.assert_array_equal(pt.queue(0), orr.queue(0))
        np.testing.assert_array_equal(pt.queue(1), orr.queue(1))
        assert_state_equal(pt, orr, step)
    img_g, img_c = pt.read_image(), orr.image()
    np.testing.assert_array_equal(img_g[..., 3], img_c[..., 3])
    np.testing.assert_allclose(img_g[..., :3], img_c[..., :3], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name,w,h,spp,paths", [("floor", 128, 128, 16, 20000), ("cornell", 160, 90, 32, 50000), ("sphere", 120, 80, 16, 30000)])
def test_full_render_matches_oracle(pta, orc, device, built, name, w, h, spp, paths):
    desc, s, o = built(name)
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": {"samples": spp, "numPaths": paths}})
    rs = pt.run(count_bytes=True)
    orr = orc.OrcRender(o, w, h, desc["camera"], spp, paths)
    steps, rays, byts = orr.run()
    assert (rs.steps, rs.total_rays, rs.traversal_bytes, rs.stack_overflows) == (steps, rays, byts, 0)
    assert rs.total_samples == w * h * spp
    # wasted rays: every slot is retired once by ticket exhaustion and traces one last ray in that step (logic.js:177-190)
    assert rs.wasted_rays == int((pt.field("flags") >> 31).sum()) == int((orr.field("flags") >> 31).sum()) == paths
    img_g, img_c = pt.read_image(), orr.image()
    np.testing.assert_array_equal(img_g[..., 3], np.full((h, w), spp, np.float32))
    np.testing.assert_array_equal(img_g[..., 3], img_c[..., 3])
    np.testing.assert_allclose(img_g[..., :3], img_c[..., :3], rtol=1e-5, atol=1e-6)
    # display.js:98-113: tonemap agrees to one 8-bit step (pow is backend-defined)
    disp = pta.init_display({"device": device, "image": {"buffer": pt.get_image_buffer(), "width": w, "height": h}})
    t_g, t_c = disp.draw().astype(np.int32), orc.tonemap(img_c).astype(np.int32)
    assert np.abs(t_g - t_c).max() <= 1


@pytest.mark.parametrize("w,h,spp,paths", [(37, 23, 12, 1000), (16, 16, 1, 300), (1, 1, 8, 64), (3, 5, 20, 5000), (50, 40, 5, 4097)])
def test_ragged_sample_counts_and_tiny_images(pta, orc, device, built, w, h, spp, paths):
    """Edge cases of logic.js:177-183: the ticket count is ceil(W*H*spp / 8), so an spp that is not a multiple of 8 leaves some pixels
    with 8 samples more than others; a pool larger than the whole job retires most slots at their first ticket; a one-pixel image.
    Whole render against the oracle: steps, rays, sample counts per pixel, radiance."""
    desc, s, o = built("cornell")
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": {"samples": spp, "numPaths": paths}})
    rs = pt.run()
    orr = orc.OrcRender(o, w, h, desc["camera"], spp, paths)
    steps, rays, _ = orr.run()
    tickets = -(-w * h * spp // 8)
    assert (rs.steps, rs.total_rays, rs.stack_overflows) == (steps, rays, 0)
    assert rs.total_samples == 8 * tickets and rs.wasted_rays == paths
    img_g, img_c = pt.read_image(), orr.image()
    np.testing.assert_array_equal(img_g[..., 3], img_c[..., 3])
    assert img_g[..., 3].sum() == 8 * tickets and set(np.unique(img_g[..., 3]).tolist()) <= {8.0 * (tickets // (w * h)), 8.0 * (tickets // (w * h) + 1)}
    np.testing.assert_allclose(img_g[..., :3], img_c[..., :3], rtol=1e-5, atol=1e-6)
    assert_state_equal(pt, orr, steps)


def test_spp8_image_bit_exact(pta, orc, device, built):
    """With spp == 8 every pixel is owned by one slot for the whole render, so even the float add order is fixed."""
    desc, s, o = built("cornell")
    w, h = 64, 36
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": {"samples": 8, "numPaths": 1000}})
    pt.run()
    orr = orc.OrcRender(o, w, h, desc["camera"], 8, 1000); orr.run()
    np.testing.assert_array_equal(pt.read_image().view(np.uint32), orr.image().view(np.uint32))


def test_floor_analytic_radiance(pta, device, built):
    """C1 analytic check: every floor sample is exactly one Lambert bounce into the sky:
    radiance 2 * 0.5^2.2; every sky sample is 2."""
    desc, s, o = built("floor")
    w = h = 128
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": {"samples": 16, "numPaths": 50000}})
    pt.run()
    img = pt.read_image(); rad = img[..., 0] / img[..., 3]
    floor_v, sky_v = 2 * 0.5 ** 2.2, 2.0
    is_floor = np.abs(rad - floor_v) < 1e-5; is_sky = np.abs(rad - sky_v) < 1e-6
    mixed = ~(is_floor | is_sky)
    assert is_floor.mean() > 0.3 and mixed.mean() < 0.05          # only horizon pixels mix the two
    assert np.all((rad[mixed] > floor_v - 1e-5) & (rad[mixed] < sky_v + 1e-5))


def test_multi_gpu_partition_on_one_gpu(pta, orc, device, built):
    """The spp split: two ticket ranges with their own slot_base, summed, give every pixel exactly
    spp samples; each share is bit-exact against the oracle run with the same parameters."""
    desc, s, o = built("cornell")
    w, h, spp, paths = 80, 45, 32, 4000
    total = w * h * spp // 8
    acc = np.zeros((h, w, 4), np.float64)
    for rank in range(2):
        lo, hi = rank * total // 2, (rank + 1) * total // 2
        st = {"samples": spp, "numPaths": paths, "slotBase": rank * paths, "ticketBegin": lo, "ticketEnd": hi}
        pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"], "settings": st})
        pt.run()
        orr = orc.OrcRender(o, w, h, desc["camera"], spp, paths, slot_base=rank * paths, ticket_begin=lo, ticket_end=hi); orr.run()
        img = pt.read_image()
        np.testing.assert_array_equal(img[..., 3], orr.image()[..., 3])
        np.testing.assert_allclose(img[..., :3], orr.image()[..., :3], rtol=1e-5, atol=1e-6)
        acc += img
    np.testing.assert_array_equal(acc[..., 3], np.full((h, w), spp))


@pytest.mark.parametrize("paths", [1, 63, 64, 65, 255, 257, 1023, 1024, 1025, 2049])
def test_pool_sizes_around_the_workgroup_boundaries(pta, orc, device, built, paths):
    """Slot pools that are not a multiple of the wave (64), of the compaction workgroup (256) or of the count record
    (1024): queues, tickets and the finished frame still match the oracle (empty, ragged and exactly-full tails)."""
    desc, s, o = built("cornell")
    w, h, spp = 24, 16, 16
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": {"samples": spp, "numPaths": paths}})
    orr = orc.OrcRender(o, w, h, desc["camera"], spp, paths)
    for step in range(3):
        gs = pt.step(); cs = orr.step_like_device()
        assert (gs.n_camera, gs.n_material) == (cs.n_camera, cs.n_material)
        np.testing.assert_array_equal(pt.queue(0), orr.queue(0)); np.testing.assert_array_equal(pt.queue(1), orr.queue(1))
        assert_state_equal(pt, orr, step)
    rs = pt.run(max_steps=200000)
    while orr.step_like_device().n_rays:
        pass
    img_g, img_c = pt.read_image(), orr.image()
    np.testing.assert_array_equal(img_g[..., 3], np.full((h, w), spp, np.float32))
    np.testing.assert_allclose(img_g[..., :3], img_c[..., :3], rtol=1e-5, atol=1e-6)
    assert rs.total_samples == w * h * spp and rs.stack_overflows == 0


def test_run_again_and_reset(pta, orc, device, built):
    """run() on a finished render traces nothing more; reset() + run() reproduces the frame (sample counts exactly, radiance
    to the atomic-order tolerance); a render stopped by max_steps can be continued to the same result."""
    desc, s, o = built("cornell")
    w, h, spp, paths = 64, 36, 16, 3000
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": {"samples": spp, "numPaths": paths}})
    r1 = pt.run(); img1 = pt.read_image().copy()
    r2 = pt.run()
    assert (r2.total_rays, r2.total_samples) == (r1.total_rays, r1.total_samples)
    np.testing.assert_array_equal(pt.read_image(), img1)
    pt.reset()
    r3 = pt.run(max_steps=5)
    assert r3.steps == 5 and r3.total_rays < r1.total_rays
    partial = pt.read_image()
    assert 0 < partial[..., 3].sum() < w * h * spp
    r4 = pt.run()
    assert (r4.steps, r4.total_rays, r4.total_samples) == (r1.steps, r1.total_rays, r1.total_samples)
    img2 = pt.read_image()
    np.testing.assert_array_equal(img2[..., 3], img1[..., 3])
    np.testing.assert_allclose(img2[..., :3], img1[..., :3], rtol=1e-5, atol=1e-6)


def test_wave_log_hook(pta, device, built):
    """pt_extend_wave_log: one record per launched wave, end >= start, classes and trip counts plausible, and the replay
    leaves the render's results alone."""
    import ctypes as C
    desc, s, o = built("cornell")
    paths = 5000
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": 96, "height": 54}, "camera": desc["camera"],
                               "settings": {"samples": 16, "numPaths": paths}})
    for _ in range(4):
        st = pt.step()
    cap = (paths + 63) // 64 + 4
    buf = np.zeros((cap, 4), np.uint64); n = C.c_int64()
    device.call("pt_extend_wave_log", 2, buf.ctypes.data_as(C.c_void_p), cap, C.byref(n))
    log = buf[: n.value]; live = log[log[:, 1] != 0]
    assert n.value == (paths + 63) // 64 and len(live) == (st.n_rays + 63) // 64
    assert np.all(live[:, 1] >= live[:, 0])
    cls = (live[:, 3] >> np.uint64(24)) & np.uint64(0xff); trips = live[:, 3] & np.uint64(0xffffff)
    assert set(np.unique(cls).tolist()) <= {0, 1} and trips.max() >= 3
    img_before = pt.read_image().copy()
    pt.run()
    assert pt.read_image()[..., 3].sum() == 96 * 54 * 16 and img_before[..., 3].sum() <= 96 * 54 * 16


def test_extensions_bit_exact_against_the_oracle(pta, orc, device):
    """SURVEY 8f.1 extensions (rgb Lambert, emitter, mirror, max_nb, seed): the oracle defines the rules, the kernels
    reproduce them -- whole path state after every step, then the finished frame."""
    desc = pta.scenes.cornell_box_ext()
    s = pta.scenes.populate(pta.init_scene(device), desc); s.build()
    o = orc.OrcScene(desc)
    ext = {"maxNb": 8, "seed": 7, "materials": desc["materials"]}
    w, h, spp, paths = 96, 54, 16, 6000
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": dict(samples=spp, numPaths=paths, **ext)})
    orr = orc.OrcRender(o, w, h, desc["camera"], spp, paths, ext=ext)
    for step in range(25):
        gs = pt.step(); cs = orr.step_like_device()
        assert (gs.n_camera, gs.n_material) == (cs.n_camera, cs.n_material), step
        assert_state_equal(pt, orr, step)
    assert pt.field("num_bounces").max() <= 10
    pt.run()
    while orr.step_like_device().n_rays:
        pass
    img_g, img_c = pt.read_image(), orr.image()
    np.testing.assert_array_equal(img_g[..., 3], img_c[..., 3])
    np.testing.assert_allclose(img_g[..., :3], img_c[..., :3], rtol=1e-5, atol=1e-5)
    assert img_g[..., 0].sum() > img_g[..., 2].sum() > 0       # warm light, red wall


def test_extensions_closed_forms_and_reset(pta, device):
    """emitter: first hit ends the path with exactly the emission; mirror under the sky: exactly 2 * albedo; extensions
    survive reset(); a new init_path_tracer without them is the reference again."""
    desc = {"meshes": [{"file": pta.scenes.FLOOR_OBJ}], "instances": [(0, [0, 0, 0], [0, 0, 0], [5, 5, 5], 2)],
            "camera": {"position": [0, 0, 8], "lookAt": [0, 0.001, 0], "fov": 20}}
    s = pta.scenes.populate(pta.init_scene(device), desc); s.build()
    base = {"device": device, "scene": s, "image": {"width": 16, "height": 16}, "camera": desc["camera"]}
    for mats, expect in (([{"type": "emitter", "emission": [3.0, 1.5, 0.25]}], [3.0, 1.5, 0.25]),
                         ([{"type": "mirror", "albedo": [0.5, 0.25, 1.0]}], [1.0, 0.5, 2.0]),
                         ([{"type": "glossy", "albedo": [0.25, 0.5, 0.125], "roughness": 0.6}], [0.5, 1.0, 0.25])):
        pt = pta.init_path_tracer(dict(base, settings={"samples": 8, "numPaths": 300, "materials": mats}))
        for again in range(2):
            pt.run()
            img = pt.read_image()
            np.testing.assert_array_equal(img[..., :3] / img[..., 3:4], np.broadcast_to(np.float32(expect), (16, 16, 3)))
            pt.reset()
    pt = pta.init_path_tracer(dict(base, settings={"samples": 8, "numPaths": 300}))      # id 2 unknown again: black
    pt.run()
    assert np.all(pt.read_image()[..., :3] == 0)
    with pytest.raises(pta.capi.PtError, match="started"):
        import ctypes as C
        pt.reset(); pt.step()
        device.call("pt_render_set_extensions", C.byref(pta.capi.PtRenderExt()))


def test_debug_view(pta, orc, device, built):
    """the rt-debug visualiser over pt_trace_rays.  Seen from above, the floor is uniformly its normal
    (0,0,1) -> blue; on the Cornell box the per-pixel hits are the oracle's and the index view encodes them."""
    desc, s, o = built("floor")
    top = {"position": [10, 10, 8], "lookAt": [10, 10.001, 0], "fov": 40}
    img = pta.init_debug_view({"device": device, "scene": s, "image": {"width": 32, "height": 24}, "camera": top}).draw()
    assert np.all(img == np.array([0, 0, 255, 255], np.uint8))
    desc, s, o = built("cornell")
    dv = pta.init_debug_view({"device": device, "scene": s, "image": {"width": 48, "height": 32}, "camera": desc["camera"]})
    ro, rd = dv.rays()
    ho, ht, hd = o.trace_rays(ro, rd)
    idx = dv.draw("index").reshape(-1, 4)
    hit = ho >= 0
    assert hit.mean() > 0.5
    want_r = (np.clip(ho[hit] / np.float32(20.0), 0, 1) * 255.0 + 0.5).astype(np.uint8)
    np.testing.assert_array_equal(idx[hit, 0], want_r)
    assert np.all(idx[~hit, :3] == 0)
    nrm = dv.draw("normal").reshape(-1, 4)
    assert len(np.unique(nrm[hit, :3], axis=0)) >= 3 and np.all(nrm[~hit, :3] == 0)


ONE_TRI = "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3"


def test_tolerance_mode_closest_hits(pta, device, built):
    """SURVEY 8f.4 / north_star's allowance ("hit indices bit-exact, except for counted grazing/edge rays within 1e-6"): with the
    compressed 4-wide BVH the closest hit is the reference-exact mode's except for ties between equidistant triangles; hit distances
    agree to 1e-6 relative.  (pt_trace_rays: one lane per ray, boxes decoded outwards, so no hit the reference finds can be lost.)"""
    one_tri = {"meshes": [{"file": ONE_TRI}, {"file": pta.scenes.FLOOR_OBJ}],
               "instances": [(0, [0, 0, 1], [0.3, 0.2, 0.1], [2, 2, 2], 0), (0, [1, 1, 2], [0, 0, 0], [1, 1, 1], 1), (1, [0, 0, 0], [0, 0, 0], [1, 1, 1], 0)],
               "camera": {"position": [0.5, 0.5, 6], "lookAt": [0.5, 0.6, 0], "fov": 30}}
    rng = np.random.default_rng(11)
    dev2 = pta.request_device(0)                      # its own context: `built` keeps one scene per context
    try:
        for name in ("cornell", "sphere", "instanced", "one_triangle_meshes"):
            if name == "one_triangle_meshes":
                desc, dev = one_tri, dev2
                s = pta.scenes.populate(pta.init_scene(dev2), desc); s.build()
            else:
                (desc, s, _), dev = built(name), device
            cam, look = np.asarray(desc["camera"]["position"], np.float32), np.asarray(desc["camera"]["lookAt"], np.float32)
            n = 200_000
            d = (look - cam)[None, :] + rng.normal(size=(n, 3)).astype(np.float32) * np.linalg.norm(look - cam) * 0.35
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            o = np.broadcast_to(cam, (n, 3)).copy()
            dev.call("pt_set_traversal_mode", pta.capi.PT_TRAVERSAL_REFERENCE)
            a = s.trace_rays(o, d)
            dev.call("pt_set_traversal_mode", pta.capi.PT_TRAVERSAL_WIDE)
            b = s.trace_rays(o, d)
            np.testing.assert_array_equal(a[0] >= 0, b[0] >= 0, err_msg=name)            # hit / miss never changes
            same = (a[0] == b[0]) & (a[1] == b[1])
            assert same.mean() > 0.9995, (name, same.mean())
            rel = np.abs(a[2] - b[2]) / np.maximum(np.abs(a[2]), 1e-6)
            assert rel.max() < 1e-6, (name, rel.max())                                   # a different index only on a tie
            assert (a[0] >= 0).mean() > 0.1
    finally:
        device.call("pt_set_traversal_mode", pta.capi.PT_TRAVERSAL_REFERENCE)
        dev2.destroy()


def test_tolerance_mode_render_counts_its_differing_rays(pta, orc, device, built):
    """A tolerance-mode render: every pixel gets exactly spp samples; with the check period at 1 EVERY traced ray is walked a second
    time exactly (by the workgroups in front of the next step's launch) and the differences are counted: none beyond 1e-6, no reference
    hit lost; the frame agrees with the reference-exact frame (paths only fork at a ray whose hit index differs); switching back restores
    bit-exact oracle parity.  PT_TRAVERSAL_AUTO keeps the reference walk for a scene this small."""
    import ctypes as C
    desc, s, o = built("cornell")
    w, h, spp, paths = 96, 54, 64, 20000
    def render():
        pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                                   "settings": {"samples": spp, "numPaths": paths}})
        rs = pt.run()
        return pt, rs, pt.read_image()
    mode = C.c_int(-1)
    try:
        device.call("pt_set_traversal_mode", pta.capi.PT_TRAVERSAL_AUTO)
        device.call("pt_get_traversal_mode", C.byref(mode))
        assert mode.value == pta.capi.PT_TRAVERSAL_REFERENCE
        device.call("pt_set_traversal_mode", pta.capi.PT_TRAVERSAL_WIDE)
        device.call("pt_get_traversal_mode", C.byref(mode))
        assert mode.value == pta.capi.PT_TRAVERSAL_WIDE
        device.call("pt_set_traversal_check", 1)
        _, rs_f, img_f = render()
        assert rs_f.checked_rays == rs_f.total_rays > 0
        assert rs_f.mismatched_rays <= 1e-3 * rs_f.checked_rays and rs_f.beyond_tolerance_rays == 0 and rs_f.lost_hit_rays == 0 and rs_f.flipped_rays == 0
        assert rs_f.max_rel_dist_diff <= 1e-6
        device.call("pt_set_traversal_check", 7)
        _, rs_7, img_7 = render()
        assert 0 < rs_7.checked_rays < rs_f.total_rays / 5 and rs_7.lost_hit_rays == 0
        np.testing.assert_array_equal(img_7[..., 3], img_f[..., 3])            # the re-walk never touches a result (float adds into a pixel are unordered)
        np.testing.assert_allclose(img_7[..., :3], img_f[..., :3], rtol=1e-5, atol=1e-6)
    finally:
        device.call("pt_set_traversal_check", 256)
        device.call("pt_set_traversal_mode", pta.capi.PT_TRAVERSAL_REFERENCE)
    pt, rs_e, img_e = render()
    assert rs_e.checked_rays == 0
    assert rs_f.total_samples == rs_e.total_samples == w * h * spp and rs_f.stack_overflows == 0
    np.testing.assert_array_equal(img_f[..., 3], np.full((h, w), spp, np.float32))
    rad_f, rad_e = img_f[..., :3] / spp, img_e[..., :3] / spp
    assert abs(rad_f.mean() / rad_e.mean() - 1) < 0.01
    rmse = np.sqrt(((rad_f - rad_e) ** 2).mean()) / rad_e.mean()
    assert rmse < 0.08, rmse                                                     # two 64-spp estimates of the same image
    orr = orc.OrcRender(o, w, h, desc["camera"], spp, paths); orr.run()
    np.testing.assert_array_equal(img_e[..., 3], orr.image()[..., 3])
    np.testing.assert_allclose(img_e[..., :3], orr.image()[..., :3], rtol=1e-5, atol=1e-6)


def test_tolerance_mode_deep_stack_spill_and_odd_meshes(pta):
    """The wide walk pushes up to three entries per node, so its stack outgrows the 14 LDS entries sooner than the reference walk's:
    150,000 nearly coincident triangles and rays that enter every box (nothing is pruned) drive it through the HBM spill, in the
    per-lane loop (pt_trace_rays) and in the kind-scheduled render kernel with every ray re-walked exactly: no guard hit, no hit lost,
    nothing beyond the tolerance.  Then one-triangle and two-triangle meshes (the degenerate N = 1 tree, wide nodes with one or two
    children) among nine meshes: the same."""
    rng = np.random.default_rng(21)
    n = 150_000
    base = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32)
    v = (base[None] + rng.uniform(-1e-3, 1e-3, (n, 3, 3)).astype(np.float32)).reshape(-1, 3)
    v[:, 2] += np.repeat(np.linspace(0, 0.5, n, dtype=np.float32), 3)
    deep = {"meshes": [pta.scenes.mesh_from_vertices(v, np.arange(3 * n).reshape(-1, 3))], "instances": [(0, [0, 0, 0], [0, 0, 0], [1, 1, 1], 0)],
            "camera": {"position": [-0.8, 0.8, 3], "lookAt": [-0.8, 0.8001, 0], "fov": 0.5}}
    one = "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3"
    two = "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3\nf 1 3 4"
    meshes = [{"file": one}, {"file": two}] + [pta.scenes.random_soup(40 + 13 * k, 100 + k, extent=0.6, size=0.3) for k in range(7)]
    odd = {"meshes": meshes, "instances": [(k, [1.5 * (k % 3) - 1.5, 1.5 * (k // 3) - 1.5, 0.2 * k], [0.1 * k, 0.2, 0.3 * k], [1, 1, 1], k % 2) for k in range(9)],
           "camera": {"position": [0, -7, 3], "lookAt": [0, 0, 0.5], "fov": 25}}
    for name, desc, w, h, spp, paths in (("deep", deep, 16, 16, 8, 256), ("odd", odd, 96, 64, 16, 5000)):
        dev = pta.request_device(0)
        try:
            s = pta.scenes.populate(pta.init_scene(dev), desc); s.build()
            if name == "deep":
                k = 192
                miss = np.concatenate([rng.uniform(-0.85, -0.75, (k, 1)), rng.uniform(0.75, 0.85, (k, 1)), np.full((k, 1), 3.0)], axis=1)
                hit = np.concatenate([rng.uniform(-0.3, 0.3, (k, 2)), np.full((k, 1), 3.0)], axis=1)
                ro = np.concatenate([miss, hit]).astype(np.float32); rd = np.tile(np.float32([0, 0, -1]), (2 * k, 1))
            else:
                cam = np.float32(desc["camera"]["position"])
                rd = (np.float32(desc["camera"]["lookAt"]) - cam)[None] + rng.normal(size=(20000, 3)).astype(np.float32) * 2.0
                rd /= np.linalg.norm(rd, axis=1, keepdims=True); ro = np.broadcast_to(cam, rd.shape).copy()
            a = s.trace_rays(ro, rd, counts=True)
            assert s.last_guard_hits == 0
            dev.call("pt_set_traversal_mode", pta.capi.PT_TRAVERSAL_WIDE)
            dev.call("pt_set_traversal_check", 1)
            b = s.trace_rays(ro, rd, counts=True)
            assert s.last_guard_hits == 0, name
            np.testing.assert_array_equal(a[0] >= 0, b[0] >= 0, err_msg=name)
            both = a[0] >= 0
            assert (np.abs(a[2][both] - b[2][both]) <= 1e-6 * a[2][both]).all(), name
            if name == "deep":
                assert b[3][:192, 0].min() > 30_000            # the rays through the empty half still enter every wide node
            pt = pta.init_path_tracer({"device": dev, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"], "settings": {"samples": spp, "numPaths": paths}})
            rs = pt.run()
            assert rs.total_samples == w * h * spp and rs.stack_overflows == 0, name
            assert rs.checked_rays == rs.total_rays > 0 and rs.lost_hit_rays == 0 and rs.beyond_tolerance_rays == 0, (name, rs.checked_rays, rs.total_rays, rs.lost_hit_rays, rs.beyond_tolerance_rays)
        finally:
            dev.destroy()


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_every_extend_variant_is_bit_exact(pta, orc, device, built, variant):
    """The extend kernel's variants (workgroup shape, LDS/HBM stack split, dispatch order, kind scheduling, tiny scene or the tree
    tops staged in LDS) never change a result: whole path state against the oracle after each of 8 steps, on a scene that fits
    the LDS stage whole, on one whose two meshes get 5-level top blocks and on one with three instances of two meshes."""
    import ctypes as C
    name = C.create_string_buffer(200)
    assert device.lib.pt_extend_variants(device.ctx, variant, name, 200) == 10 and name.value
    try:
        device.call("pt_set_extend_variant", variant)
        for scene_name, w, h, spp, paths in (("cornell", 64, 36, 8, 3000), ("sphere", 48, 32, 8, 2500), ("soup_ties", 48, 32, 8, 2500)):
            desc, s, o = built(scene_name)
            pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                                       "settings": {"samples": spp, "numPaths": paths}})
            orr = orc.OrcRender(o, w, h, desc["camera"], spp, paths)
            for step in range(8):
                gs = pt.step(); cs = orr.step_like_device()
                assert (gs.n_camera, gs.n_material) == (cs.n_camera, cs.n_material)
                assert_state_equal(pt, orr, step)
    finally:
        device.call("pt_set_extend_variant", -1)


def test_tree_top_blocks_with_tiny_and_many_meshes(pta, orc):
    """Extend variant 6 (TLAS, objects and the first levels of every mesh's tree staged in LDS) on the shapes its block builder has
    to get right: a one-triangle mesh (the degenerate tree with its self pointer), a two-triangle mesh, a mesh shallower than the
    block on one side, nine meshes (fewer levels fit), and a scene with too many objects for any block (falls back to variant 5).
    Whole path state against the oracle after each step."""
    one_tri = {"file": "v 0 0 0\nv 1 0 0\nv 0 1 0.3\nf 1 2 3\n"}
    sc = pta.scenes
    few = {"meshes": [sc.random_soup(3000, 11), one_tri, {"file": sc.FLOOR_OBJ}, sc.random_soup(5, 12)],
           "instances": [(0, [0, 0, 1], [0.2, 0.1, 0], [1, 1, 1], 0), (1, [1, -2, 0.5], [0, 0, 0.5], [3, 3, 3], 1), (2, [0, 0, -1], [0, 0, 0], [1, 1, 1], 0),
                         (3, [-2, 1, 2], [0.3, 0, 0], [0.5, 0.5, 0.5], 1), (1, [-3, -3, 1], [1, 0, 0], [4, 4, 4], 0)],
           "camera": {"position": [2, -14, 5], "lookAt": [0, 0, 0.5], "fov": 35}}
    nine = {"meshes": [sc.random_soup(200 + 37 * k, 20 + k) for k in range(9)],
            "instances": [(k, [3 * (k % 3) - 3, 3 * (k // 3) - 3, 0], [0, 0, 0.1 * k], [0.4, 0.4, 0.4], k & 1) for k in range(9)],
            "camera": {"position": [0, -16, 6], "lookAt": [0, 0, 0], "fov": 40}}
    many = {"meshes": [sc.random_soup(400, 31)],
            "instances": [(0, [2 * (k % 8) - 7, 2 * (k // 8) - 5, 0], [0, 0, 0.05 * k], [0.3, 0.3, 0.3], k & 1) for k in range(48)],
            "camera": {"position": [0, -20, 8], "lookAt": [0, 0, 0], "fov": 40}}
    dev = pta.request_device(0)
    try:
        dev.call("pt_set_extend_variant", 6)
        for desc in (few, nine, many):
            s = pta.scenes.populate(pta.init_scene(dev), desc); s.build()
            o = orc.OrcScene(desc)
            np.testing.assert_array_equal(s.objects_buffer(), o.objects_buffer())            # read-backs keep the reference's roots
            np.testing.assert_array_equal(s.blas_nodes()[:, NODE_COLS], o.blas_nodes()[:, NODE_COLS])
            pt = pta.init_path_tracer({"device": dev, "scene": s, "image": {"width": 48, "height": 32}, "camera": desc["camera"],
                                       "settings": {"samples": 8, "numPaths": 2500}})
            orr = orc.OrcRender(o, 48, 32, desc["camera"], 8, 2500)
            hits = 0
            for step in range(8):
                gs = pt.step(); cs = orr.step_like_device()
                assert (gs.n_camera, gs.n_material) == (cs.n_camera, cs.n_material)
                assert_state_equal(pt, orr, step)
                hits += int((pt.field("hit_obj") >= 0).sum())
            assert hits > 100
            pt.destroy()
    finally:
        dev.destroy()


def test_error_behaviour(pta, device):
    """Reference: console.warn + undefined (scene.js:149,525); here: negative status + message."""
    dev2 = pta.request_device(0)
    with pytest.raises(pta.capi.PtError, match="not been built"):
        pta.init_path_tracer({"device": dev2, "scene": None, "image": {"width": 8, "height": 8},
                              "camera": {"position": [0, 0, 0], "lookAt": [1, 0, 0], "fov": 60}, "settings": {"samples": 8}})
    desc = pta.scenes.floor_scene()
    ok = pta.scenes.populate(pta.init_scene(dev2), desc); ok.build()
    with pytest.raises(pta.capi.PtError, match="max_nb"):         # num_bounces shares a word with two other members: bounded
        pta.init_path_tracer({"device": dev2, "scene": ok, "image": {"width": 8, "height": 8}, "camera": desc["camera"],
                              "settings": {"samples": 8, "numPaths": 64, "maxNb": (1 << 22) + 1}})
    sc = pta.init_scene(dev2)
    with pytest.warns(UserWarning):
        sc.instance_mesh(3, [0, 0, 0], [0, 0, 0], [1, 1, 1], 0)
    with pytest.raises(pta.capi.PtError):
        sc.build()                               # no instances
    dev2.destroy()


def test_full_size_10m_triangle_scene(pta, orc, device):
    """BASELINE configs[3] geometry at full size (2236^2 quads = 9,999,392 triangles): BVH bit-exact against the
    oracle, closest hits + visit counts bit-exact on 200k rays, and six wavefront iterations of a 1 M-slot 4K
    render with bit-exact hit indices, queue sizes and ticket counter."""
    desc = pta.scenes.grid_scene()
    s = pta.scenes.populate(pta.init_scene(device), desc); s.build()
    o = orc.OrcScene(desc)
    assert s.info().n_tris == o.n_tris == 9_999_392
    np.testing.assert_array_equal(s.mesh_perm(0), o.mesh_perm(0))
    g_nodes = s.blas_nodes(); c_nodes = o.blas_nodes()
    np.testing.assert_array_equal(g_nodes[:, NODE_COLS], c_nodes[:, NODE_COLS])
    del g_nodes, c_nodes
    np.testing.assert_array_equal(s.blas_tris().view(np.uint32), o.blas_tris().view(np.uint32))
    ro, rd = rays_for(desc, 200_000, 9)
    g = s.trace_rays(ro, rd, counts=True); c = o.trace_rays(ro, rd, counts=True, threads=16)
    assert o.last_overflow == 0
    np.testing.assert_array_equal(g[0], c[0]); np.testing.assert_array_equal(g[1], c[1])
    np.testing.assert_array_equal(g[2].view(np.uint32), c[2].view(np.uint32)); np.testing.assert_array_equal(g[3], c[3])
    assert c[3][:, 0].max() < 5000 and (g[0] >= 0).mean() > 0.3
    w, h, spp, paths = 3840, 2160, 8, 1_000_000
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": {"samples": spp, "numPaths": paths}})
    orr = orc.OrcRender(o, w, h, desc["camera"], spp, paths)
    for step in range(6):
        gs = pt.step(); cs = orr.step_like_device(threads=16)
        assert (gs.n_camera, gs.n_material) == (cs.n_camera, cs.n_material)
        for f in ("hit_obj", "hit_tri", "pixel_index", "num_bounces"):
            np.testing.assert_array_equal(pt.field(f), orr.field(f))
        np.testing.assert_array_equal(pt.field("path_d").view(np.uint32)[:, :3], orr.field("path_d").view(np.uint32)[:, :3])
    assert pt.field("pixel_counter")[0] == orr.field("pixel_counter")[0]
    np.testing.assert_array_equal(pt.read_image()[..., 3], orr.image()[..., 3])


def test_deep_stack_uses_the_hbm_spill(pta, orc, device):
    """150,000 nearly coincident triangles.  Rays through the empty half of their common bounding box enter every
    box and hit nothing, so nothing is pruned and the near-first walk keeps one pending entry per tree level: the
    stack grows past the 16 entries held in LDS (the oracle reports the depth) and the HBM spill is exercised.
    Hits, distances and visit counts must still match the oracle bit for bit."""
    rng = np.random.default_rng(21)
    n = 150_000
    base = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32)
    v = (base[None] + rng.uniform(-1e-3, 1e-3, (n, 3, 3)).astype(np.float32)).reshape(-1, 3)
    v[:, 2] += np.repeat(np.linspace(0, 0.5, n, dtype=np.float32), 3)
    mesh = pta.scenes.mesh_from_vertices(v, np.arange(3 * n).reshape(-1, 3))
    desc = {"meshes": [mesh], "instances": [(0, [0, 0, 0], [0, 0, 0], [1, 1, 1], 0)], "camera": None}
    s = pta.scenes.populate(pta.init_scene(device), desc); s.build()
    o = orc.OrcScene(desc)
    np.testing.assert_array_equal(s.blas_nodes()[:, NODE_COLS], o.blas_nodes()[:, NODE_COLS])
    k = 192
    miss = np.concatenate([rng.uniform(-0.85, -0.75, (k, 1)), rng.uniform(0.75, 0.85, (k, 1)), np.full((k, 1), 3.0)], axis=1)
    hit = np.concatenate([rng.uniform(-0.3, 0.3, (k, 2)), np.full((k, 1), 3.0)], axis=1)
    ro = np.concatenate([miss, hit]).astype(np.float32)
    rd = np.tile(np.float32([0, 0, -1]), (2 * k, 1))
    g = s.trace_rays(ro, rd, counts=True); c = o.trace_rays(ro, rd, counts=True, threads=16)
    assert o.last_overflow == 0 and o.last_max_depth > 16, o.last_max_depth
    np.testing.assert_array_equal(g[0], c[0]); np.testing.assert_array_equal(g[1], c[1])
    np.testing.assert_array_equal(g[2].view(np.uint32), c[2].view(np.uint32)); np.testing.assert_array_equal(g[3], c[3])
    assert np.all(c[0][:k] == -1) and np.all(c[0][k:] == 0) and c[3][:k, 0].min() > 100_000
    # the same geometry through the render path's extend kernel (64-thread workgroups, 16 LDS entries): a camera in
    # the empty corner looking straight down
    cam = {"position": [-0.8, 0.8, 3], "lookAt": [-0.8, 0.8001, 0], "fov": 0.5}
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": 16, "height": 16}, "camera": cam, "settings": {"samples": 8, "numPaths": 256}})
    orr = orc.OrcRender(o, 16, 16, cam, 8, 256)
    for _ in range(2):
        gs = pt.step(); cs = orr.step_like_device(threads=16)
        assert gs.n_rays == cs.n_rays
        np.testing.assert_array_equal(pt.field("hit_tri"), orr.field("hit_tri"))
    rs = pt.run(max_steps=2, count_bytes=True)
    assert rs.stack_overflows == 0 and rs.traversal_bytes > 256 * 64 * 100_000


def test_staged_scene_stack_bound(pta, orc, device, built):
    """A scene small enough for the LDS stage gets a stack bound from the built trees.  The Cornell box is shallow (its
    staged kernel has no spill path and no overflow guard: that is what every Cornell test above runs) and no oracle ray
    ever holds more entries than the bound; a 31-triangle mesh whose Morton codes are 0 and the thirty single bits builds
    a chain as deep as the code is long, is over the bound of the LDS stack, and takes the staged kernel with the spill
    path -- both bit-exact against the oracle, whole path state after every step."""
    desc, s, o = built("cornell")
    assert 0 < s.stack_bound() <= 12
    ro, rd = rays_for(desc, 20000, 5)
    o.trace_rays(ro, rd, counts=True)
    assert o.last_max_depth <= s.stack_bound()
    sb = built("sphere")[1].stack_bound()                            # too large to stage, but proper trees: a bound within the 64-entry stack,
    assert 12 < sb <= 64, sb                                         # so the default kernel walks without its guards (Stack GUARDED = false)
    ro2, rd2 = rays_for(built("sphere")[0], 20000, 6)
    built("sphere")[2].trace_rays(ro2, rd2, counts=True)
    assert built("sphere")[2].last_max_depth <= sb
    pos = [np.zeros(3)]
    for axis in range(3):
        for j in range(10):
            p = np.zeros(3); p[axis] = (2.0 ** j + 0.5) / 1024.0; pos.append(p)
    pos = np.asarray(pos)
    eps = 1e-4
    tri = np.array([[-eps, -eps, 0], [eps, -eps, 0], [0, eps, 0]])
    v = (pos[:, None, :] + tri[None]).reshape(-1, 3).astype(np.float32)
    mesh = pta.scenes.mesh_from_vertices(v, np.arange(len(v)).reshape(-1, 3))
    cam = {"position": [-1.0, 0.2, 0.3], "lookAt": [0.3, 0.0, 0.0], "fov": 40.0}
    # two instances: a one-instance TLAS is the degenerate tree whose right child is the node itself, for which no bound exists
    cdesc = {"meshes": [mesh], "instances": [(0, [0, 0, 0], [0, 0, 0], [1, 1, 1], 0), (0, [0.1, 0.3, -0.2], [0.2, 0.1, 0.4], [1, 1, 1], 1)], "camera": cam}
    assert built("floor")[1].stack_bound() == 2**31 - 1              # one instance: unknown, the guarded kernel runs
    dev2 = pta.request_device(0)
    try:
        cs = pta.scenes.populate(pta.init_scene(dev2), cdesc); cs.build()
        co = orc.OrcScene(cdesc)
        np.testing.assert_array_equal(cs.blas_nodes()[:, NODE_COLS], co.blas_nodes()[:, NODE_COLS])
        assert 12 < cs.stack_bound() <= 31, cs.stack_bound()
        w, h, spp, paths = 48, 32, 8, 2000
        pt = pta.init_path_tracer({"device": dev2, "scene": cs, "image": {"width": w, "height": h}, "camera": cam, "settings": {"samples": spp, "numPaths": paths}})
        orr = orc.OrcRender(co, w, h, cam, spp, paths)
        for step in range(6):
            gs = pt.step(); c_ = orr.step_like_device()
            assert (gs.n_camera, gs.n_material) == (c_.n_camera, c_.n_material)
            assert_state_equal(pt, orr, step)
    finally:
        dev2.destroy()


def test_rccl_reduce_at_world_size_one(pta, device, built):
    """SURVEY 8e in the library: pt_comm_create over this box's one GPU (and the process-per-GPU form with a unique id), one RCCL
    sum-reduce of the accumulation buffer onto rank 0: the frame must come through untouched, also in the middle of a render
    (the reduce first adds the unfinished tickets' partial sums, like pt_image_read).  Two contexts on one GPU are refused."""
    desc, s, o = built("cornell")
    w, h, spp, paths = 64, 36, 16, 3000
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": {"samples": spp, "numPaths": paths}})
    pt.run(max_steps=6)
    comm = pta.multigpu.LibComm([device])
    try:
        comm.reduce_image(0)
        mid = pt.read_image().copy()
        assert 0 < mid[..., 3].sum() < w * h * spp
        pt.run()
        before = pt.read_image().copy()
        comm.reduce_image(0)
        np.testing.assert_array_equal(pt.read_image().view(np.uint32), before.view(np.uint32))
        np.testing.assert_array_equal(before[..., 3], np.full((h, w), spp, np.float32))
        with pytest.raises(pta.capi.PtError, match="bad argument"):
            comm.reduce_image(1)
    finally:
        comm.destroy()
    c2 = pta.multigpu.LibComm([device], id128=pta.multigpu.LibComm.unique_id(), rank=0, world=1)
    try:
        c2.reduce_image(0)
        np.testing.assert_array_equal(pt.read_image().view(np.uint32), before.view(np.uint32))
    finally:
        c2.destroy()
    dev2 = pta.request_device(0)
    try:
        with pytest.raises(pta.capi.PtError, match="one communicator rank per GPU"):
            pta.multigpu.LibComm([device, dev2])
    finally:
        dev2.destroy()


def test_reduce_semantics_at_world_size_two():
    """pt_comm_reduce_image with TWO ranks (a second in-place sum of intact buffers would double-count): the reduce moves the
    samples to the root and empties the other ranks, so a reduce in the middle of a render followed by one at its end gives exactly spp
    samples per pixel and the oracle's sum of the two shares.  RCCL refuses two ranks on this box's one GPU, so the ranks are two contexts
    of one process on the library's loopback backend (PT_COMM_LOOPBACK=1: the same pt_comm code path, the sum by a kernel instead of
    ncclReduce); the backend is fixed per process, hence the child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "loopback_reduce_check.py")], env=dict(os.environ, PT_COMM_LOOPBACK="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "loopback reduce ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_pools_on_one_gpu(pta, orc, device, built):
    """settings.pools = 3: three path pools that borrow one scene and add into one image, driven together by
    pt_render_run_many.  Every pool traces exactly the rays of the oracle's run of the same share (its own slot base and
    third of the tickets), the image has exactly spp samples everywhere and the radiance of the three oracle shares
    added up (float-add order tolerance); reset() + run() reproduces it; a pool context refuses to build scenes."""
    desc, s, o = built("cornell")
    w, h, spp, paths, k_pools = 96, 54, 24, 4000, 3
    pt = pta.init_path_tracer({"device": device, "scene": s, "image": {"width": w, "height": h}, "camera": desc["camera"],
                               "settings": {"samples": spp, "numPaths": paths, "pools": k_pools}})
    assert isinstance(pt, pta.host.PooledPathTracer) and len(pt.pools) == k_pools
    rs = pt.run()
    img = pt.read_image()
    total = (w * h * spp + 7) // 8
    ref = np.zeros((h, w, 4), np.float64); rays = []
    for k in range(k_pools):
        orr = orc.OrcRender(o, w, h, desc["camera"], spp, paths, slot_base=k * paths, ticket_begin=total * k // k_pools,
                            ticket_end=total * (k + 1) // k_pools)
        steps, n_rays, _ = orr.run()
        rays.append(n_rays); ref += orr.image()
        assert pt.last_pool_stats[k].total_rays == n_rays and pt.last_pool_stats[k].steps == steps
    assert rs.total_rays == sum(rays) and rs.total_samples == w * h * spp
    np.testing.assert_array_equal(img[..., 3], np.full((h, w), spp, np.float32))
    np.testing.assert_allclose(img[..., :3], ref[..., :3], rtol=2e-5, atol=1e-5)
    pt.reset(); rs2 = pt.run()
    assert rs2.total_rays == rs.total_rays
    img2 = pt.read_image()
    np.testing.assert_array_equal(img2[..., 3], img[..., 3])
    np.testing.assert_allclose(img2[..., :3], img[..., :3], rtol=2e-5, atol=1e-5)
    with pytest.raises(pta.capi.PtError, match="borrows"):
        pta.init_scene(pt.pools[1].device).register_mesh(pta.scenes.cornell_box()["meshes"][0])
    # the owner keeps count of its borrowers: it cannot rebuild its scene, replace the image they add into, or go away
    with pytest.raises(pta.capi.PtError, match="still borrow"):
        s.build()
    with pytest.raises(pta.capi.PtError, match="accumulates into"):
        pta.init_path_tracer({"device": device, "scene": s, "image": {"width": 8, "height": 8}, "camera": desc["camera"], "settings": {"samples": 8, "numPaths": 64}})
    assert device.lib.pt_destroy(device.ctx) == -3 and b"still borrow" in device.lib.pt_last_error(device.ctx)      # PT_ERR_STATE
    import ctypes as C
    twice = (C.c_void_p * 2)(pt.pools[1].device.ctx, pt.pools[1].device.ctx)
    assert device.lib.pt_render_run_many(twice, 2, 0, 0, None) == -1          # PT_ERR_ARG
    assert b"twice" in device.lib.pt_last_error(pt.pools[1].device.ctx)
    pt.destroy()
    s.build()                                                                 # borrowers gone: allowed again
