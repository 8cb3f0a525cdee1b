"""Pins for the CPU oracle.  The reference has no tests or golden vectors for its device code (SURVEY 4, 8c:
"parity unpinned"), so the oracle is pinned by independent properties instead:
stable-sort agreement, Karras-tree invariants, box containment, the N=1 layout, brute-force closest hit,
closed-form radiance, and the math spec against libm."""
import ctypes as C

import numpy as np
import pytest


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------------------------------ math
def test_sincos_spec_accuracy(orc):
    th = np.concatenate([np.linspace(0, 6.2831855, 20001, dtype=np.float32), np.float32([0, 1.5707964, 3.1415927, 4.712389, 6.2831855])])
    got = np.array([orc.sincos(t) for t in th], dtype=np.float64)
    assert np.abs(got[:, 0] - np.sin(th.astype(np.float64))).max() < 2.5e-7
    assert np.abs(got[:, 1] - np.cos(th.astype(np.float64))).max() < 2.5e-7


def test_hash_and_rand2(orc):
    L = orc.lib()
    # material.js:154-158 evaluated by hand in numpy uint32 arithmetic
    def ref_hash(px, py):
        px, py = np.uint32(px), np.uint32(py)
        with np.errstate(over="ignore"):
            p2x = np.uint32(1103515245) * ((px >> np.uint32(1)) ^ py)
            p2y = np.uint32(1103515245) * ((py >> np.uint32(1)) ^ px)
            h = np.uint32(1103515245) * (p2x ^ (p2y >> np.uint32(3)))
        return int(h ^ (h >> np.uint32(16)))
    for a, b in [(0, 1), (1, 2), (123456, 123457), (0xffffffff, 0), (0x3f800000, 0x40000000)]:
        assert L.orc_base_hash(a, b) == ref_hash(a, b)
    r0, r1 = C.c_float(), C.c_float()
    for seed in [0.008, 0.5, 1.008, 123.008]:
        L.orc_rand2(seed, C.byref(r0), C.byref(r1))
        s = np.float32(seed)
        n = ref_hash((s + np.float32(1)).view(np.uint32), (s + np.float32(2)).view(np.uint32))
        assert r0.value == np.float32(np.float32(n & 0x7fffffff) / np.float32(2147483648.0))
        assert r1.value == np.float32(np.float32((n * 48271) & 0x7fffffff) / np.float32(2147483648.0))
        assert 0.0 <= r0.value <= 1.0 and 0.0 <= r1.value <= 1.0


def test_f32_to_u32_edges(orc):
    L = orc.lib()
    assert [L.orc_f32_to_u32(x) for x in (float("nan"), -1.0, -0.0, 0.0, 0.99, 1023.99, 4294967296.0, float("inf"))] == \
        [0, 0, 0, 0, 0, 1023, 0xffffffff, 0xffffffff]


# ------------------------------------------------------------------------------------------ build
def test_sort_is_stable_sort(orc):
    L = orc.lib()
    rng = np.random.default_rng(0)
    for n, hi in [(1, 10), (2, 2), (1000, 16), (100_000, 1 << 30), (65_537, 3)]:
        keys = rng.integers(0, hi, n, dtype=np.uint32)
        k = keys.copy(); idx = np.empty(n, np.int32)
        L.orc_stable_sort(_p(k), _p(idx), n)
        want = np.argsort(keys, kind="stable")
        np.testing.assert_array_equal(idx, want)
        np.testing.assert_array_equal(k, keys[want])


def build_tree(orc, keys_sorted):
    n = len(keys_sorted)
    par = np.empty(2 * n, np.int32)
    orc.lib().orc_radix_tree(_p(keys_sorted), n, _p(par))
    return par


@pytest.mark.parametrize("n,hi", [(2, 1 << 30), (3, 4), (257, 1 << 30), (5000, 64), (5000, 1), (40_000, 1 << 30)])
def test_karras_tree_invariants(orc, n, hi):
    """Every inner node covers a contiguous leaf range, node 0 is the root, every leaf and every inner node
    but the root has exactly one parent, the split is the largest s with a longer common prefix."""
    rng = np.random.default_rng(n + hi)
    keys = np.sort(rng.integers(0, hi, n, dtype=np.uint32))
    par = build_tree(orc, keys)
    children = {i: [] for i in range(n - 1)}
    for c in range(1, n - 1):
        children[par[c]].append(c)
    for leaf in range(n):
        children[par[n + leaf]].append(-(leaf + 1))
    assert all(len(v) == 2 for v in children.values())
    # leaf ranges by post-order from the root
    lo = np.zeros(n - 1, np.int64); hi_ = np.zeros(n - 1, np.int64); cnt = np.zeros(n - 1, np.int64)
    order, stack = [], [0]
    while stack:
        x = stack.pop(); order.append(x)
        stack.extend(c for c in children[x] if c >= 0)
    assert len(order) == n - 1                       # all inner nodes reachable from node 0
    for x in reversed(order):
        rngs = [(-(c + 1), -(c + 1), 1) if c < 0 else (lo[c], hi_[c], cnt[c]) for c in children[x]]
        lo[x] = min(r[0] for r in rngs); hi_[x] = max(r[1] for r in rngs); cnt[x] = sum(r[2] for r in rngs)
        assert cnt[x] == hi_[x] - lo[x] + 1          # contiguous
        a, b = sorted(rngs)                          # left child covers the lower keys
        assert a[1] + 1 == b[0]
        assert x in (lo[x], hi_[x])                  # Karras: node i sits at one end of its range
    assert (lo[0], hi_[0]) == (0, n - 1)


def test_up_pass_boxes_and_nudge(orc):
    L = orc.lib()
    rng = np.random.default_rng(5)
    n = 3000
    tris = np.full((n, 12), 3.1415, np.float32)
    for k in range(3):
        tris[:, 4 * k:4 * k + 3] = rng.uniform(-2, 2, (n, 3)) + rng.uniform(-0.2, 0.2, (n, 3))
    tris[::7, 2] = 0.0; tris[::7, 6] = 0.0; tris[::7, 10] = 0.0          # flat at z = 0 -> exact zeros to nudge
    bmin = tris.reshape(n, 3, 4)[:, :, :3].min((0, 1)).astype(np.float32); bmax = tris.reshape(n, 3, 4)[:, :, :3].max((0, 1)).astype(np.float32)
    aabb = np.empty((n, 8), np.float32); keys = np.empty(n, np.uint32); idx = np.empty(n, np.int32); nodes = np.zeros((n, 16), np.float32)
    L.orc_tri_aabb_morton(_p(tris), n, _p(bmin), _p(bmax), _p(aabb), _p(keys))
    L.orc_build_bvh(_p(aabb), _p(keys), n, _p(idx), _p(nodes))
    ni = nodes.view(np.int32)
    assert np.all(np.diff(keys.astype(np.int64)) >= 0)
    v = tris.reshape(n, 3, 4)[idx][:, :, :3]                           # sorted order
    lo_leaf, hi_leaf = v.min(1), v.max(1)

    def box_of(child):      # exact box of a subtree, recursively
        if child < 0:
            t = -(child + 1); return lo_leaf[t].copy(), hi_leaf[t].copy()
        a = box_of(int(ni[child, 3])); b = box_of(int(ni[child, 11]))
        return np.minimum(a[0], b[0]), np.maximum(a[1], b[1])
    import sys
    sys.setrecursionlimit(10000)
    for node in rng.integers(0, n - 1, 200):
        for off, c in ((0, 3), (8, 11)):
            lo, hi = box_of(int(ni[node, c]))
            lo = np.where(lo == 0, np.float32(-1e-8), lo); hi = np.where(hi == 0, np.float32(1e-8), hi)   # bvh-up-pass.js:209-210
            got_lo, got_hi = nodes[node, off:off + 3], nodes[node, off + 4:off + 7]
            # a nudged zero can only lose against a real negative / positive value further up
            assert np.all(got_lo <= lo) and np.all(got_hi >= hi)
            assert np.all((got_lo == lo) | (lo == np.float32(-1e-8)) | (got_lo == np.float32(-1e-8)))
    assert not np.any(nodes[: n - 1, [0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14]] == 0.0)   # 0 means "not written" in the reference
    assert np.all(ni[: n - 1, 7] == 2) and np.all(ni[:, 15] == 0)


def test_n1_degenerate_layout(orc, pta):
    """SURVEY 'Degenerate N=1': node 0 has a real l child and an all-zero r child with r_child = 0."""
    s = orc.OrcScene(pta.scenes.floor_scene())
    t = s.tlas_nodes(); ti = t.view(np.int32); tf = t.view(np.float32)
    assert ti[0, 3] == -1 and ti[0, 11] == 0 and ti[0, 7] == 1
    np.testing.assert_array_equal(tf[0, 0:3], np.float32([-40, -40, -1.3])); np.testing.assert_array_equal(tf[0, 4:7], np.float32([60, 60, -1.3]))
    assert not t[0, 8:16].any()
    b = s.blas_nodes(); bi = b.view(np.int32)
    assert s.n_tris == 2 and bi[0, 3] == -1 and bi[0, 11] == -2 and not b[1].any()
    assert b.view(np.float32)[0, 2] == np.float32(-1e-8) and b.view(np.float32)[0, 6] == np.float32(1e-8)   # flat z nudged
    ob = s.objects_buffer()
    np.testing.assert_array_equal(ob.view(np.float32)[0, 0:4], np.float32([5, 0, 0, 10]))                  # row 0 of localToWorld
    assert ob.view(np.int32)[0, 12] == 0 and ob.view(np.int32)[0, 13] == 0


def test_pack_rebases_children_and_orders_meshes(orc, pta):
    """pack-t-mesh.js: mesh order = ascending id of USED meshes, offsets = running triangle count."""
    sc = pta.scenes
    desc = {"meshes": [sc.random_soup(10, 1), sc.random_soup(7, 2), sc.random_soup(5, 3)],
            "instances": [(2, [0, 0, 0], [0, 0, 0], [1, 1, 1], 1), (0, [9, 0, 0], [0, 0, 0], [1, 1, 1], 0)], "camera": None}
    s = orc.OrcScene(desc)
    assert s.n_tris == 15                                  # mesh 1 unused
    bi = s.blas_nodes().view(np.int32)
    assert set(np.concatenate([-(bi[:9, [3, 11]].ravel()[bi[:9, [3, 11]].ravel() < 0]) - 1])) == set(range(10))
    leaves2 = bi[10:14, [3, 11]].ravel(); leaves2 = -(leaves2[leaves2 < 0]) - 1
    assert set(leaves2) == set(range(10, 15))
    inner2 = bi[10:14, [3, 11]].ravel(); assert np.all(inner2[inner2 >= 0] >= 10)
    ob = s.objects_buffer().view(np.int32)
    perm = s.tlas_perm()
    assert [ob[i, 12] for i in range(2)] == [[10, 0][p] for p in perm]     # bvh_offset follows the TLAS order
    assert [ob[i, 13] for i in range(2)] == [[1, 0][p] for p in perm]


# -------------------------------------------------------------------------------------- traversal
@pytest.mark.parametrize("which", ["cornell", "soup", "instanced"])
def test_bvh_traversal_equals_brute_force(orc, pta, which):
    sc = pta.scenes
    desc = {"cornell": sc.cornell_box(), "instanced": sc.instanced_scene(n_side=6, grid=3),
            "soup": {"meshes": [sc.random_soup(600, 4)], "instances": [(0, [0, 0, 0], [0.3, 0.1, 0.2], [1, 2, 1], 0)],
                     "camera": {"position": [0, -14, 3], "lookAt": [0, 0, 0], "fov": 25}}}[which]
    s = orc.OrcScene(desc)
    rng = np.random.default_rng(11)
    n = 60_000
    cam = np.asarray(desc["camera"]["position"], np.float32)
    o = np.where(rng.random((n, 1)) < 0.5, cam, rng.normal(0, 3, (n, 3))).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32) + (np.asarray(desc["camera"]["lookAt"], np.float32) - o) * 0.5
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ho, ht, hd = s.trace_rays(o, d)
    bo, bt, bd = s.brute_force(o, d)
    assert s.last_overflow == 0
    agree = (ho == bo) & (ht == bt)
    # disagreements may only be grazing / tie cases: both found a hit at (nearly) the same distance
    bad = ~agree     # (the Cornell boxes stand ON the floor: coplanar bottom faces tie with the floor triangles)
    assert bad.mean() < (1e-2 if which == "cornell" else 2e-4), bad.mean()
    assert np.all((ho[bad] >= 0) & (bo[bad] >= 0)) and np.all(np.abs(hd[bad] - bd[bad]) <= 1e-4 * np.maximum(1, bd[bad]))
    hit = agree & (ho >= 0)
    assert hit.mean() > 0.1 and np.all(np.abs(hd[hit] - bd[hit]) <= 2e-5 * np.maximum(1, bd[hit]))


def test_visit_counts_and_guards(orc, pta):
    s = orc.OrcScene(pta.scenes.floor_scene())
    # a ray exactly through the world origin meets the all-zero r box of the N=1 TLAS node (r_child = 0 self loop):
    # the guards must end it, and the real hit must survive
    o = np.float32([[3, 3, 3], [15, 15, 5]]); d = np.float32([[-1, -1, -1], [-5, -5, -5]])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ho, ht, hd, cn = s.trace_rays(o, d, counts=True)
    assert ho[1] == 0 and ht[1] in (0, 1)
    assert s.last_overflow <= 1


# ----------------------------------------------------------------------------------------- render
def test_floor_closed_form_radiance(orc, pta):
    """a floor sample is exactly one Lambert bounce into the sky (2 * 0.5^2.2), a sky sample is 2."""
    desc = pta.scenes.floor_scene()
    s = orc.OrcScene(desc)
    w = h = 64
    r = orc.OrcRender(s, w, h, desc["camera"], 16, 5000)
    steps, rays, byts = r.run()
    img = r.image()
    np.testing.assert_array_equal(img[..., 3], np.full((h, w), 16, np.float32))
    rad = img[..., 0] / img[..., 3]
    fv = 2 * 0.5 ** 2.2
    floor_px = np.abs(rad - fv) < 1e-5; sky_px = np.abs(rad - 2.0) < 1e-6
    assert floor_px.mean() > 0.3 and (~(floor_px | sky_px)).mean() < 0.08
    assert np.array_equal(img[..., 0], img[..., 1]) and np.array_equal(img[..., 0], img[..., 2])
    # every path: camera ray (+ one bounce ray if it hit) + one wasted ray per slot killed by ticket exhaustion
    assert rays >= w * h * 16 and byts > 44 * rays


def test_furnace_energy_bound(orc, pta):
    """Inside the Cornell box radiance never exceeds the sky's 2.0 and is positive where the sky is reachable."""
    desc = pta.scenes.cornell_box()
    s = orc.OrcScene(desc)
    r = orc.OrcRender(s, 64, 36, desc["camera"], 32, 4000); r.run()
    img = r.image(); rad = img[..., :3] / img[..., 3:]
    assert np.all(img[..., 3] == 32) and rad.max() <= 2.0 + 1e-5 and rad.min() >= 0.0 and rad.mean() > 0.3


def test_state_machine_consequences(orc, pta):
    """num_bounces advances by 2 per bounce, RR starts at nb == 5, nothing survives nb > 20."""
    desc = pta.scenes.cornell_box()
    s = orc.OrcScene(desc)
    r = orc.OrcRender(s, 32, 18, desc["camera"], 64, 2000)
    seen = set()
    for _ in range(40):
        r.step(); seen.update(np.unique(r.field("num_bounces")).tolist())
    assert seen <= {0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20} and {0, 2, 4} <= seen


def test_ticket_exhaustion_kills_slots(orc, pta):
    desc = pta.scenes.floor_scene()
    s = orc.OrcScene(desc)
    r = orc.OrcRender(s, 8, 8, desc["camera"], 8, 100)       # 64 tickets for 100 slots
    st = r.step()
    assert st.n_camera == 100 and st.n_live == 64             # dead slots still trace their first ray (logic.js:187-190)
    assert (r.field("flags") >> 31).sum() == 36
    r.run()
    assert r.image()[..., 3].sum() == 8 * 8 * 8


def test_device_phase_order_is_the_reference_order(orc):
    """The HIP step runs the next step's logic right behind the trace.  The oracle can be stepped either way; both must
    trace identical queues every step and end in the same state and image, which is what lets the GPU parity tests compare
    at the device's step boundary."""
    pta = orc.pta
    desc = pta.scenes.cornell_box()
    s = orc.OrcScene(desc)
    a = orc.OrcRender(s, 48, 27, desc["camera"], 16, 1500)
    b = orc.OrcRender(s, 48, 27, desc["camera"], 16, 1500)
    for step in range(40):
        sa = a.step(); sb = b.step_like_device()
        assert (sa.n_camera, sa.n_material) == (sb.n_camera, sb.n_material)
        for q in (0, 1):
            np.testing.assert_array_equal(a.queue(q), b.queue(q))
        for f in ("hit_obj", "hit_tri", "pixel_index"):
            np.testing.assert_array_equal(a.field(f), b.field(f))
    ra = a.run(); rb_steps = 0
    while True:
        st = b.step_like_device(); rb_steps += 1
        if st.n_rays == 0:
            break
    np.testing.assert_array_equal(a.image().view(np.uint32), b.image().view(np.uint32))
    for f in orc.FIELDS:
        np.testing.assert_array_equal(a.field(f).view(np.uint32), b.field(f).view(np.uint32))


# ---- extensions: rules defined by the oracle, nothing in the reference to pin them -- closed forms instead
def _plane_scene(pta, material):
    """the reference's floor quad, 100 x 100 units, seen from straight above"""
    return {"meshes": [{"file": pta.scenes.FLOOR_OBJ}], "instances": [(0, [0, 0, 0], [0, 0, 0], [5, 5, 5], material)],
            "camera": {"position": [0, 0, 8], "lookAt": [0, 0.001, 0], "fov": 20}}


def test_extensions_default_off(orc, pta):
    desc = pta.scenes.cornell_box()
    s = orc.OrcScene(desc)
    a = orc.OrcRender(s, 32, 18, desc["camera"], 16, 700)
    b = orc.OrcRender(s, 32, 18, desc["camera"], 16, 700, ext={"maxNb": 0, "seed": 0, "materials": []})
    a.run(); b.run()
    np.testing.assert_array_equal(a.image().view(np.uint32), b.image().view(np.uint32))
    for f in orc.FIELDS:
        np.testing.assert_array_equal(a.field(f).view(np.uint32), b.field(f).view(np.uint32))


def test_extension_emitter_and_mirror_closed_forms(orc, pta):
    e = [3.0, 1.5, 0.25]
    for mats, expect in (([{"type": "emitter", "emission": e}], e),                         # first hit ends the path: exactly e
                         ([{"type": "mirror", "albedo": [0.5, 0.25, 1.0]}], [1.0, 0.5, 2.0]),    # one bounce into the sky (2): 2 * albedo
                         # glossy: whatever the roughness, the blurred mirror direction stays above the plane: one bounce into the sky
                         ([{"type": "glossy", "albedo": [0.5, 0.25, 1.0], "roughness": 0.0}], [1.0, 0.5, 2.0]),
                         ([{"type": "glossy", "albedo": [0.25, 0.5, 0.125], "roughness": 0.6}], [0.5, 1.0, 0.25]),
                         ([{"type": "glossy", "albedo": [0.5, 0.5, 0.5], "roughness": 7.0}], [1.0, 1.0, 1.0])):  # roughness is clamped to 1
        desc = _plane_scene(pta, 2)
        r = orc.OrcRender(orc.OrcScene(desc), 16, 16, desc["camera"], 8, 300, ext={"materials": mats})
        r.run()
        img = r.image()
        assert np.all(img[..., 3] == 8)
        np.testing.assert_array_equal(img[..., :3] / img[..., 3:4], np.broadcast_to(np.float32(expect), (16, 16, 3)))
    # a material id without an entry keeps the reference's unknown-id rule: zero BRDF, the path contributes nothing
    desc = _plane_scene(pta, 5)
    r = orc.OrcRender(orc.OrcScene(desc), 8, 8, desc["camera"], 8, 100, ext={"materials": [{"type": "emitter", "emission": e}]})
    r.run()
    assert np.all(r.image()[..., :3] == 0) and np.all(r.image()[..., 3] == 8)


def test_extension_max_nb_and_seed(orc, pta):
    desc = pta.scenes.cornell_box()
    s = orc.OrcScene(desc)
    def bounces(ext):
        r = orc.OrcRender(s, 32, 18, desc["camera"], 16, 700, ext=ext)
        seen = set()
        for _ in range(40):
            r.step(); seen.update(np.unique(r.field("num_bounces")).tolist())
        return r, seen
    r0, seen0 = bounces(None)
    r1, seen1 = bounces({"maxNb": 4})
    assert max(seen0) > 6 and max(seen1) <= 6          # nb advances by 2 per bounce; a path is cut once nb > max_nb
    r2, _ = bounces({"seed": 7}); r3, _ = bounces({"seed": 7}); r4, _ = bounces({"seed": 8})
    np.testing.assert_array_equal(r2.field("random_seed"), r3.field("random_seed"))
    assert not np.array_equal(r2.field("random_seed"), r0.field("random_seed"))
    assert not np.array_equal(r2.field("random_seed"), r4.field("random_seed"))
