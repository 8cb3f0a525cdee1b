"""Seeded synthetic scenes for the configurations BASELINE.json names.

Nothing here comes from the reference except the 20x20 floor quad, which is the one asset it
bundles (media/floor.obj, data only).  The reference's demo meshes (cow/bunny/buddha, main.js:11-14)
are git-ignored there and absent, so same-scale procedural stand-ins are generated instead.
A scene description is a dict:
    meshes    : list of {'triangles': f32[n,12] (parse-obj.js:64-68 layout), 'bounds': {'min','max'}} or {'file': objText}
    instances : list of (mesh_index, position, rotationXYZ, scale, material)   -> scene.instanceMesh(...)
    camera    : {'position','lookAt','fov'}                                      -> initPathTracer camera
"""
import math

import numpy as np

FLOOR_OBJ = "v -10.0 -10.0 0.0\nv  10.0 -10.0 0.0\nv  10.0  10.0 0.0\nv -10.0  10.0 0.0\n\nf 1 2 3\nf 1 3 4"

PAD = np.float32(3.1415)   # parse-obj.js:65-67 pads every vec3 with 3.1415


def mesh_from_vertices(verts, faces):
    """verts f32[nv,3], faces int[nt,3] -> mesh dict with bounds over all vertices (parse-obj.js:33-38)."""
    verts = np.asarray(verts, dtype=np.float32)
    faces = np.asarray(faces, dtype=np.int64)
    tris = np.full((faces.shape[0], 12), PAD, dtype=np.float32)
    for k in range(3):
        tris[:, 4 * k:4 * k + 3] = verts[faces[:, k]]
    v64 = verts.astype(np.float64)
    return {"triangles": tris, "bounds": {"min": v64.min(axis=0).tolist(), "max": v64.max(axis=0).tolist()}}


def _quad(a, b, c, d):
    return [[a, b, c], [a, c, d]]


def _box_faces(lo, hi, rot_z=0.0, center=None):
    """axis-aligned box lo..hi rotated about z around its centre: 8 verts, 12 tris (outward CCW)."""
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    cx = (lo + hi) / 2 if center is None else np.asarray(center, float)
    v = np.array([[x, y, z] for z in (lo[2], hi[2]) for y in (lo[1], hi[1]) for x in (lo[0], hi[0])], float)
    c, s = math.cos(rot_z), math.sin(rot_z)
    p = v - cx
    v = np.stack([c * p[:, 0] - s * p[:, 1], s * p[:, 0] + c * p[:, 1], p[:, 2]], axis=1) + cx
    f = (_quad(0, 2, 3, 1) + _quad(4, 5, 7, 6) + _quad(0, 1, 5, 4) + _quad(2, 6, 7, 3) + _quad(0, 4, 6, 2) + _quad(1, 3, 7, 5))
    return v, np.asarray(f)


def cornell_box(single_mesh=False):
    """C2: 5 walls (10 tris) + a tall and a short box (24 tris) = 34 triangles, z up, open towards -y
    so the constant sky (logic.js:116) lights it.  Materials are per instance in the reference
    (material.js:104), so the walls (material 0, albedo .5) and the boxes (material 1, albedo .2) are
    two meshes / two instances; single_mesh=True folds them into one instance (the N=1 TLAS case)."""
    wv = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0], [-1, -1, 2], [1, -1, 2], [1, 1, 2], [-1, 1, 2]], float)
    wf = np.asarray(_quad(0, 1, 2, 3) + _quad(4, 7, 6, 5) + _quad(3, 2, 6, 7) + _quad(0, 3, 7, 4) + _quad(1, 5, 6, 2))
    b1v, b1f = _box_faces([-0.65, -0.05, 0.0], [-0.05, 0.55, 1.2], rot_z=0.30)
    b2v, b2f = _box_faces([0.10, -0.60, 0.0], [0.70, 0.00, 0.6], rot_z=-0.28)
    bv = np.concatenate([b1v, b2v]); bf = np.concatenate([b1f, b2f + 8])
    cam = {"position": [0.0, -4.6, 1.0], "lookAt": [0.0, 0.0, 1.0], "fov": 20.0}
    if single_mesh:
        m = mesh_from_vertices(np.concatenate([wv, bv]), np.concatenate([wf, bf + 8]))
        return {"meshes": [m], "instances": [(0, [0, 0, 0], [0, 0, 0], [1, 1, 1], 0)], "camera": cam}
    return {"meshes": [mesh_from_vertices(wv, wf), mesh_from_vertices(bv, bf)],
            "instances": [(0, [0, 0, 0], [0, 0, 0], [1, 1, 1], 0), (1, [0, 0, 0], [0, 0, 0], [1, 1, 1], 1)], "camera": cam}


def cornell_box_ext():
    """C2's extension variant (no reference counterpart): red / green side walls, a ceiling light, a mirror
    tall box and a glossy short box.  Material ids 2.. index `materials` (settings.materials of init_path_tracer)."""
    wv = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0], [-1, -1, 2], [1, -1, 2], [1, 1, 2], [-1, 1, 2]], float)
    grey = np.asarray(_quad(0, 1, 2, 3) + _quad(4, 7, 6, 5) + _quad(3, 2, 6, 7))     # floor, ceiling, back
    left, right = np.asarray(_quad(0, 3, 7, 4)), np.asarray(_quad(1, 5, 6, 2))
    lv = np.array([[-0.35, -0.35, 1.995], [0.35, -0.35, 1.995], [0.35, 0.35, 1.995], [-0.35, 0.35, 1.995]], float)
    lf = np.asarray(_quad(0, 3, 2, 1))
    b1v, b1f = _box_faces([-0.65, -0.05, 0.0], [-0.05, 0.55, 1.2], rot_z=0.30)
    b2v, b2f = _box_faces([0.10, -0.60, 0.0], [0.70, 0.00, 0.6], rot_z=-0.28)
    meshes = [mesh_from_vertices(wv, grey), mesh_from_vertices(wv, left), mesh_from_vertices(wv, right), mesh_from_vertices(lv, lf),
              mesh_from_vertices(b1v, b1f), mesh_from_vertices(b2v, b2f)]
    ident = ([0, 0, 0], [0, 0, 0], [1, 1, 1])
    inst = [(0, *ident, 0), (1, *ident, 2), (2, *ident, 3), (3, *ident, 4), (4, *ident, 5), (5, *ident, 6)]
    materials = [{"type": "lambert", "albedo": [0.63, 0.065, 0.05]}, {"type": "lambert", "albedo": [0.14, 0.45, 0.091]},
                 {"type": "emitter", "emission": [17.0, 12.0, 4.0]}, {"type": "mirror", "albedo": [0.9, 0.9, 0.9]},
                 {"type": "glossy", "albedo": [0.8, 0.7, 0.3], "roughness": 0.35}]
    return {"meshes": meshes, "instances": inst, "camera": {"position": [0.0, -4.6, 1.0], "lookAt": [0.0, 0.0, 1.0], "fov": 20.0},
            "materials": materials}


def floor_scene():
    """C1: the reference's floor instance and camera (main.js:33,42-46)."""
    return {"meshes": [{"file": FLOOR_OBJ}], "instances": [(0, [10, 10, -1.3], [0, 0, 0], [5, 5, 5], 0)],
            "camera": {"position": [15, 15, 5], "lookAt": [10, 10, 0], "fov": 60}}


# ---- seeded value noise ------------------------------------------------------------------------
def _hash_u32(x, y, z, seed):
    h = (x.astype(np.uint64) * 0x9E3779B1 + y.astype(np.uint64) * 0x85EBCA77 + z.astype(np.uint64) * 0xC2B2AE3D + seed * 0x27D4EB2F) & 0xFFFFFFFF
    h ^= h >> 15; h = (h * 0x2C1B3C6D) & 0xFFFFFFFF
    h ^= h >> 12; h = (h * 0x297A2D39) & 0xFFFFFFFF
    h ^= h >> 15
    return h.astype(np.float64) / 4294967296.0


def value_noise3(p, seed, octaves=4):
    """p f64[n,3] -> f64[n] in [0,1): trilinear-smoothstep lattice noise, `octaves` octaves."""
    out = np.zeros(p.shape[0]); amp, tot = 1.0, 0.0
    q = p.copy()
    for o in range(octaves):
        f = np.floor(q); t = q - f; t = t * t * (3 - 2 * t)
        i = f.astype(np.int64) & 0xFFFFF
        acc = 0.0
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    w = (t[:, 0] if dx else 1 - t[:, 0]) * (t[:, 1] if dy else 1 - t[:, 1]) * (t[:, 2] if dz else 1 - t[:, 2])
                    acc = acc + w * _hash_u32(i[:, 0] + dx, i[:, 1] + dy, i[:, 2] + dz, seed + 101 * o)
        out += amp * acc; tot += amp; amp *= 0.5; q = q * 2.0
    return out / tot


def icosphere(level=6, seed=3, radius=1.0, displace=0.25):
    """C3 stand-in for the absent bunny: level-6 icosphere = 20*4^6 = 81,920 triangles with a seeded
    radial value-noise displacement."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], float)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    for _ in range(level):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        es = np.sort(e, axis=1)
        uniq, inv = np.unique(es, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = v.shape[0]; v = np.concatenate([v, mid]); n = f.shape[0]
        a, b, c = base + inv[:n], base + inv[n:2 * n], base + inv[2 * n:]
        f = np.concatenate([np.stack([f[:, 0], a, c], 1), np.stack([f[:, 1], b, a], 1), np.stack([f[:, 2], c, b], 1), np.stack([a, b, c], 1)])
    r = radius * (1.0 + displace * (value_noise3(v * 2.5 + 7.0, seed) - 0.5) * 2.0)
    return mesh_from_vertices(v * r[:, None], f)


def bunny_scale_scene(level=6, seed=3):
    """C3: displaced icosphere resting over the reference floor; camera like main.js:42-46 but closer."""
    return {"meshes": [icosphere(level, seed), {"file": FLOOR_OBJ}],
            "instances": [(0, [0, 0, 1.25], [0.3, 0.2, 0.5], [1, 1, 1], 1), (1, [0, 0, 0], [0, 0, 0], [1, 1, 1], 0)],
            "camera": {"position": [3.2, 3.2, 2.4], "lookAt": [0, 0, 1.1], "fov": 28}}


def _cached(key, make):
    """Large generated meshes are kept as .npz under $PT_SCENE_CACHE (default /tmp/pt_scene_cache) between runs."""
    import os
    d = os.environ.get("PT_SCENE_CACHE", "/tmp/pt_scene_cache")
    f = os.path.join(d, key + ".npz")
    if os.path.exists(f):
        z = np.load(f)
        return {"triangles": z["triangles"], "bounds": {"min": z["bmin"].tolist(), "max": z["bmax"].tolist()}}
    m = make()
    try:
        os.makedirs(d, exist_ok=True)
        np.savez(f + ".tmp.npz", triangles=m["triangles"], bmin=np.asarray(m["bounds"]["min"]), bmax=np.asarray(m["bounds"]["max"]))
        os.replace(f + ".tmp.npz", f)
    except OSError:
        pass
    return m


def displaced_grid(nx=2236, ny=2236, seed=4, extent=20.0, height=1.5):
    """C4: nx*ny quads = 2*nx*ny triangles (2236^2 -> 9,999,392) over [-extent/2, extent/2]^2, z = seeded value noise."""
    if nx * ny >= 250_000:
        return _cached(f"grid_{nx}_{ny}_{seed}_{extent}_{height}", lambda: _displaced_grid(nx, ny, seed, extent, height))
    return _displaced_grid(nx, ny, seed, extent, height)


def _displaced_grid(nx, ny, seed, extent, height):
    xs = np.linspace(-extent / 2, extent / 2, nx + 1); ys = np.linspace(-extent / 2, extent / 2, ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    P = np.stack([X.reshape(-1), Y.reshape(-1), np.zeros(X.size)], axis=1)
    P[:, 2] = height * value_noise3(P * np.array([0.35, 0.35, 1.0]) + 11.0, seed, octaves=5)
    idx = (np.arange(ny)[:, None] * (nx + 1) + np.arange(nx)[None, :]).reshape(-1)
    a, b, c, d = idx, idx + 1, idx + nx + 2, idx + nx + 1
    f = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)])
    return mesh_from_vertices(P, f)


def grid_scene(nx=2236, ny=2236, seed=4):
    return {"meshes": [displaced_grid(nx, ny, seed)], "instances": [(0, [0, 0, 0], [0, 0, 0], [1, 1, 1], 0)],
            "camera": {"position": [9.0, -9.0, 7.0], "lookAt": [0, 0, 0.5], "fov": 32}}


def instanced_scene(n_side=500, seed=5, grid=10):
    """C5: a 2*n_side^2-triangle base mesh (500 -> 500k) instanced grid x grid times on the
    reference's lattice (main.js:25-31) with rotations from a seeded LCG instead of Math.random,
    over the scaled floor (main.js:33)."""
    base = displaced_grid(n_side, n_side, seed, extent=1.6, height=0.9)
    inst, state = [], (seed * 2654435761) & 0xFFFFFFFF
    for i in range(grid * grid):
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        inst.append((0, [(i % grid) * 2, (i // grid) * 2, 0], [math.pi / 2, 0, state / 4294967296.0 * math.pi * 2.0], [1, 1, 1], 1))
    inst.append((1, [10, 10, -1.3], [0, 0, 0], [5, 5, 5], 0))
    return {"meshes": [base, {"file": FLOOR_OBJ}], "instances": inst,
            "camera": {"position": [15, 15, 5], "lookAt": [10, 10, 0], "fov": 60}}


def instanced_scene_ext(n_side=500, seed=5, grid=10):
    """C5 as BASELINE.json words it ("mixed diffuse/specular/glossy materials, 8-bounce"; no reference counterpart):
    instanced_scene with the instances' materials cycling through the reference's dark grey (1), an orange Lambert (2), a mirror (3)
    and two glossy metals (4, 5); `settings` carries the material table and the bounce cut (num_bounces advances by 2 per bounce,
    logic.js:85,137: 8 bounces = maxNb 16)."""
    desc = instanced_scene(n_side, seed, grid)
    cycle = [1, 2, 3, 4, 5]
    desc["instances"] = [(mi, pos, rot, scl, cycle[i % 5] if mi == 0 else mat) for i, (mi, pos, rot, scl, mat) in enumerate(desc["instances"])]
    desc["materials"] = [{"type": "lambert", "albedo": [0.8, 0.35, 0.05]}, {"type": "mirror", "albedo": [0.9, 0.9, 0.9]},
                         {"type": "glossy", "albedo": [0.95, 0.75, 0.35], "roughness": 0.15}, {"type": "glossy", "albedo": [0.6, 0.7, 0.9], "roughness": 0.5}]
    desc["settings"] = {"maxNb": 16, "materials": desc["materials"]}
    return desc


def random_soup(n_tris, seed, extent=4.0, size=0.35):
    """Unstructured triangle soup for parity tests (deep, irregular trees; many Morton ties at small extent)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, size=(n_tris, 1, 3))
    v = (c + rng.uniform(-size, size, size=(n_tris, 3, 3))).astype(np.float32).reshape(-1, 3)
    return mesh_from_vertices(v, np.arange(3 * n_tris).reshape(-1, 3))


def populate(scene, desc):
    """Feed a description through the reference-shaped scene API (registerMesh / instanceMesh)."""
    ids = [scene.register_mesh(m) for m in desc["meshes"]]
    for (mi, pos, rot, scl, mat) in desc["instances"]:
        scene.instance_mesh(ids[mi], pos, rot, scl, mat)
    return scene
