"""Host side of the path tracer, mirroring the reference's JS entry points over the C ABI.

    init_scene(device)        <- initScene(device)        scripts/scene/scene.js:1
    init_path_tracer(params)  <- initPathTracer(params)   scripts/render/path-tracer.js:1
    init_display(params)      <- initDisplay(params)      scripts/display/display.js:2

The primary host is the Node.js one under js/ (same three factories over an N-API addon); this
Python twin exists because tests and bench.py are Python.  Host-side numerics (OBJ parsing, TRS
matrices, transformed boxes) are in float64 like the JS and are rounded to f32 only where the
reference rounds them (DataView.setFloat32 / Float32Array stores, builders.js:33-58,99-107,164)."""
import ctypes as C
import math
import re

import numpy as np

from . import capi


# ----------------------------------------------------------------------------- scripts/scene/util
def parse_obj(text):
    """parse-obj.js:1-90.  Only lines starting with v / f / o matter; 1-based indices through a
    sentinel vertex; parseInt semantics ('3/1/2' -> 3); fan triangulation; 'o' resets the vertex
    list; bounds over every v line.  Returns dict(numTriangles, triangleArray f64[12n], bounds)."""
    tris = []
    mn = [1e30, 1e30, 1e30]
    mx = [-1e30, -1e30, -1e30]
    verts = [1e30, 1e30, 1e30]
    for line in text.split("\n"):
        if len(line) <= 1 or line[0] not in "vfo":
            continue
        tok = re.split(r"\s+", line)
        if tok[0] == "v":
            v = [_parse_float(tok[k]) if k < len(tok) else math.nan for k in (1, 2, 3)]
            for k in range(3):
                mn[k] = _js_min(mn[k], v[k])
                mx[k] = _js_max(mx[k], v[k])
            verts.extend(v)
        elif tok[0] == "f":
            idxs = [i for i in (_parse_int(t) for t in tok[1:]) if i is not None]
            if not idxs:
                continue

            def vert(i):
                return [verts[3 * i + k] if 0 <= 3 * i + k < len(verts) else math.nan for k in range(3)]
            r = vert(idxs[0])
            for i in range(1, len(idxs) - 1):
                a, b = vert(idxs[i]), vert(idxs[i + 1])
                tris.extend(r + [3.1415] + a + [3.1415] + b + [3.1415])
        elif tok[0] == "o":
            verts = [1e30, 1e30, 1e30]
    return {"numTriangles": len(tris) // 12, "triangleArray": np.asarray(tris, dtype=np.float64),
            "bounds": {"min": mn, "max": mx}}


_FLOAT_RE = re.compile(r"^\s*[+-]?(Infinity|\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?)")
_INT_RE = re.compile(r"^\s*[+-]?\d+")


def _parse_float(s):   # JS parseFloat: longest numeric prefix, else NaN
    m = _FLOAT_RE.match(s)
    return float(m.group(0)) if m else math.nan


def _parse_int(s):     # JS parseInt(s) (radix 10 prefix), NaN -> None
    m = _INT_RE.match(s)
    return int(m.group(0)) if m else None


def _js_min(a, b):     # Math.min propagates NaN
    return math.nan if (a != a or b != b) else min(a, b)


def _js_max(a, b):
    return math.nan if (a != a or b != b) else max(a, b)


def _mul4(a, b):       # transform-utils.js:61-80, column-major
    return [a[r] * b[4 * c] + a[4 + r] * b[4 * c + 1] + a[8 + r] * b[4 * c + 2] + a[12 + r] * b[4 * c + 3]
            for c in range(4) for r in range(4)]


def _translate(p):
    return [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, p[0], p[1], p[2], 1.0]


def _scale(s):
    return [s[0], 0.0, 0.0, 0.0, 0.0, s[1], 0.0, 0.0, 0.0, 0.0, s[2], 0.0, 0.0, 0.0, 0.0, 1.0]


def _rot_xyz(rot):     # transform-utils.js:100-137 (and :139-176 with negated angles)
    c, s = math.cos(rot[0]), math.sin(rot[0])
    rx = [1., 0., 0., 0., 0., c, s, 0., 0., -s, c, 0., 0., 0., 0., 1.]
    c, s = math.cos(rot[1]), math.sin(rot[1])
    ry = [c, 0., -s, 0., 0., 1., 0., 0., s, 0., c, 0., 0., 0., 0., 1.]
    c, s = math.cos(rot[2]), math.sin(rot[2])
    rz = [c, s, 0., 0., -s, c, 0., 0., 0., 0., 1., 0., 0., 0., 0., 1.]
    return rx, ry, rz


def compute_transform_matrices(position, rotation, scale):
    """transform-utils.js:45-59: L2W = T * (Rz*Ry*Rx) * S ; W2L = S^-1 * (Rx^-1*Ry^-1*Rz^-1) * T^-1."""
    rx, ry, rz = _rot_xyz(rotation)
    l2wr = _mul4(rz, _mul4(ry, rx))
    irx, iry, irz = _rot_xyz([-rotation[0], -rotation[1], -rotation[2]])
    w2lr = _mul4(irx, _mul4(iry, irz))
    l2w = _mul4(_translate(position), _mul4(l2wr, _scale(scale)))
    w2l = _mul4(_scale([1. / scale[0], 1. / scale[1], 1. / scale[2]]),
                _mul4(w2lr, _translate([-position[0], -position[1], -position[2]])))
    return {"localToWorld": l2w, "worldToLocal": w2l}


def get_transformed_bounding_box(mats, box):
    """transform-utils.js:11-43: box of the 8 transformed corners."""
    bmin, bmax, m = box["min"], box["max"], mats["localToWorld"]
    pts = [[bmin[0], bmin[1], bmin[2]], [bmax[0], bmin[1], bmin[2]], [bmin[0], bmax[1], bmin[2]], [bmin[0], bmin[1], bmax[2]],
           [bmax[0], bmin[1], bmax[2]], [bmax[0], bmax[1], bmin[2]], [bmin[0], bmax[1], bmax[2]], [bmax[0], bmax[1], bmax[2]]]
    mn, mx = [1e30] * 3, [-1e30] * 3
    for p in pts:
        q = [m[r] * p[0] + m[4 + r] * p[1] + m[8 + r] * p[2] + m[12 + r] for r in range(3)]
        mn = [_js_min(mn[k], q[k]) for k in range(3)]
        mx = [_js_max(mx[k], q[k]) for k in range(3)]
    return {"min": mn, "max": mx}


def mesh_arrays(mesh):
    """What builders.js:155-165 uploads for one mesh: (triangles f32[12n], bounds f32[6], n, bounds dict)."""
    if "file" in mesh:
        parsed = parse_obj(mesh["file"])
        tris = np.ascontiguousarray(parsed["triangleArray"], dtype=np.float32)   # builders.js:164 Float32Array.set
        bounds, n = parsed["bounds"], parsed["numTriangles"]
    else:
        tris = np.ascontiguousarray(mesh["triangles"], dtype=np.float32).reshape(-1)
        bounds, n = mesh["bounds"], tris.size // 12
    b6 = np.asarray(list(bounds["min"]) + list(bounds["max"]), dtype=np.float32)   # aabb-z-idx.js:116-122 setFloat32
    return tris, b6, n, bounds


def compute_instances(objects, mesh_bounds, mesh_handles):
    """scene.js:35-46: matrices and world boxes of every object, as pt_instance records (f32-rounded
    where builders.js:33-58 / :99-107 round)."""
    arr = (capi.PtInstance * max(len(objects), 1))()
    for i, o in enumerate(objects):
        t = o["transform"]
        mats = compute_transform_matrices(t["position"], t["rotation"], t["scale"])
        box = get_transformed_bounding_box(mats, mesh_bounds[o["meshID"]])
        o["transformMatrices"], o["bounds"] = mats, box
        arr[i].l2w[:] = np.asarray(mats["localToWorld"], dtype=np.float32).tolist()
        arr[i].w2l[:] = np.asarray(mats["worldToLocal"], dtype=np.float32).tolist()
        arr[i].aabb_min[:] = np.asarray(box["min"], dtype=np.float32).tolist()
        arr[i].aabb_max[:] = np.asarray(box["max"], dtype=np.float32).tolist()
        arr[i].mesh = mesh_handles[o["meshID"]]
        arr[i].material = int(o["material"])
    return arr


# ----------------------------------------------------------------------------------- device
class Device:
    """Stands in for the reference's GPUDevice (main.js:3-4): one pt_ctx on one MI355X."""

    def __init__(self, device_id=0, stream=None, sharing=None):
        self.lib = capi.load()
        h = C.c_void_p()
        if sharing is not None:      # a second context on the same GPU that borrows `sharing`'s built scene (pt_amd.h)
            rc = self.lib.pt_create_sharing_scene(sharing.ctx, C.byref(h))
            if rc < 0:
                raise capi.PtError(f"pt_create_sharing_scene failed ({rc}): {self.lib.pt_last_error(sharing.ctx).decode()}")
            self.parent = sharing    # keeps the scene's owner alive
        else:
            rc = self.lib.pt_create(int(device_id), C.byref(h))
            if rc < 0:
                raise capi.PtError(f"pt_create failed ({rc}): {self.lib.pt_last_error(None).decode()}")
        self.ctx = h
        if stream is not None:
            self.call("pt_set_stream", C.c_void_p(stream))

    def call(self, name, *args):
        return capi.check(self.lib, self.ctx, getattr(self.lib, name)(self.ctx, *args), name)

    def destroy(self):
        if self.ctx:
            rc = self.lib.pt_destroy(self.ctx)      # refused (PT_ERR_STATE) while pool contexts still borrow this one's scene
            if rc < 0:
                raise capi.PtError(f"pt_destroy failed ({rc}): {self.lib.pt_last_error(self.ctx).decode()}")
            self.ctx = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def request_device(device_id=0, stream=None):
    return Device(device_id, stream)


# -------------------------------------------------------------------------- scripts/scene/scene.js
class Scene:
    def __init__(self, device):
        self.device = device
        self.meshes = []      # {'mesh': {...}, 'handle': int, 'bounds': {...}, 'numTriangles': n}
        self.objects = []
        self.built = False

    def register_mesh(self, mesh):
        """scene.js:518-521.  mesh = {'file': objText} like the reference, or {'triangles': f32[n,12],
        'bounds': {'min','max'}} for procedurally generated meshes (skips the OBJ text round trip)."""
        tris, b6, n, bounds = mesh_arrays(mesh)
        handle = self.device.call("pt_mesh_upload", tris.ctypes.data_as(C.c_void_p), n, b6.ctypes.data_as(C.c_void_p))
        self.meshes.append({"mesh": mesh, "handle": handle, "bounds": bounds, "numTriangles": n})
        return len(self.meshes) - 1

    def instance_mesh(self, mesh_id, position, rotation, scale, material):
        """scene.js:523-529 (warns and returns on a bad id)."""
        if mesh_id >= len(self.meshes) or mesh_id < 0:
            import warnings
            warnings.warn(f"ERROR in scene::instanceMesh: mesh with ID {mesh_id} does not exist")
            return
        self.objects.append({"type": "mesh", "meshID": mesh_id, "material": material,
                             "transform": {"position": position, "rotation": rotation, "scale": scale}})

    def instances(self):
        return compute_instances(self.objects, [m["bounds"] for m in self.meshes], [m["handle"] for m in self.meshes])

    def build(self):
        """scene.js:15-146."""
        arr = self.instances()
        self.device.call("pt_scene_build", arr, len(self.objects))
        self.built = True

    # parity read-backs ------------------------------------------------------------------
    def info(self):
        inf = capi.PtSceneInfo()
        self.device.call("pt_scene_get_info", C.byref(inf))
        return inf

    def stack_bound(self):
        b = C.c_int32()
        self.device.call("pt_scene_get_stack_bound", C.byref(b))
        return b.value

    def _get(self, name, n, dtype, cols):
        out = np.empty((n, cols), dtype=dtype)
        self.device.call(name, out.ctypes.data_as(C.c_void_p), n)
        return out

    def blas_nodes(self):
        return self._get("pt_get_blas_nodes", self.info().n_tris, np.uint32, 16)

    def blas_tris(self):
        return self._get("pt_get_blas_tris", self.info().n_tris, np.float32, 12)

    def tlas_nodes(self):
        return self._get("pt_get_tlas_nodes", self.info().n_objects, np.uint32, 16)

    def objects_buffer(self):
        return self._get("pt_get_objects", self.info().n_objects, np.uint32, 32)

    def tlas_perm(self):
        return self._get("pt_get_tlas_perm", self.info().n_objects, np.int32, 1).reshape(-1)

    def mesh_perm(self, mesh_id):
        m = self.meshes[mesh_id]
        out = np.empty(m["numTriangles"], dtype=np.int32)
        self.device.call("pt_get_mesh_perm", m["handle"], out.ctypes.data_as(C.c_void_p), out.size)
        return out

    def trace_rays(self, o, d, counts=False):
        """Test hook over the extend kernel (ray-trace.js:53-69)."""
        o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        ho, ht = np.empty(n, np.int32), np.empty(n, np.int32)
        hd = np.empty(n, np.float32)
        cn = np.empty((n, 4), np.int32) if counts else None
        self.device.call("pt_trace_rays", o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), n,
                         ho.ctypes.data_as(C.c_void_p), ht.ctypes.data_as(C.c_void_p), hd.ctypes.data_as(C.c_void_p),
                         cn.ctypes.data_as(C.c_void_p) if counts else None)
        g = C.c_int64()
        self.device.call("pt_trace_last_guard_hits", C.byref(g))
        self.last_guard_hits = g.value      # rays stopped by the stack / iteration guards (0 on every sane scene)
        return (ho, ht, hd, cn) if counts else (ho, ht, hd)


def init_scene(device):
    return Scene(device)


# ------------------------------------------------------------------ scripts/render/path-tracer.js
MATERIAL_TYPES = {"lambert": capi.PT_MAT_LAMBERT, "emitter": capi.PT_MAT_EMITTER, "mirror": capi.PT_MAT_MIRROR, "glossy": capi.PT_MAT_GLOSSY}


def render_ext(settings, cls=None):
    """settings.{maxNb, seed, materials} -> pt_render_ext, or None when none of them is given (the reference has none of
    these: SURVEY 8f.1).  materials: list of {"type": "lambert"|"emitter"|"mirror"|"glossy", "albedo": [r,g,b], "emission": [r,g,b],
    "roughness": x (glossy only; travels in emission[0])} describing material ids 2, 3, ..."""
    if not any(k in settings for k in ("maxNb", "seed", "materials")):
        return None
    ext = (cls or capi.PtRenderExt)()
    ext.max_nb = int(settings.get("maxNb", 0)); ext.seed = int(settings.get("seed", 0)) & 0xFFFFFFFF
    mats = settings.get("materials", [])
    if len(mats) > capi.PT_MAX_EXT_MATERIALS:
        raise ValueError(f"at most {capi.PT_MAX_EXT_MATERIALS} extension materials")
    ext.n_materials = len(mats)
    for k, m in enumerate(mats):
        ext.materials[k].type = MATERIAL_TYPES[m["type"]]
        ext.materials[k].albedo[:] = [float(x) for x in m.get("albedo", [0, 0, 0])]
        ext.materials[k].emission[:] = [float(x) for x in m.get("emission", [0, 0, 0])]
        if m["type"] == "glossy":
            ext.materials[k].emission[0] = float(m.get("roughness", 0.0))
    return ext


class PathTracer:
    FIELDS = {"pixel_index": (np.int32, 1), "num_bounces": (np.int32, 1), "random_seed": (np.float32, 1),
              "path_throughput": (np.float32, 4), "material_throughput_pdf": (np.float32, 4), "path_o": (np.float32, 4),
              "path_d": (np.float32, 4), "hit_obj": (np.int32, 1), "hit_tri": (np.int32, 1), "flags": (np.uint32, 1),
              "samples_in_pixel": (np.int32, 1)}

    def __init__(self, params):
        self.device = params["device"]
        self.params = params
        p = capi.PtRenderParams()
        p.width, p.height = int(params["image"]["width"]), int(params["image"]["height"])
        p.cam_pos[:] = [float(x) for x in params["camera"]["position"]]
        p.cam_look_at[:] = [float(x) for x in params["camera"]["lookAt"]]
        p.fov_deg = float(params["camera"]["fov"])
        s = params["settings"]
        p.spp = int(s["samples"])
        p.num_paths = int(s.get("numPaths", 0))
        p.slot_base = int(s.get("slotBase", 0))
        p.ticket_begin = int(s.get("ticketBegin", 0))
        p.ticket_end = int(s.get("ticketEnd", -1))
        p.image_dev = s.get("imageDev", None)
        self.num_paths = p.num_paths or 1_000_000
        self.width, self.height = p.width, p.height
        self.device.call("pt_render_init", C.byref(p))
        ext = render_ext(s)
        if ext is not None:
            self.device.call("pt_render_set_extensions", C.byref(ext))

    def reset(self):
        """Render the same frame again (what calling initPathTracer a second time gives), no reallocation."""
        self.device.call("pt_render_reset")

    def step(self, want_stats=True):
        """path-tracer.js:49-123."""
        st = capi.PtStepStats()
        self.device.call("pt_render_step", C.byref(st) if want_stats else None)
        return st

    def run(self, max_steps=0, time_extend=False, count_bytes=False):
        """main.js:61-72 frame loop, headless: step until both stage-2 queues are empty."""
        rs = capi.PtRunStats()
        flags = ((capi.PT_RUN_TIME_EXTEND_ALL if time_extend == "all" else capi.PT_RUN_TIME_EXTEND) if time_extend else 0) | (capi.PT_RUN_COUNT_BYTES if count_bytes else 0)
        self.device.call("pt_render_run", int(max_steps), flags, C.byref(rs))
        return rs

    def destroy(self):
        """Nothing to release (the context owns the buffers); the pooled tracer releases its borrowing contexts here."""

    def get_image_buffer(self):
        """path-tracer.js:125-127: the device image buffer (address)."""
        p = C.c_void_p()
        self.device.call("pt_image_dev_ptr", C.byref(p))
        return p.value

    def read_image(self):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self.device.call("pt_image_read", out.ctypes.data_as(C.c_void_p), out.size)
        return out

    def field(self, name):
        if name == "pixel_counter":
            out = np.empty(1, np.int64)
        else:
            dt, cols = self.FIELDS[name]
            out = np.empty((self.num_paths, cols) if cols > 1 else self.num_paths, dtype=dt)
        self.device.call("pt_path_state_read", name.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes)
        return out

    def queue(self, which):
        out = np.empty(self.num_paths, np.int32)
        n = C.c_int64(out.size)
        self.device.call("pt_queue_read", int(which), out.ctypes.data_as(C.c_void_p), C.byref(n))
        return out[: n.value]


class PooledPathTracer:
    """settings.pools = K > 1 (no reference counterpart): K independent path pools of settings.numPaths slots each on one
    GPU -- SURVEY 8e's sample split applied inside the device.  Pool k is a context that borrows the scene
    (pt_create_sharing_scene), has its own stream, RNG streams (slotBase + k * numPaths) and an equal part of this GPU's
    pixel tickets, and adds into the one image; run() drives them together (pt_render_run_many), so that one pool's
    compaction launches run in the tail of another's extend launch.  Every pool is bit-identical to the oracle's run of
    the same share (`pools[k]` is an ordinary PathTracer: step(), field(), queue())."""

    def __init__(self, params):
        s = dict(params["settings"])
        k_pools = int(s.pop("pools"))
        w, h = int(params["image"]["width"]), int(params["image"]["height"])
        num_paths = int(s.get("numPaths", 0)) or 1_000_000
        total = (w * h * int(s["samples"]) + 7) // 8                           # logic.js:177
        t0 = int(s.get("ticketBegin", 0)); t1 = int(s.get("ticketEnd", -1))
        if t1 < 0:
            t0, t1 = 0, total
        base = int(s.get("slotBase", 0))
        self.device, self.width, self.height = params["device"], w, h
        self.pools = []
        image = s.get("imageDev", None)
        for k in range(k_pools):
            dev = params["device"] if k == 0 else Device(sharing=params["device"])
            sk = dict(s, numPaths=num_paths, slotBase=base + k * num_paths, imageDev=image,
                      ticketBegin=t0 + (t1 - t0) * k // k_pools, ticketEnd=t0 + (t1 - t0) * (k + 1) // k_pools)
            pt = PathTracer(dict(params, device=dev, settings=sk))
            if image is None:
                image = pt.get_image_buffer()                                   # pool 0 owns the image, the others add into it
            self.pools.append(pt)

    def reset(self):
        for pt in self.pools:
            pt.reset()

    def run(self, max_steps=0, time_extend=False, count_bytes=False):
        n = len(self.pools)
        ctxs = (C.c_void_p * n)(*[pt.device.ctx for pt in self.pools])
        rs = (capi.PtRunStats * n)()
        flags = ((capi.PT_RUN_TIME_EXTEND_ALL if time_extend == "all" else capi.PT_RUN_TIME_EXTEND) if time_extend else 0) | (capi.PT_RUN_COUNT_BYTES if count_bytes else 0)
        lib = self.device.lib
        capi.check(lib, self.device.ctx, lib.pt_render_run_many(ctxs, n, int(max_steps), flags, rs), "pt_render_run_many")
        out = capi.PtRunStats()
        out.steps = max(r.steps for r in rs)
        for f in ("total_rays", "total_samples", "stack_overflows", "extend_launches", "traversal_bytes", "wasted_rays",
                  "checked_rays", "mismatched_rays", "flipped_rays", "beyond_tolerance_rays", "lost_hit_rays"):
            setattr(out, f, sum(getattr(r, f) for r in rs))
        out.max_rel_dist_diff = max(r.max_rel_dist_diff for r in rs)
        out.extend_ms = sum(r.extend_ms for r in rs)
        self.last_pool_stats = list(rs)
        return out

    def get_image_buffer(self):
        return self.pools[0].get_image_buffer()

    def read_image(self):
        img = None
        for pt in reversed(self.pools):          # each read first adds that pool's unfinished tickets; pool 0's read sees them all
            img = pt.read_image()
        return img

    def destroy(self):
        for pt in self.pools[1:]:
            pt.device.destroy()


def hw_queues():
    """Hardware queues HIP gives this process (GPU_MAX_HW_QUEUES, HIP's default 4).  Streams beyond that number share a queue and their
    kernels serialise, so more than three pools only pay with more queues.  The package asks for 8 when the variable is unset
    (path-tracer_amd/__init__.py; it has to be in the environment before the HIP runtime starts)."""
    import os
    try:
        return max(1, int(os.environ.get("GPU_MAX_HW_QUEUES", "4")))
    except ValueError:
        return 4


def default_pool_shape(width, height):
    """(pools, numPaths) the hosts use when the caller fixes neither: three pools of a quarter of the frame's pixels each, between 750,000
    slots (the optimum at 1080p) and 2,000,000 (4K); at 4K, with at least 8 hardware queues, four pools of 1,500,000 (+5...6 % on the 4K
    configs; with HIP's default of 4 queues a fourth pool shares a queue and loses 6 %); profiles/r01_pool_size_sweep_v3.json,
    profiles/r03_pool_shape_and_weight_sweeps.txt.  The reference has one pool of 1,000,000 (path-tracer.js:4); the shape only changes
    which random stream renders which pixel ticket and when launches run (each pool is bit-identical to the oracle's run of its share)."""
    paths = int(min(2_000_000, max(750_000, width * height // 4)))
    if paths >= 1_500_000 and hw_queues() >= 8:
        return 4, paths * 3 // 4
    return 3, paths


def init_path_tracer(params):
    """initPathTracer (path-tracer.js:1).  A caller that fixes neither settings.numPaths nor settings.pools gets default_pool_shape for
    the frame (the reference's own shape -- one pool of NUM_PATHS = 1,000,000, path-tracer.js:4 -- costs 38 % of the throughput on
    MI355X: bench.py `reference_pool_shape`); {"numPaths": 1000000, "pools": 1} asks for the reference's."""
    s = params["settings"]
    if "numPaths" not in s and "pools" not in s:
        pools, paths = default_pool_shape(int(params["image"]["width"]), int(params["image"]["height"]))
        params = dict(params, settings=dict(s, numPaths=paths, pools=pools))
    if int(params["settings"].get("pools", 1)) > 1:
        return PooledPathTracer(params)
    return PathTracer(params)


# ----------------------------------------------------------------- scripts/display/display.js
class Display:
    """initDisplay({device, canvas, image:{buffer,width,height}}) -> {draw}.  The canvas becomes an
    RGBA8 array (and optionally a PNG file): display.js:55-72,98-113."""

    def __init__(self, params):
        self.device = params["device"]
        self.image = params["image"]
        self.canvas = params.get("canvas")      # None or a file path ending in .png / .ppm

    def draw(self):
        w, h = int(self.image["width"]), int(self.image["height"])
        out = np.empty((h, w, 4), dtype=np.uint8)
        self.device.call("pt_tonemap_rgba8", C.c_void_p(self.image.get("buffer")), w, h, out.ctypes.data_as(C.c_void_p))
        if self.canvas:
            write_image(self.canvas, out)
        return out


def init_display(params):
    return Display(params)


class DebugView:
    """initDebug-style visualiser of the traversal (debug/rt-debug.js:69-170): one primary ray per pixel
    through the extend kernel (pt_trace_rays), coloured by the world-space normal of the closest hit (mode "normal",
    rt-debug.js:150-156) or by the hit indices (mode "index", the commented line :157); a miss is black.  The camera is
    rt-debug.js's: pixel corner, no jitter, sin(fov/2) with fov 80 unless given."""

    def __init__(self, params):
        self.device, self.scene = params["device"], params["scene"]
        self.image, self.canvas = params["image"], params.get("canvas")
        self.camera = params["camera"]

    def rays(self):
        w, h = int(self.image["width"]), int(self.image["height"])
        f32 = np.float32
        pos, look = np.asarray(self.camera["position"], f32), np.asarray(self.camera["lookAt"], f32)
        sinfov = f32(math.sin(0.5 * float(self.camera.get("fov", 80.0)) * math.pi / 180.0))
        fwd = look - pos; fwd = fwd / np.sqrt((fwd * fwd).sum(dtype=f32))
        right = np.asarray([fwd[1], -fwd[0], 0.0], f32); right = right / np.sqrt((right * right).sum(dtype=f32))
        up = np.cross(right, fwd).astype(f32)
        x, y = np.meshgrid(np.arange(w, dtype=f32), np.arange(h, dtype=f32))
        sx = x / f32(w) * f32(2) - f32(1); sy = -(y / f32(h) * f32(2) - f32(1))
        local = np.stack([f32(w / h) * sx * sinfov, np.ones_like(sx), sy * sinfov], -1)
        local = local / np.sqrt((local * local).sum(-1, keepdims=True, dtype=f32))
        d = local[..., 0:1] * right + local[..., 1:2] * fwd + local[..., 2:3] * up
        o = np.broadcast_to(pos, d.shape)
        return np.ascontiguousarray(o.reshape(-1, 3), f32), np.ascontiguousarray(d.reshape(-1, 3), f32)

    def draw(self, mode="normal"):
        w, h = int(self.image["width"]), int(self.image["height"])
        o, d = self.rays()
        hit_obj, hit_tri, hit_dist = self.scene.trace_rays(o, d)
        hit = (hit_obj >= 0) & (hit_dist <= 1e5)
        col = np.zeros((w * h, 3), np.float32)
        if mode == "index":
            col[hit, 0] = hit_obj[hit] / np.float32(20.0); col[hit, 1] = hit_tri[hit] / np.float32(5000.0)
        else:
            tris = self.scene.blas_tris()[hit_tri[hit]].reshape(-1, 3, 4)[..., :3]
            ln = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
            ln /= np.linalg.norm(ln, axis=1, keepdims=True)
            l2w = self.scene.objects_buffer().view(np.float32).reshape(-1, 32)[hit_obj[hit], :12].reshape(-1, 3, 4)[..., :3]
            n = np.einsum("nij,nj->ni", l2w, ln)
            col[hit] = n / np.linalg.norm(n, axis=1, keepdims=True)
        out = np.full((h, w, 4), 255, np.uint8)
        out[..., :3] = (np.clip(col, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8).reshape(h, w, 3)
        if self.canvas:
            write_image(self.canvas, out)
        return out


def init_debug_view(params):
    return DebugView(params)


def write_image(path, rgba8):
    """PNG (zlib, no dependencies) or binary PPM by extension."""
    import struct
    import zlib
    h, w, _ = rgba8.shape
    if path.endswith(".ppm"):
        with open(path, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (w, h))
            f.write(np.ascontiguousarray(rgba8[:, :, :3]).tobytes())
        return

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    raw = b"".join(b"\x00" + np.ascontiguousarray(rgba8[y]).tobytes() for y in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
