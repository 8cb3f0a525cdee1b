// pt-host.js -- headless Node.js host of the MI355X path tracer.
//
// Keeps the reference's three entry points and call sequence (main.js:16-72):
//     initScene(device)       -> { registerMesh, instanceMesh, build, kernels }        scene.js:1,13
//     initPathTracer(params)  -> { step, getImageBuffer, run, readImage }               path-tracer.js:1,47
//     initDisplay(params)     -> { draw }                                               display.js:2,53
// The WebGPU device becomes a pt context on one MI355X (requestDevice), every WGSL dispatch becomes a
// call into libpt_amd.so through the N-API addon.  Host-side numerics stay in JS doubles and are rounded to
// f32 exactly where the reference rounds them (Float32Array stores / DataView.setFloat32).
"use strict"
const fs = require("fs")
const zlib = require("zlib")
const path = require("path")

let native = null
function addon() {
    if (!native) native = require(path.join(__dirname, "pt_amd.node"))   // throws if the addon was not built
    return native
}

// ---- requestAdapter()/requestDevice() (main.js:3-4) ------------------------------------------------------------
// options.traversal (no reference counterpart; pt_amd.h "tolerance mode"): "reference" (default, the bit-exact walk of scene.js:313-475),
// "tolerance" (the 4-wide collapse north_star's allowance admits; the differing rays are counted in run()'s stats) or "auto" (tolerance
// where the scene has at least 2^18 packed triangles per object); options.checkPeriod: one 64-ray queue block in N is walked a second time exactly (default 256).
const TRAVERSAL = { reference: 0, tolerance: 1, auto: 2 }
function requestDevice(deviceId = 0, options = {}) {
    const ctx = addon().create(deviceId)
    if (options.traversal !== undefined || options.checkPeriod !== undefined) {
        if (options.traversal !== undefined && !(options.traversal in TRAVERSAL)) throw new Error(`requestDevice: unknown traversal ${options.traversal}`)
        addon().setTraversal(ctx, TRAVERSAL[options.traversal || "reference"], options.checkPeriod === undefined ? -1 : options.checkPeriod)
    }
    return { ctx, destroy() { addon().destroy(ctx) } }
}

// ---- OBJ text -> padded triangle list + bounds (semantics of parse-obj.js:1-90) ---------------------------------
// v / f / o records only; indices are 1-based through a sentinel vertex and read like parseInt ("7/2/3" -> 7);
// polygons are fanned from their first corner; an "o" record starts a fresh vertex list; the bounds cover every
// v record whether or not a face uses it.  Each corner is padded to four floats with 3.1415.
function parseObj(text) {
    const PAD = 3.1415
    const lo = [1e30, 1e30, 1e30], hi = [-1e30, -1e30, -1e30]
    const out = []
    let pool = [1e30, 1e30, 1e30]
    for (const line of text.split("\n")) {
        if (line.length <= 1 || !"vfo".includes(line[0])) continue
        const tok = line.split(/\s+/)
        if (tok[0] === "v") {
            for (let k = 0; k < 3; k++) {
                const c = parseFloat(tok[k + 1])
                lo[k] = Math.min(lo[k], c); hi[k] = Math.max(hi[k], c)
                pool.push(c)
            }
        } else if (tok[0] === "f") {
            const corners = tok.slice(1).map(t => parseInt(t)).filter(i => !isNaN(i))
            const at = i => [pool[3 * i], pool[3 * i + 1], pool[3 * i + 2], PAD]
            for (let k = 1; k + 1 < corners.length; k++) out.push(...at(corners[0]), ...at(corners[k]), ...at(corners[k + 1]))
        } else if (tok[0] === "o") {
            pool = [1e30, 1e30, 1e30]
        }
    }
    return { numTriangles: out.length / 12, triangleArray: out, bounds: { min: lo, max: hi } }
}

// ---- TRS matrices, column-major 4x4 (semantics of transform-utils.js:45-176) ------------------------------------
function matMul(a, b) {             // element (r,c) = sum_k a(r,k) b(k,c), accumulated left to right
    const m = new Array(16)
    for (let c = 0; c < 4; c++)
        for (let r = 0; r < 4; r++)
            m[4 * c + r] = a[r] * b[4 * c] + a[4 + r] * b[4 * c + 1] + a[8 + r] * b[4 * c + 2] + a[12 + r] * b[4 * c + 3]
    return m
}
const translation = p => [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, p[0], p[1], p[2], 1]
const scaling = s => [s[0], 0, 0, 0, 0, s[1], 0, 0, 0, 0, s[2], 0, 0, 0, 0, 1]
function axisRotations(a) {
    const [cx, sx, cy, sy, cz, sz] = [Math.cos(a[0]), Math.sin(a[0]), Math.cos(a[1]), Math.sin(a[1]), Math.cos(a[2]), Math.sin(a[2])]
    return {
        x: [1, 0, 0, 0, 0, cx, sx, 0, 0, -sx, cx, 0, 0, 0, 0, 1],
        y: [cy, 0, -sy, 0, 0, 1, 0, 0, sy, 0, cy, 0, 0, 0, 0, 1],
        z: [cz, sz, 0, 0, -sz, cz, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1],
    }
}
function computeTransformMatrices(t) {
    const f = axisRotations(t.rotation), b = axisRotations(t.rotation.map(x => -x))
    const rot = matMul(f.z, matMul(f.y, f.x)), unrot = matMul(b.x, matMul(b.y, b.z))
    return {
        localToWorld: matMul(translation(t.position), matMul(rot, scaling(t.scale))),
        worldToLocal: matMul(scaling(t.scale.map(s => 1. / s)), matMul(unrot, translation(t.position.map(x => -x)))),
    }
}
function getTransformedBoundingBox(mats, box) {   // box of the 8 mapped corners, in the reference's corner order
    const m = mats.localToWorld, a = box.min, b = box.max
    const corners = [[a[0], a[1], a[2]], [b[0], a[1], a[2]], [a[0], b[1], a[2]], [a[0], a[1], b[2]],
                     [b[0], a[1], b[2]], [b[0], b[1], a[2]], [a[0], b[1], b[2]], [b[0], b[1], b[2]]]
    const min = [1e30, 1e30, 1e30], max = [-1e30, -1e30, -1e30]
    for (const p of corners)
        for (let r = 0; r < 3; r++) {
            const q = m[r] * p[0] + m[4 + r] * p[1] + m[8 + r] * p[2] + m[12 + r]
            min[r] = Math.min(min[r], q); max[r] = Math.max(max[r], q)
        }
    return { min, max }
}

// ---- initScene (scene.js) ---------------------------------------------------------------------------------------
function initScene(device) {
    const meshes = [], objects = []
    let built = false

    function registerMesh(mesh) {          // scene.js:518-521; {file: objText} or {triangles, bounds} (generated meshes)
        const parsed = mesh.file !== undefined ? parseObj(mesh.file)
            : { numTriangles: mesh.triangles.length / 12, triangleArray: mesh.triangles, bounds: mesh.bounds }
        const tris = Float32Array.from(parsed.triangleArray)                        // builders.js:164
        const b6 = Float32Array.from([...parsed.bounds.min, ...parsed.bounds.max])  // aabb-z-idx.js:116-122
        const handle = addon().meshUpload(device.ctx, tris, b6)
        meshes.push({ mesh, handle, bounds: parsed.bounds, numTriangles: parsed.numTriangles })
        return meshes.length - 1
    }
    function instanceMesh(meshID, position, rotation, scale, material) {            // scene.js:523-529
        if (meshID >= meshes.length || meshID < 0) {
            console.warn(`ERROR in scene::instanceMesh: mesh with ID ${meshID} does not exist`)
            return
        }
        objects.push({ type: "mesh", meshID, material, transform: { position, rotation, scale } })
    }
    async function build() {                                                        // scene.js:15-146
        const words = new ArrayBuffer(160 * Math.max(objects.length, 1))
        const f = new Float32Array(words), i = new Int32Array(words)
        objects.forEach((o, k) => {
            o.transformMatrices = computeTransformMatrices(o.transform)
            o.bounds = getTransformedBoundingBox(o.transformMatrices, meshes[o.meshID].bounds)
            f.set(o.transformMatrices.localToWorld, 40 * k); f.set(o.transformMatrices.worldToLocal, 40 * k + 16)
            f.set(o.bounds.min, 40 * k + 32); f.set(o.bounds.max, 40 * k + 35)
            i[40 * k + 38] = meshes[o.meshID].handle; i[40 * k + 39] = o.material
        })
        addon().sceneBuild(device.ctx, f, objects.length)
        built = true
    }
    const notBuilt = () => { if (!built) console.warn("ERROR in scene::getSceneBindGroupInfo: scene has not been built yet") }
    const kernels = {   // the scene -> render hand-off (scene.js:13): device buffers instead of WGSL strings + a bind group
        getSceneBindGroupInfo() { notBuilt(); return built ? addon().sceneInfo(device.ctx) : null },
        readBuffer(which, out, mesh) { return mesh === undefined ? addon().readBuffer(device.ctx, which, out) : addon().readBuffer(device.ctx, which, out, meshes[mesh].handle) },
        traceRays(o, d, hitObj, hitTri, hitDist) { addon().traceRays(device.ctx, o, d, hitObj, hitTri, hitDist || null) },
    }
    return { registerMesh, instanceMesh, build, kernels }
}

// ---- initPathTracer (path-tracer.js) ----------------------------------------------------------------------------
// A caller that fixes neither settings.numPaths nor settings.pools gets the measured-best shape for the frame: three pools of a quarter of
// its pixels each, between 750,000 slots (1080p) and 2,000,000 (4K).  The reference's own shape, one pool of NUM_PATHS = 1,000,000
// (path-tracer.js:4), costs 38 % of the throughput on MI355X; { numPaths: 1000000, pools: 1 } asks for it.  The shape only changes which
// random stream renders which pixel ticket and when launches run.
// At 4K, with at least 8 hardware queues (GPU_MAX_HW_QUEUES; set to 8 below unless the caller has set it -- it is read when the HIP
// runtime starts, i.e. at the first requestDevice), four pools of 1,500,000: streams beyond the queue count share a queue and serialise.
if (process.env.GPU_MAX_HW_QUEUES === undefined) process.env.GPU_MAX_HW_QUEUES = "8"
function defaultPoolShape(width, height) {
    const numPaths = Math.min(2000000, Math.max(750000, Math.floor(width * height / 4)))
    const queues = Math.max(1, parseInt(process.env.GPU_MAX_HW_QUEUES || "4", 10) || 4)
    return (numPaths >= 1500000 && queues >= 8) ? { pools: 4, numPaths: Math.floor(numPaths * 3 / 4) } : { pools: 3, numPaths }
}
function initPathTracer(params) {
    let s = params.settings || {}
    if (s.numPaths === undefined && s.pools === undefined) { s = Object.assign({}, s, defaultPoolShape(params.image.width, params.image.height)); params = Object.assign({}, params, { settings: s }) }
    const device = params.device
    if ((s.pools || 1) > 1) return initPooledPathTracer(params)
    addon().renderInit(device.ctx, {
        width: params.image.width, height: params.image.height,
        position: params.camera.position, lookAt: params.camera.lookAt, fov: params.camera.fov,
        samples: s.samples, numPaths: s.numPaths || 0, slotBase: s.slotBase || 0,
        ticketBegin: s.ticketBegin || 0, ticketEnd: s.ticketEnd === undefined ? -1 : s.ticketEnd,
        imageOf: s.imageOf,
    })
    // extensions the reference does not have (default-off): settings.maxNb, settings.seed, settings.materials =
    // [{type: "lambert"|"emitter"|"mirror"|"glossy", albedo: [r,g,b], emission: [r,g,b], roughness: x (glossy)}, ...] for material ids 2, 3, ...
    if (s.maxNb !== undefined || s.seed !== undefined || s.materials !== undefined) {
        const types = { lambert: 0, emitter: 1, mirror: 2, glossy: 3 }, mats = s.materials || [], f = new Float32Array(8 * mats.length)
        mats.forEach((m, k) => {
            if (!(m.type in types)) throw new Error(`unknown material type ${m.type}`)
            f.set(m.albedo || [0, 0, 0], 8 * k); f[8 * k + 3] = types[m.type]; f.set(m.emission || [0, 0, 0], 8 * k + 4)
            if (m.type === "glossy") f[8 * k + 4] = m.roughness || 0
        })
        addon().renderSetExtensions(device.ctx, s.maxNb || 0, s.seed || 0, f)
    }
    async function step() { return addon().renderStep(device.ctx) }                 // path-tracer.js:49-123
    function getImageBuffer() { return { device, width: params.image.width, height: params.image.height } }   // :125-127
    async function run(maxSteps = 0, flags = 0) { return addon().renderRun(device.ctx, maxSteps, flags) }     // main.js:61-72 loop
    function readImage() {
        const out = new Float32Array(params.image.width * params.image.height * 4)
        addon().imageRead(device.ctx, out)
        return out
    }
    return { step, getImageBuffer, run, readImage, reset: () => addon().renderReset(device.ctx), destroy() {} }
}

// settings.pools = K > 1 (no reference counterpart): K path pools of settings.numPaths slots on one GPU -- contexts that
// borrow the scene, each with its own stream, RNG streams (slotBase + k * numPaths) and an equal part of the pixel tickets,
// all adding into pool 0's image, stepped together by renderRunMany (pt_amd.h "several path pools on one GPU").
function initPooledPathTracer(params) {
    const s = params.settings, K = s.pools, W = params.image.width, H = params.image.height
    const numPaths = s.numPaths || 1000000, total = Math.ceil(W * H * s.samples / 8)          // logic.js:177
    let t0 = s.ticketBegin || 0, t1 = s.ticketEnd === undefined ? -1 : s.ticketEnd
    if (t1 < 0) { t0 = 0; t1 = total }
    const pools = []
    for (let k = 0; k < K; k++) {
        const dev = k === 0 ? params.device : { ctx: addon().createSharingScene(params.device.ctx) }
        const sk = Object.assign({}, s, { pools: 1, numPaths, slotBase: (s.slotBase || 0) + k * numPaths,
            ticketBegin: t0 + Math.floor((t1 - t0) * k / K), ticketEnd: t0 + Math.floor((t1 - t0) * (k + 1) / K),
            imageOf: k === 0 ? undefined : params.device.ctx })
        pools.push(Object.assign(initPathTracer(Object.assign({}, params, { device: dev, settings: sk })), { device: dev }))
    }
    async function run(maxSteps = 0, flags = 0) {
        const rs = addon().renderRunMany(pools.map(p => p.device.ctx), maxSteps, flags)
        const sum = f => rs.reduce((a, r) => a + r[f], 0)
        return { steps: Math.max(...rs.map(r => r.steps)), totalRays: sum("totalRays"), totalSamples: sum("totalSamples"),
                 stackOverflows: sum("stackOverflows"), extendMs: sum("extendMs"), extendLaunches: sum("extendLaunches"),
                 traversalBytes: sum("traversalBytes"), wastedRays: sum("wastedRays"), checkedRays: sum("checkedRays"), mismatchedRays: sum("mismatchedRays"),
                 flippedRays: sum("flippedRays"), beyondToleranceRays: sum("beyondToleranceRays"), lostHitRays: sum("lostHitRays"),
                 maxRelDistDiff: Math.max(...rs.map(r => r.maxRelDistDiff)), pools: rs }
    }
    function readImage() { let img; for (let k = K - 1; k >= 0; k--) img = pools[k].readImage(); return img }   // each read adds that pool's open tickets
    async function step() {                                                       // every pool takes one step; the sizes add up
        const r = []; for (const p of pools) r.push(await p.step())
        const sum = f => r.reduce((a, x) => a + x[f], 0)
        return { camera: sum("camera"), material: sum("material"), rays: sum("rays"), step: r[0].step, pools: r }
    }
    function destroy() { for (let k = K - 1; k > 0; k--) addon().destroy(pools[k].device.ctx) }   // the borrowers; pool 0 is the caller's device
    return { step, run, readImage, pools, destroy, getImageBuffer: pools[0].getImageBuffer, reset: () => pools.forEach(p => p.reset()) }
}

// ---- several GPUs of one node (SURVEY 8e; no reference counterpart) ---------------------------------------------------
// The samples are split: GPU g of G gets the whole scene (build it once per device), the g-th part of the pixel tickets and
// its own RNG streams -- shareOf(g, G, image, settings) gives the settings for its initPathTracer -- and renders into its own
// accumulation buffer.  initMultiGpu(devices).reduceImage(root) then MOVES the buffers onto devices[root]'s with one RCCL
// sum-reduce over xGMI (pt_comm_reduce_image: the other GPUs' images are empty afterwards, so a later reduce counts nothing twice);
// that image is the single-GPU image in form (sums of radiance and sample counts).
function shareOf(g, G, image, settings) {
    const total = Math.ceil(image.width * image.height * settings.samples / 8)                   // logic.js:177
    const pools = settings.pools || 1, numPaths = settings.numPaths || 1000000
    return Object.assign({}, settings, { numPaths, slotBase: g * pools * numPaths,
        ticketBegin: Math.floor(total * g / G), ticketEnd: Math.floor(total * (g + 1) / G) })
}
function initMultiGpu(devices) {
    const comm = addon().commCreate(devices.map(d => d.ctx))                                      // rank i = devices[i]
    return { reduceImage(root = 0) { addon().commReduceImage(comm, root) }, destroy() { addon().commDestroy(comm) } }
}

// ---- initDisplay (display.js): the canvas is a PNG (or raw RGBA8) ------------------------------------------------
function initDisplay(params) {
    const { width, height } = params.image
    function draw() {                                                               // display.js:55-72, 98-113
        const rgba = new Uint8Array(width * height * 4)
        addon().tonemap(params.device.ctx, rgba)
        if (params.canvas) fs.writeFileSync(params.canvas, encodePNG(rgba, width, height))
        return rgba
    }
    return { draw }
}

// ---- debug view (debug/rt-debug.js:69-170; SURVEY 8f.2) ----------------------------------------------------------------
// One primary ray per pixel through the extend kernel, coloured by the world normal of the closest hit ("normal") or by the
// hit indices ("index", the commented line rt-debug.js:157); a miss is black.  rt-debug.js's camera: pixel corner, no
// jitter, sin(fov / 2), fov 80 unless given.
function initDebugView(params) {
    const { width: w, height: h } = params.image, cam = params.camera, f = Math.fround
    const sub = (a, b) => a.map((x, i) => f(x - b[i])), norm = v => { const l = f(Math.sqrt(f(f(f(v[0] * v[0]) + f(v[1] * v[1])) + f(v[2] * v[2])))); return v.map(x => f(x / l)) }
    const cross = (a, b) => [f(f(a[1] * b[2]) - f(a[2] * b[1])), f(f(a[2] * b[0]) - f(a[0] * b[2])), f(f(a[0] * b[1]) - f(a[1] * b[0]))]
    function draw(mode = "normal") {
        const pos = cam.position.map(f), fwd = norm(sub(cam.lookAt.map(f), pos)), right = norm([fwd[1], -fwd[0], 0]), up = cross(right, fwd)
        const sinfov = f(Math.sin(0.5 * (cam.fov === undefined ? 80 : cam.fov) * Math.PI / 180)), aspect = f(w / h)
        const o = new Float32Array(w * h * 3), d = new Float32Array(w * h * 3)
        for (let y = 0; y < h; y++) for (let x = 0; x < w; x++) {
            const sx = f(f(f(x / w) * 2) - 1), sy = -f(f(f(y / h) * 2) - 1)
            const l = norm([f(f(aspect * sx) * sinfov), 1, f(sy * sinfov)]), i = 3 * (y * w + x)
            for (let k = 0; k < 3; k++) { o[i + k] = pos[k]; d[i + k] = f(f(f(l[0] * right[k]) + f(l[1] * fwd[k])) + f(l[2] * up[k])) }
        }
        const hitObj = new Int32Array(w * h), hitTri = new Int32Array(w * h), hitDist = new Float32Array(w * h)
        addon().traceRays(params.device.ctx, o, d, hitObj, hitTri, hitDist)
        const info = params.scene.kernels.getSceneBindGroupInfo()
        const tris = new Float32Array(12 * info.nTris), objs = new Float32Array(32 * info.nObjects)
        params.scene.kernels.readBuffer("blasTris", tris); params.scene.kernels.readBuffer("objects", objs)
        const rgba = new Uint8Array(w * h * 4)
        for (let p = 0; p < w * h; p++) {
            let col = [0, 0, 0]
            if (hitObj[p] >= 0 && hitDist[p] <= 1e5) {
                if (mode === "index") col = [hitObj[p] / 20, hitTri[p] / 5000, 0]
                else {
                    const t = 12 * hitTri[p], v = k => [tris[t + 4 * k], tris[t + 4 * k + 1], tris[t + 4 * k + 2]]
                    const ln = norm(cross(sub(v(1), v(0)), sub(v(2), v(0)))), m = 32 * hitObj[p]
                    col = norm([0, 1, 2].map(r => objs[m + 4 * r] * ln[0] + objs[m + 4 * r + 1] * ln[1] + objs[m + 4 * r + 2] * ln[2]))
                }
            }
            for (let k = 0; k < 3; k++) rgba[4 * p + k] = Math.floor(Math.min(Math.max(col[k], 0), 1) * 255 + 0.5)
            rgba[4 * p + 3] = 255
        }
        if (params.canvas) fs.writeFileSync(params.canvas, encodePNG(rgba, w, h))
        return rgba
    }
    return { draw }
}

function encodePNG(rgba, w, h) {
    const crcTable = new Int32Array(256).map((_, n) => { let c = n; for (let k = 0; k < 8; k++) c = c & 1 ? 0xedb88320 ^ (c >>> 1) : c >>> 1; return c })
    const crc = buf => { let c = -1; for (const b of buf) c = crcTable[(c ^ b) & 255] ^ (c >>> 8); return (c ^ -1) >>> 0 }
    const chunk = (tag, data) => {
        const body = Buffer.concat([Buffer.from(tag, "ascii"), data]), len = Buffer.alloc(4), sum = Buffer.alloc(4)
        len.writeUInt32BE(data.length); sum.writeUInt32BE(crc(body))
        return Buffer.concat([len, body, sum])
    }
    const raw = Buffer.alloc((w * 4 + 1) * h)
    for (let y = 0; y < h; y++) Buffer.from(rgba.buffer, rgba.byteOffset + y * w * 4, w * 4).copy(raw, y * (w * 4 + 1) + 1)
    const ihdr = Buffer.alloc(13); ihdr.writeUInt32BE(w, 0); ihdr.writeUInt32BE(h, 4); ihdr.set([8, 6, 0, 0, 0], 8)
    return Buffer.concat([Buffer.from([137, 80, 78, 71, 13, 10, 26, 10]), chunk("IHDR", ihdr), chunk("IDAT", zlib.deflateSync(raw)), chunk("IEND", Buffer.alloc(0))])
}

module.exports = { requestDevice, initScene, initPathTracer, defaultPoolShape, initDisplay, initDebugView, initMultiGpu, shareOf, parseObj, computeTransformMatrices, getTransformedBoundingBox, encodePNG }
