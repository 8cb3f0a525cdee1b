#!/usr/bin/env node
// main.js -- headless counterpart of the reference's scripts/main.js: same call sequence
// (registerMesh x k -> instanceMesh x m -> await build() -> initPathTracer -> initDisplay -> loop { await step(); draw() }),
// no browser.  The reference's demo meshes (cow / bunny / buddha) are not distributed with it, so the demo scene is
// the floor it does bundle under a lattice of generated boxes standing in for the 100 instanced meshes (main.js:25-33).
//   node main.js [--width W] [--height H] [--samples S] [--paths N] [--pools K] [--gpus G] [--grid G] [--out frame.png] [--progress K] [--json] [--ext] [--traversal reference|tolerance|auto]
// --gpus G splits the samples over GPUs 0..G-1 (each builds the scene and renders its part of the pixel tickets with its own RNG
// streams); the accumulation buffers are summed onto GPU 0 by one RCCL reduce before the final draw (pt-host.js initMultiGpu).
// --progress K redraws (rewrites the PNG) every K steps, the headless form of the reference's draw-every-frame loop.
// --ext switches the extensions on (not in the reference): box i gets material 1 + i % 4, i.e. the reference's dark grey,
// an orange Lambert, a mirror and a warm emitter; bounce cut at nb > 8; seed 3.
"use strict"
const pt = require("./pt-host.js")

const FLOOR = "v -10.0 -10.0 0.0\nv  10.0 -10.0 0.0\nv  10.0  10.0 0.0\nv -10.0  10.0 0.0\n\nf 1 2 3\nf 1 3 4"
const BOX = ["v -0.5 -0.5 0", "v 0.5 -0.5 0", "v 0.5 0.5 0", "v -0.5 0.5 0", "v -0.5 -0.5 1.4", "v 0.5 -0.5 1.4", "v 0.5 0.5 1.4", "v -0.5 0.5 1.4",
             "f 1 4 3 2", "f 5 6 7 8", "f 1 2 6 5", "f 3 4 8 7", "f 1 5 8 4", "f 2 3 7 6"].join("\n")

const EXT = { maxNb: 8, seed: 3, materials: [{ type: "lambert", albedo: [0.8, 0.35, 0.05] }, { type: "mirror", albedo: [0.9, 0.9, 0.9] },
                                            { type: "emitter", emission: [6, 4, 2] }] }

function arg(name, dflt) {
    const i = process.argv.indexOf("--" + name)
    return i > 0 ? (process.argv[i + 1] === undefined || process.argv[i + 1].startsWith("--") ? true : process.argv[i + 1]) : dflt
}

async function main() {
    const width = +arg("width", 512), height = +arg("height", 512), samples = +arg("samples", 16), grid = +arg("grid", 10)
    const gpus = Math.max(1, +arg("gpus", 1)), t0 = Date.now()
    const shape = (arg("paths", "") === "" && arg("pools", "") === "") ? pt.defaultPoolShape(width, height) : { numPaths: +arg("paths", 0), pools: +arg("pools", 1) }
    const settings = Object.assign({ samples }, shape, arg("ext", false) ? EXT : {})
    const camera = { lookAt: [10, 10, 0], position: [15, 15, 5], fov: 60 }       // main.js:42-46
    const devices = [], scenes = [], tracers = []
    for (let g = 0; g < gpus; g++) {                 // one device, one scene replica and one tracer per GPU (G = 1: the reference's sequence)
        const device = pt.requestDevice(+arg("device", 0) + g, arg("traversal", "") ? { traversal: arg("traversal", "") } : {})
        const scene = pt.initScene(device)
        const box = scene.registerMesh({ file: BOX })
        const floor = scene.registerMesh({ file: FLOOR })
        let lcg = 12345                              // seeded stand-in for Math.random() (main.js:28) so runs are repeatable
        for (let i = 0; i < grid * grid; i++) {
            lcg = (Math.imul(lcg, 1664525) + 1013904223) >>> 0
            scene.instanceMesh(box, [(i % grid) * 2, Math.floor(i / grid) * 2, -1.3], [0, 0, lcg / 4294967296 * Math.PI * 2.], [1, 1, 1],
                               arg("ext", false) ? 1 + i % 4 : 1)
        }
        scene.instanceMesh(floor, [10, 10, -1.3], [0, 0, 0], [5, 5, 5], 0)
        await scene.build()
        devices.push(device); scenes.push(scene)
        tracers.push(pt.initPathTracer({ device, scene, image: { width, height }, camera,
                                         settings: gpus > 1 ? pt.shareOf(g, gpus, { width, height }, settings) : settings }))
    }
    const device = devices[0], scene = scenes[0]
    const multi = gpus > 1 || arg("force-comm", false) ? pt.initMultiGpu(devices) : null
    const tracer = gpus === 1 ? tracers[0] : {       // every GPU takes the step; the queue sizes add up
        async step() { const r = []; for (const t of tracers) r.push(await t.step()); const sum = f => r.reduce((a, x) => a + x[f], 0)
                       return { camera: sum("camera"), material: sum("material"), rays: sum("rays") } },
        getImageBuffer: tracers[0].getImageBuffer, readImage: () => tracers[0].readImage(), destroy() { tracers.forEach(t => t.destroy()) },
    }
    const display = pt.initDisplay({ device, canvas: arg("out", null), image: { buffer: tracer.getImageBuffer(), width, height } })

    // the reference's frame loop never learns that the render finished (main.js:61-72); headless, it stops on empty queues
    let steps = 0, rays = 0, frames = 0
    const t1 = Date.now()
    for (;;) {
        const st = await tracer.step()
        steps++; rays += st.rays
        if (+arg("progress", 0) > 0 && steps % +arg("progress", 0) === 0) { display.draw(); frames++ }     // main.js:69
        if (st.camera + st.material === 0) break
    }
    if (multi) multi.reduceImage(0)                  // the GPUs' accumulation buffers, summed onto GPU 0 (RCCL over xGMI)
    const t2 = Date.now()
    const rgba = display.draw()
    const img = tracer.readImage()
    let wsum = 0, rsum = 0
    for (let i = 0; i < width * height; i++) { wsum += img[4 * i + 3]; rsum += img[4 * i] }
    const report = { width, height, samples, steps, rays, gpus, reduced: !!multi, frames: frames + 1, buildMs: t1 - t0, renderMs: t2 - t1, sampleCount: wsum, radianceSum: rsum,
                     firstPixel: Array.from(rgba.slice(0, 4)), info: scene.kernels.getSceneBindGroupInfo() }
    console.log(arg("json", false) ? JSON.stringify(report) : report)
    if (multi) multi.destroy()
    tracer.destroy()                 // pool contexts borrow the device's scene: they go first
    devices.forEach(d => d.destroy())
}

main().catch(e => { console.error(e.message || e); process.exit(1) })
