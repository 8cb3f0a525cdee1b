// Generates tests/golden/host_fixtures.json by running the reference's two pure-JS host files
// (scripts/scene/util/parse-obj.js, transform-utils.js) under Node.  Run in the build container only
// (the reference is not present on the GPU box):   node tests/golden/make_host_fixtures.js
// The fixture holds inputs and the reference's outputs; no reference source is copied.
const fs = require("fs"), vm = require("vm"), path = require("path")
const REF = process.env.PT_REFERENCE || "/root/reference"
for (const f of ["scripts/scene/util/parse-obj.js", "scripts/scene/util/transform-utils.js"])
    vm.runInThisContext(fs.readFileSync(path.join(REF, f), "utf8"), { filename: f })

const objs = {
    floor: fs.readFileSync(path.join(REF, "media/floor.obj"), "utf8"),
    quad_and_slashes: "# comment\nv 0 0 0\nv 1 0 0\nv 1 1 0.5\nv 0 1 -2.5e-1\nvn 0 0 1\nvt 0 0\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf 1//1 3//1 4//1\n",
    object_reset: "o first\nv 0 0 0\nv 2 0 0\nv 0 2 0\nf 1 2 3\no second\nv 5 5 5\nv 6 5 5\nv 5 6 5\nv 5 5 6\nf 1 2 3\nf 1 3 4\nf 2 3 4\n",
    pentagon_fan: "v 1 0 0\nv 0.309 0.951 0\nv -0.809 0.588 0\nv -0.809 -0.588 0\nv 0.309 -0.951 0\nf 1 2 3 4 5\n",
    tabs_and_blank: "v\t1.5\t2.5\t3.5\nv 4 5 6\n\nv 7 8 10\nf  1  2  3\ng ignored\ns off\n",
    unreferenced_vertex_bounds: "v 100 -100 50\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 2 3 4\n",
}
const transforms = [
    { position: [10, 10, -1.3], rotation: [0, 0, 0], scale: [5, 5, 5] },
    { position: [0, 0, 0], rotation: [0, 0, 0], scale: [1, 1, 1] },
    { position: [2, 4, 0], rotation: [Math.PI / 2, 0, 1.2345], scale: [1, 1, 1] },
    { position: [-3.5, 0.25, 7], rotation: [0.3, -1.1, 2.9], scale: [0.5, 2, 1.25] },
    { position: [1e3, -1e-3, 0], rotation: [3.0, 0.1, -0.7], scale: [3, 3, 0.1] },
]
const boxes = [{ min: [-10, -10, 0], max: [10, 10, 0] }, { min: [-1, -2, -3], max: [4, 5, 6] }]

const out = { generator: "tests/golden/make_host_fixtures.js", node: process.version, parseObj: [], transforms: [] }
for (const name in objs) {
    const r = parseObj(objs[name])
    out.parseObj.push({ name, text: objs[name], numTriangles: r.numTriangles, triangleArray: r.triangleArray, bounds: r.bounds })
}
for (const t of transforms) {
    const m = computeTransformMatrices(t)
    const bb = boxes.map(b => getTransformedBoundingBox(m, b))
    out.transforms.push({ transform: t, localToWorld: m.localToWorld, worldToLocal: m.worldToLocal, boxes, transformedBoxes: bb })
}
fs.writeFileSync(path.join(__dirname, "host_fixtures.json"), JSON.stringify(out))
console.log("wrote", out.parseObj.length, "parseObj and", out.transforms.length, "transform fixtures")
