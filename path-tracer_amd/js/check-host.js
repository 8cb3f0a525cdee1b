// check-host.js -- compares this host's parseObj / computeTransformMatrices / getTransformedBoundingBox with the golden
// vectors the reference's own JS produced (tests/golden/host_fixtures.json).  Exit code 0 = all equal.  No GPU needed.
"use strict"
const fs = require("fs"), path = require("path")
const host = require("./pt-host.js")
const fix = JSON.parse(fs.readFileSync(path.join(__dirname, "..", "..", "tests", "golden", "host_fixtures.json"), "utf8"))
const same = (a, b) => JSON.stringify(a) === JSON.stringify(b)
let bad = 0
for (const c of fix.parseObj) {
    const r = host.parseObj(c.text)
    if (r.numTriangles !== c.numTriangles || !same(r.triangleArray, c.triangleArray) || !same(r.bounds, c.bounds)) { console.log("parseObj mismatch:", c.name); bad++ }
}
for (const c of fix.transforms) {
    const m = host.computeTransformMatrices(c.transform)
    if (!same(m.localToWorld, c.localToWorld) || !same(m.worldToLocal, c.worldToLocal)) { console.log("matrix mismatch:", JSON.stringify(c.transform)); bad++ }
    c.boxes.forEach((b, i) => { if (!same(host.getTransformedBoundingBox(m, b), c.transformedBoxes[i])) { console.log("box mismatch"); bad++ } })
}
console.log(bad ? `${bad} mismatches` : `host logic matches the reference on ${fix.parseObj.length} OBJ and ${fix.transforms.length} transform fixtures`)
process.exit(bad ? 1 : 0)
