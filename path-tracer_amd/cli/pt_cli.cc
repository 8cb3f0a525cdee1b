// pt_cli.cc -- Node-free host for libpt_amd.so (SURVEY 8b "caller", 8f.3): the reference's main.js call sequence
// (registerMesh x k -> instanceMesh x m -> build -> initPathTracer -> loop { step } -> display) in C++ over the C ABI.
//
//   pt_cli [--obj FILE]... [--instance MESH,px,py,pz,rx,ry,rz,sx,sy,sz,MATERIAL]... [--camera px,py,pz,lx,ly,lz,fov]
//          [--width W] [--height H] [--samples S] [--paths N] [--pools K] [--device D] [--gpus G] [--progress K] [--traversal reference|tolerance|auto] [--out frame.png|frame.ppm] [--json]
//          [--material TYPE,r,g,b[,e0,e1,e2]]... [--max-nb N] [--seed S]      extensions (material ids 2, 3, ...; TYPE lambert|emitter|mirror|glossy)
// --gpus G: the samples are split over GPUs 0..G-1 of this node -- every GPU gets the scene, its own RNG streams and the
// g-th part of the pixel tickets -- and the accumulation buffers are summed onto GPU 0 by one RCCL reduce (pt_comm_reduce_image).
//   pt_cli --dump-host FILE px,py,pz,rx,ry,rz,sx,sy,sz      host-side results only (no GPU): parse + matrices + world box as JSON
//
// Without --obj the scene is the reference's bundled floor with its main.js:33 instance and main.js:42-46 camera.
// --progress K writes the frame every K steps (the reference redraws every animation frame, main.js:61-72).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/pt_amd.h"
#include "pt_host.hpp"

using namespace pt_host;

static const char* FLOOR_OBJ = "v -10.0 -10.0 0.0\nv  10.0 -10.0 0.0\nv  10.0  10.0 0.0\nv -10.0  10.0 0.0\n\nf 1 2 3\nf 1 3 4";

static std::vector<double> csv(const std::string& s) {
    std::vector<double> v; std::stringstream ss(s); std::string t;
    while (std::getline(ss, t, ',')) v.push_back(std::strtod(t.c_str(), nullptr));
    return v;
}
static bool read_file(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary); if (!f) return false;
    std::stringstream ss; ss << f.rdbuf(); out = ss.str(); return true;
}
static void json_arr(const double* v, int n) {      // NaN -> null, +-inf -> (-)Infinity (what Python's json reads back)
    printf("[");
    for (int i = 0; i < n; i++) {
        const char* sep = i ? ", " : "";
        if (v[i] != v[i]) printf("%snull", sep);
        else if (std::isinf(v[i])) printf("%s%sInfinity", sep, v[i] < 0 ? "-" : "");
        else printf("%s%.17g", sep, v[i]);
    }
    printf("]");
}

// ---- PNG without zlib: stored deflate blocks ----------------------------------------------------------------------------
static uint32_t crc_table[256];
static void crc_init() { for (uint32_t n = 0; n < 256; n++) { uint32_t c = n; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1; crc_table[n] = c; } }
static uint32_t crc32(const uint8_t* p, size_t n, uint32_t c = 0xffffffffu) { for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8); return c; }
static void be32(std::vector<uint8_t>& v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }
static void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    be32(out, (uint32_t)data.size());
    std::vector<uint8_t> td(type, type + 4); td.insert(td.end(), data.begin(), data.end());
    out.insert(out.end(), td.begin(), td.end());
    be32(out, crc32(td.data(), td.size()) ^ 0xffffffffu);
}
static bool write_image(const std::string& path, const uint8_t* rgba, int w, int h) {
    FILE* f = fopen(path.c_str(), "wb"); if (!f) return false;
    if (path.size() > 4 && path.substr(path.size() - 4) == ".ppm") {
        fprintf(f, "P6\n%d %d\n255\n", w, h);
        for (int i = 0; i < w * h; i++) fwrite(rgba + 4 * i, 1, 3, f);
    } else {
        crc_init();
        std::vector<uint8_t> raw; raw.reserve((size_t)h * (4 * w + 1));
        for (int y = 0; y < h; y++) { raw.push_back(0); raw.insert(raw.end(), rgba + (size_t)4 * w * y, rgba + (size_t)4 * w * (y + 1)); }
        std::vector<uint8_t> z = {0x78, 0x01};
        uint32_t a = 1, b = 0;
        for (uint8_t c : raw) { a = (a + c) % 65521; b = (b + a) % 65521; }
        for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
            size_t n = std::min<size_t>(65535, raw.size() - off);
            z.push_back(off + n >= raw.size() ? 1 : 0);
            z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
            z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
            if (raw.empty()) break;
        }
        be32(z, (b << 16) | a);
        std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a}, ihdr;
        be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h); ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0});
        chunk(out, "IHDR", ihdr); chunk(out, "IDAT", z); chunk(out, "IEND", {});
        fwrite(out.data(), 1, out.size(), f);
    }
    fclose(f);
    return true;
}

static int dump_host(const std::string& file, const std::string& trs) {
    std::string text; if (!read_file(file, text)) { fprintf(stderr, "cannot read %s\n", file.c_str()); return 2; }
    ParsedObj p = parse_obj(text);
    std::vector<double> t = csv(trs); t.resize(9, 0.0);
    TransformMatrices m = compute_transform_matrices(&t[0], &t[3], &t[6]);
    Bounds wb = get_transformed_bounding_box(m.l2w, p.bounds);
    printf("{\"numTriangles\": %lld, \"triangleArray\": ", (long long)p.num_triangles()); json_arr(p.tris.data(), (int)p.tris.size());
    printf(", \"bounds\": {\"min\": "); json_arr(p.bounds.mn, 3); printf(", \"max\": "); json_arr(p.bounds.mx, 3);
    printf("}, \"localToWorld\": "); json_arr(m.l2w.m, 16); printf(", \"worldToLocal\": "); json_arr(m.w2l.m, 16);
    printf(", \"worldBox\": {\"min\": "); json_arr(wb.mn, 3); printf(", \"max\": "); json_arr(wb.mx, 3); printf("}}\n");
    return 0;
}

#define CHECK(call) do { int rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, pt_last_error(ctx)); if (ctx) pt_destroy(ctx); return 1; } } while (0)

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);      // before the HIP runtime starts: four path pools (the 4K default) need more than HIP's four hardware queues; a caller's value is kept
    std::vector<std::string> objs, insts; std::string cam, out; bool json = false;
    int width = 512, height = 512, samples = 16, paths = 0, device = 0, progress = 0, pools = 0, gpus = 1;
    pt_render_ext ext; memset(&ext, 0, sizeof ext); bool have_ext = false; int traversal = PT_TRAVERSAL_REFERENCE;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--dump-host") { std::string f = next(); return dump_host(f, next()); }
        else if (a == "--obj") objs.push_back(next());
        else if (a == "--instance") insts.push_back(next());
        else if (a == "--camera") cam = next();
        else if (a == "--width") width = atoi(next().c_str());
        else if (a == "--height") height = atoi(next().c_str());
        else if (a == "--samples") samples = atoi(next().c_str());
        else if (a == "--paths") paths = atoi(next().c_str());
        else if (a == "--pools") pools = std::max(1, atoi(next().c_str()));
        else if (a == "--device") device = atoi(next().c_str());
        else if (a == "--gpus") gpus = std::max(1, atoi(next().c_str()));
        else if (a == "--max-nb") { ext.max_nb = atoi(next().c_str()); have_ext = true; }
        else if (a == "--seed") { ext.seed = (uint32_t)strtoul(next().c_str(), nullptr, 10); have_ext = true; }
        else if (a == "--material") {
            std::string spec = next(); size_t comma = spec.find(',');
            std::string type = spec.substr(0, comma); std::vector<double> v = comma == std::string::npos ? std::vector<double>{} : csv(spec.substr(comma + 1));
            v.resize(6, 0.0);
            if (ext.n_materials >= PT_MAX_EXT_MATERIALS) { fprintf(stderr, "at most %d --material\n", PT_MAX_EXT_MATERIALS); return 2; }
            pt_material& m = ext.materials[ext.n_materials++];
            m.type = type == "lambert" ? PT_MAT_LAMBERT : type == "emitter" ? PT_MAT_EMITTER : type == "mirror" ? PT_MAT_MIRROR : type == "glossy" ? PT_MAT_GLOSSY : -1;
            if (m.type < 0) { fprintf(stderr, "unknown material type %s\n", type.c_str()); return 2; }
            for (int k = 0; k < 3; k++) { m.albedo[k] = (float)v[k]; m.emission[k] = (float)v[3 + k]; }     // glossy: e0 = roughness
            have_ext = true;
        }
        else if (a == "--traversal") { std::string m = next(); traversal = m == "reference" ? PT_TRAVERSAL_REFERENCE : m == "tolerance" ? PT_TRAVERSAL_WIDE : m == "auto" ? PT_TRAVERSAL_AUTO : -1;
                                        if (traversal < 0) { fprintf(stderr, "--traversal reference|tolerance|auto\n"); return 2; } }
        else if (a == "--progress") progress = atoi(next().c_str());
        else if (a == "--out") out = next();
        else if (a == "--json") json = true;
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    // neither --paths nor --pools: the measured-best shape for the frame (three pools of a quarter of its pixels each, 750,000 .. 2,000,000
    // slots); the reference's own shape is --paths 1000000 --pools 1 (path-tracer.js:4), 38 % slower on MI355X
    // at 4K, with at least 8 hardware queues (GPU_MAX_HW_QUEUES, set to 8 at the top of main unless the caller has set it), four pools of 1,500,000
    if (paths <= 0 && pools <= 0) {
        pools = 3; paths = (int)std::min<long long>(2000000, std::max<long long>(750000, (long long)width * height / 4));
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        if (paths >= 1500000 && q && atoi(q) >= 8) { pools = 4; paths = paths / 4 * 3; }
    }
    pools = std::max(1, pools);
    std::vector<std::string> texts;
    if (objs.empty()) { texts.push_back(FLOOR_OBJ); if (insts.empty()) insts.push_back("0,10,10,-1.3,0,0,0,5,5,5,0"); }   // main.js:33
    for (const std::string& f : objs) { std::string t; if (!read_file(f, t)) { fprintf(stderr, "cannot read %s\n", f.c_str()); return 2; } texts.push_back(t); }
    if (insts.empty()) for (size_t m = 0; m < texts.size(); m++) insts.push_back(std::to_string(m) + ",0,0,0,0,0,0,1,1,1,0");
    std::vector<double> c = cam.empty() ? std::vector<double>{15, 15, 5, 10, 10, 0, 60} : csv(cam);                       // main.js:42-46
    c.resize(7, 60.0);

    pt_ctx* ctx = nullptr;                                                        // GPU `device`; with --gpus G also device + 1 .. device + G - 1
    std::vector<pt_ctx*> gpu_ctx;
    std::vector<int> handles; std::vector<Bounds> bounds;
    std::vector<ParsedObj> parsed;
    for (const std::string& t : texts) parsed.push_back(parse_obj(t));
    for (int g = 0; g < gpus; g++) {
        pt_ctx* cg = nullptr;
        { int rc_ = pt_create(device + g, &cg); if (rc_ < 0) { fprintf(stderr, "pt_create(%d) failed (%d): %s\n", device + g, rc_, pt_last_error(nullptr)); return 1; } }
        if (traversal != PT_TRAVERSAL_REFERENCE && pt_set_traversal_mode(cg, traversal) < 0) { fprintf(stderr, "pt_set_traversal_mode: %s\n", pt_last_error(cg)); return 1; }
        if (g == 0) ctx = cg;
        gpu_ctx.push_back(cg);
        for (const ParsedObj& p : parsed) {                                       // registerMesh (scene.js:518), once per GPU: the scene is replicated
            std::vector<float> tris(p.tris.begin(), p.tris.end());                // builders.js:164 Float32Array.set
            float b6[6]; for (int k = 0; k < 3; k++) { b6[k] = (float)p.bounds.mn[k]; b6[3 + k] = (float)p.bounds.mx[k]; }
            int h = pt_mesh_upload(cg, tris.data(), p.num_triangles(), b6);
            if (h < 0) { fprintf(stderr, "pt_mesh_upload failed (%d): %s\n", h, pt_last_error(cg)); return 1; }
            if (g == 0) { handles.push_back(h); bounds.push_back(p.bounds); }
        }
    }
    std::vector<pt_instance> inst;
    for (const std::string& s : insts) {                                          // instanceMesh (scene.js:523-529) + scene.js:35-46
        std::vector<double> v = csv(s); v.resize(11, 0.0);
        int mesh = (int)v[0];
        if (mesh < 0 || mesh >= (int)handles.size()) { fprintf(stderr, "warning: instance of unknown mesh %d skipped\n", mesh); continue; }   // scene.js:525
        TransformMatrices m = compute_transform_matrices(&v[1], &v[4], &v[7]);
        Bounds wb = get_transformed_bounding_box(m.l2w, bounds[mesh]);
        pt_instance pi; memset(&pi, 0, sizeof pi);
        for (int k = 0; k < 16; k++) { pi.l2w[k] = (float)m.l2w.m[k]; pi.w2l[k] = (float)m.w2l.m[k]; }
        for (int k = 0; k < 3; k++) { pi.aabb_min[k] = (float)wb.mn[k]; pi.aabb_max[k] = (float)wb.mx[k]; }
        pi.mesh = handles[mesh]; pi.material = (int)v[10];
        inst.push_back(pi);
    }
    for (pt_ctx* cg : gpu_ctx) CHECK(pt_scene_build(cg, inst.data(), (int)inst.size()));
    pt_scene_info si; CHECK(pt_scene_get_info(ctx, &si));

    pt_render_params rp; memset(&rp, 0, sizeof rp);
    rp.width = width; rp.height = height; rp.spp = samples; rp.num_paths = paths; rp.ticket_end = -1; rp.fov_deg = (float)c[6];
    for (int k = 0; k < 3; k++) { rp.cam_pos[k] = (float)c[k]; rp.cam_look_at[k] = (float)c[3 + k]; }
    // --pools K > 1 (no reference counterpart, pt_amd.h "several path pools on one GPU"): K contexts that borrow the scene,
    // pool k with its own RNG streams and the k-th part of the pixel tickets, all adding into pool 0's image
    // --gpus G: the same split one level up -- GPU g takes the g-th part of the tickets (and its pools split that part)
    std::vector<pt_ctx*> pool;                                                    // pool[g * pools + k]
    const long long per_pool = paths > 0 ? paths : 1000000, total = ((long long)width * height * samples + 7) / 8;   // logic.js:177
    const int shares = gpus * pools;
    for (int g = 0; g < gpus; g++)
        for (int k = 0; k < pools; k++) {
            pt_ctx* pc = gpu_ctx[g];
            if (k > 0) CHECK(pt_create_sharing_scene(gpu_ctx[g], &pc));
            pool.push_back(pc);
            const int sh = g * pools + k;
            rp.image_dev = nullptr;
            if (shares > 1) {
                rp.num_paths = (int32_t)per_pool; rp.slot_base = sh * per_pool; rp.ticket_begin = total * sh / shares; rp.ticket_end = total * (sh + 1) / shares;
                if (k > 0) CHECK(pt_image_dev_ptr(gpu_ctx[g], &rp.image_dev));
            }
            CHECK(pt_render_init(pc, &rp));
            if (have_ext) CHECK(pt_render_set_extensions(pc, &ext));
        }
    pt_comm* comm = nullptr;
    if (gpus > 1 || getenv("PT_CLI_FORCE_COMM")) CHECK(pt_comm_create(gpu_ctx.data(), gpus, &comm));   // one rank per GPU

    std::vector<uint8_t> rgba((size_t)width * height * 4);
    long long steps = 0, rays = 0; int frames = 0;
    for (;;) {                                                                    // main.js:61-72, stopping on empty queues
        pt_step_stats st; memset(&st, 0, sizeof st);
        for (pt_ctx* pc : pool) { pt_step_stats s1; CHECK(pt_render_step(pc, &s1)); st.n_rays += s1.n_rays; st.n_camera += s1.n_camera; st.n_material += s1.n_material; }
        steps++; rays += st.n_rays;
        bool done = st.n_camera + st.n_material == 0;
        if (done && comm) CHECK(pt_comm_reduce_image(comm, 0));                   // sum of the GPUs' accumulation buffers onto GPU 0 (RCCL)
        if (!out.empty() && (done || (progress > 0 && steps % progress == 0))) {
            CHECK(pt_tonemap_rgba8(ctx, nullptr, 0, 0, rgba.data()));
            if (!write_image(out, rgba.data(), width, height)) { fprintf(stderr, "cannot write %s\n", out.c_str()); pt_destroy(ctx); return 2; }
            frames++;
        }
        if (done) break;
    }
    std::vector<float> img((size_t)width * height * 4);
    for (int k = pools - 1; k >= 0; k--) CHECK(pt_image_read(pool[k], img.data(), (int64_t)img.size()));   // GPU 0's pools; each read adds that pool's open tickets
    double wsum = 0, rsum = 0; for (size_t i = 0; i < (size_t)width * height; i++) { wsum += img[4 * i + 3]; rsum += img[4 * i]; }
    if (json) printf("{\"width\": %d, \"height\": %d, \"samples\": %d, \"steps\": %lld, \"rays\": %lld, \"sampleCount\": %.17g, \"radianceSum\": %.17g, "
                     "\"nTris\": %lld, \"nObjects\": %d, \"frames\": %d, \"gpus\": %d, \"reduced\": %s}\n", width, height, samples, steps, rays, wsum, rsum, (long long)si.n_tris, si.n_objects, frames, gpus, comm ? "true" : "false");
    else printf("%dx%d, %d spp: %lld steps, %lld rays, %lld triangles in %d instances%s%s\n", width, height, samples, steps, rays,
                (long long)si.n_tris, si.n_objects, out.empty() ? "" : " -> ", out.c_str());
    if (comm) pt_comm_destroy(comm);
    for (int g = gpus - 1; g >= 0; g--) {
        for (int k = pools - 1; k > 0; k--) pt_destroy(pool[g * pools + k]);      // the borrowers first
        pt_destroy(gpu_ctx[g]);
    }
    return 0;
}
