// addon.cc -- thin N-API binding of the C ABI in include/pt_amd.h for the headless Node.js host.
// One JS function per pt_* entry point; typed arrays go through as raw pointers, nothing is computed here.
// Built with plain g++ against /usr/include/node/node_api.h (N-API 8, Node 12) and linked to ../libpt_amd.so.
#include <node_api.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include "pt_amd.h"

#define NAPI_OK(call)                                                                  \
    do {                                                                               \
        if ((call) != napi_ok) { napi_throw_error(env, nullptr, "N-API call failed: " #call); return nullptr; } \
    } while (0)

static napi_value throw_pt(napi_env env, pt_ctx* c, const char* what, int rc) {
    std::string msg = std::string(what) + " failed (" + std::to_string(rc) + "): " + pt_last_error(c);
    napi_throw_error(env, nullptr, msg.c_str());
    return nullptr;
}

static bool get_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < want) {
        napi_throw_type_error(env, nullptr, "wrong number of arguments");
        return false;
    }
    return true;
}
static pt_ctx* ctx_of(napi_env env, napi_value v) {
    void* p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) { napi_throw_type_error(env, nullptr, "expected a pt context"); return nullptr; }
    if (!*(pt_ctx**)p) { napi_throw_error(env, nullptr, "context destroyed"); return nullptr; }
    return *(pt_ctx**)p;
}
// typed array -> pointer + byte length (null for JS null/undefined)
static bool ta(napi_env env, napi_value v, void** data, size_t* bytes) {
    napi_valuetype t; napi_typeof(env, v, &t);
    *data = nullptr; *bytes = 0;
    if (t == napi_null || t == napi_undefined) return true;
    bool is_ta = false; napi_is_typedarray(env, v, &is_ta);
    if (!is_ta) { napi_throw_type_error(env, nullptr, "expected a TypedArray"); return false; }
    napi_typedarray_type tt; size_t len; napi_value ab; size_t off;
    if (napi_get_typedarray_info(env, v, &tt, &len, data, &ab, &off) != napi_ok) return false;
    static const size_t esz[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
    *bytes = len * esz[tt];
    return true;
}
static int64_t i64(napi_env env, napi_value v) { int64_t x = 0; napi_get_value_int64(env, v, &x); return x; }
static double f64(napi_env env, napi_value v) { double x = 0; napi_get_value_double(env, v, &x); return x; }
static napi_value num(napi_env env, double x) { napi_value v; napi_create_double(env, x, &v); return v; }
static void set(napi_env env, napi_value obj, const char* k, double x) { napi_set_named_property(env, obj, k, num(env, x)); }
static double prop(napi_env env, napi_value obj, const char* k, double dflt) {
    bool has = false; napi_has_named_property(env, obj, k, &has);
    if (!has) return dflt;
    napi_value v; napi_get_named_property(env, obj, k, &v);
    napi_valuetype t; napi_typeof(env, v, &t);
    return t == napi_number ? f64(env, v) : dflt;
}
static void prop3(napi_env env, napi_value obj, const char* k, float out[3]) {
    napi_value a; napi_get_named_property(env, obj, k, &a);
    for (uint32_t i = 0; i < 3; i++) { napi_value e; napi_get_element(env, a, i, &e); out[i] = (float)f64(env, e); }
}

static void finalize_ctx(napi_env, void* data, void*) {
    pt_ctx** h = (pt_ctx**)data;
    if (*h) pt_destroy(*h);
    delete h;
}

// create(deviceId) -> context handle            [pt_create]
static napi_value Create(napi_env env, napi_callback_info info) {
    napi_value a[1]; if (!get_args(env, info, 1, a)) return nullptr;
    pt_ctx* c = nullptr; int rc = pt_create((int)i64(env, a[0]), &c);
    if (rc < 0) return throw_pt(env, nullptr, "pt_create", rc);
    pt_ctx** h = new pt_ctx*(c);
    napi_value ext; NAPI_OK(napi_create_external(env, h, finalize_ctx, nullptr, &ext));
    return ext;
}
// createSharingScene(ctx) -> context handle that borrows ctx's built scene      [pt_create_sharing_scene]
static napi_value CreateSharingScene(napi_env env, napi_callback_info info) {
    napi_value a[1]; if (!get_args(env, info, 1, a)) return nullptr;
    pt_ctx* src = ctx_of(env, a[0]); if (!src) return nullptr;
    pt_ctx* c = nullptr; int rc = pt_create_sharing_scene(src, &c);
    if (rc < 0) return throw_pt(env, src, "pt_create_sharing_scene", rc);
    pt_ctx** h = new pt_ctx*(c);
    napi_value ext; NAPI_OK(napi_create_external(env, h, finalize_ctx, nullptr, &ext));
    return ext;
}
// destroy(ctx)                                  [pt_destroy]
static napi_value Destroy(napi_env env, napi_callback_info info) {
    napi_value a[1]; if (!get_args(env, info, 1, a)) return nullptr;
    void* p = nullptr; napi_get_value_external(env, a[0], &p);
    if (p && *(pt_ctx**)p) {
        pt_ctx* c = *(pt_ctx**)p;
        int rc = pt_destroy(c);                    // refused while pool contexts still borrow this one's scene
        if (rc < 0) return throw_pt(env, c, "pt_destroy", rc);
        *(pt_ctx**)p = nullptr;
    }
    return nullptr;
}
// meshUpload(ctx, Float32Array tris12, Float32Array bounds6) -> handle     [pt_mesh_upload]
static napi_value MeshUpload(napi_env env, napi_callback_info info) {
    napi_value a[3]; if (!get_args(env, info, 3, a)) return nullptr;
    pt_ctx* c = ctx_of(env, a[0]); if (!c) return nullptr;
    void *t, *b; size_t tb, bb;
    if (!ta(env, a[1], &t, &tb) || !ta(env, a[2], &b, &bb)) return nullptr;
    if (bb != 24 || tb % 48) { napi_throw_range_error(env, nullptr, "meshUpload: tris must be 12 floats per triangle, bounds 6 floats"); return nullptr; }
    int rc = pt_mesh_upload(c, (const float*)t, (int64_t)(tb / 48), (const float*)b);
    if (rc < 0) return throw_pt(env, c, "pt_mesh_upload", rc);
    return num(env, rc);
}
// sceneBuild(ctx, ArrayBuffer-backed typed array of pt_instance records (40 words each), n)   [pt_scene_build]
static napi_value SceneBuild(napi_env env, napi_callback_info info) {
    napi_value a[3]; if (!get_args(env, info, 3, a)) return nullptr;
    pt_ctx* c = ctx_of(env, a[0]); if (!c) return nullptr;
    void* d; size_t nb; if (!ta(env, a[1], &d, &nb)) return nullptr;
    int n = (int)i64(env, a[2]);
    if (nb < (size_t)n * sizeof(pt_instance)) { napi_throw_range_error(env, nullptr, "sceneBuild: instance buffer too small"); return nullptr; }
    int rc = pt_scene_build(c, (const pt_instance*)d, n);
    if (rc < 0) return throw_pt(env, c, "pt_scene_build", rc);
    return nullptr;
}
// sceneInfo(ctx) -> {nTris, nObjects, nMeshesUsed}                          [pt_scene_get_info]
static napi_value SceneInfo(napi_env env, napi_callback_info info) {
    napi_value a[1]; if (!get_args(env, info, 1, a)) return nullptr;
    pt_ctx* c = ctx_of(env, a[0]); if (!c) return nullptr;
    pt_scene_info si; int rc = pt_scene_get_info(c, &si);
    if (rc < 0) return throw_pt(env, c, "pt_scene_get_info", rc);
    napi_value o; napi_create_object(env, &o);
    set(env, o, "nTris", (double)si.n_tris); set(env, o, "nObjects", si.n_objects); set(env, o, "nMeshesUsed", si.n_meshes_used);
    return o;
}
// readBuffer(ctx, which, typedArray[, mesh]): which in blasNodes|blasTris|tlasNodes|objects|tlasPerm|meshPerm   [pt_get_*]
static napi_value ReadBuffer(napi_env env, napi_callback_info info) {
    napi_value a[4]; size_t argc = 4; napi_get_cb_info(env, info, &argc, a, nullptr, nullptr);
    if (argc < 3) { napi_throw_type_error(env, nullptr, "readBuffer(ctx, which, out[, mesh])"); return nullptr; }
    pt_ctx* c = ctx_of(env, a[0]); if (!c) return nullptr;
    char which[32]; size_t wl; napi_get_value_string_utf8(env, a[1], which, sizeof which, &wl);
    void* d; size_t nb; if (!ta(env, a[2], &d, &nb)) return nullptr;
    int rc = PT_ERR_ARG;
    if (!strcmp(which, "blasNodes")) rc = pt_get_blas_nodes(c, (pt_bvh_node*)d, (int64_t)(nb / 64));
    else if (!strcmp(which, "blasTris")) rc = pt_get_blas_tris(c, (pt_triangle*)d, (int64_t)(nb / 48));
    else if (!strcmp(which, "tlasNodes")) rc = pt_get_tlas_nodes(c, (pt_bvh_node*)d, (int64_t)(nb / 64));
    else if (!strcmp(which, "objects")) rc = pt_get_objects(c, (pt_object*)d, (int64_t)(nb / 128));
    else if (!strcmp(which, "tlasPerm")) rc = pt_get_tlas_perm(c, (int32_t*)d, (int64_t)(nb / 4));
    else if (!strcmp(which, "meshPerm") && argc == 4) rc = pt_get_mesh_perm(c, (int)i64(env, a[3]), (int32_t*)d, (int64_t)(nb / 4));
    if (rc < 0) return throw_pt(env, c, "readBuffer", rc);
    return nullptr;
}
// traceRays(ctx, Float32Array o3, Float32Array d3, Int32Array hitObj, Int32Array hitTri, Float32Array hitDist|null)   [pt_trace_rays]
static napi_value TraceRays(napi_env env, napi_callback_info info) {
    napi_value a[6]; if (!get_args(env, info, 6, a)) return nullptr;
    pt_ctx* c = ctx_of(env, a[0]); if (!c) return nullptr;
    void* p[5]; size_t nb[5];
    for (int i = 0; i < 5; i++) if (!ta(env, a[1 + i], &p[i], &nb[i])) return nullptr;
    int64_t n = (int64_t)(nb[0] / 12);
    if (nb[1] != nb[0] || (p[2] && nb[2] != (size_t)n * 4) || (p[3] && nb[3] != (size_t)n * 4) || (p[4] && nb[4] != (size_t)n * 4)) {
        napi_throw_range_error(env, nullptr, "traceRays: array sizes disagree"); return nullptr;
    }
    int rc = pt_trace_rays(c, (const float*)p[0], (const float*)p[1], n, (int32_t*)p[2], (int32_t*)p[3], (float*)p[4], nullptr);
    if (rc < 0) return throw_pt(env, c, "pt_trace_rays", rc);
    return nullptr;
}
// renderInit(ctx, {width,height,position[3],lookAt[3],fov,samples,numPaths,slotBase,ticketBegin,ticketEnd})   [pt_render_init]
static napi_value RenderInit(napi_env env, napi_callback_info info) {
    napi_value a[2]; if (!get_args(env, info, 2, a)) return nullptr;
    pt_ctx* c = ctx_of(env, a[0]); if (!c) return nullptr;
    pt_render_params p; memset(&p, 0, sizeof p);
    p.width = (int32_t)prop(env, a[1], "width", 0); p.height = (int32_t)prop(env, a[1], "height", 0);
    prop3(env, a[1], "position", p.cam_pos); prop3(env, a[1], "lookAt", p.cam_look_at);
    p.fov_deg = (float)prop(env, a[1], "fov", 60); p.spp = (int32_t)prop(env, a[1], "samples", 8);
    p.num_paths = (int32_t)prop(env, a[1], "numPaths", 0); p.slot_base = (int64_t)prop(env, a[1], "slotBase", 0);
    p.ticket_begin = (int64_t)prop(env, a[1], "ticketBegin", 0); p.ticket_end = (int64_t)prop(env, a[1], "ticketEnd", -1);
    bool has = false; napi_has_named_property(env, a[1], "imageOf", &has);       // accumulate into another context's image (pools)
    if (has) {
        napi_value v; napi_get_named_property(env, a[1], "imageOf", &v);
        napi_valuetype t; napi_typeof(env, v, &t);
        if (t == napi_external) {
            pt_ctx* other = ctx_of(env, v); if (!other) return nullptr;
            int r2 = pt_image_dev_ptr(other, &p.image_dev);
            if (r2 < 0) return throw_pt(env, other, "pt_image_dev_ptr", r2);
        }
    }
    int rc = pt_render_init(c, &p);
    if (rc < 0) return throw_pt(env, c, "pt_render_init", rc);
    return nullptr;
}
// renderSetExtensions(ctx, maxNb, seed, Float32Array mats|null)              [pt_render_set_extensions]
// mats: 8 floats per extension material (ids 2, 3, ...): albedo r g b, type (0 Lambert, 1 emitter, 2 mirror, 3 glossy), emission r g b (glossy: roughness, 0, 0), 0
static napi_value RenderSetExtensions(napi_env env, napi_callback_info info) {
    napi_value a[4]; if (!get_args(env, info, 4, a)) return nullptr;
    pt_ctx* c = ctx_of(env, a[0]); if (!c) return nullptr;
    pt_render_ext e; memset(&e, 0, sizeof e);
    e.max_nb = (int32_t)i64(env, a[1]); e.seed = (uint32_t)i64(env, a[2]);
    void* d; size_t nb; if (!ta(env, a[3], &d, &nb)) return nullptr;
    if (nb % 32 || nb / 32 > PT_MAX_EXT_MATERIALS) { napi_throw_range_error(env, nullptr, "renderSetExtensions: 8 floats per material, at most 14 materials"); return nullptr; }
    e.n_materials = (int32_t)(nb / 32);
    const float* f = (const float*)d;
    for (int k = 0; k < e.n_materials; k++) {
        for (int j = 0; j < 3; j++) { e.materials[k].albedo[j] = f[8 * k + j]; e.materials[k].emission[j] = f[8 * k + 4 + j]; }
        e.materials[k].type = (int32_t)f[8 * k + 3];
    }
    int rc = pt_render_set_extensions(c, &e);
    if (rc < 0) return throw_pt(env, c, "pt_render_set_extensions", rc);
    return nullptr;
}
// renderReset(ctx)                                                          [pt_render_reset]
static napi_value RenderReset(napi_env env, napi_callback_info info) {
    napi_value a[1]; if (!get_args(env, info, 1, a)) return nullptr;
    pt_ctx* c = ctx_of(env, a[0]); if (!c) return nullptr;
    int rc = pt_render_reset(c);
    if (rc < 0) return throw_pt(env, c, "pt_render_reset", rc);
    return nullptr;
}
// renderStep(ctx) -> {camera, material, rays, step}                         [pt_render_step]
static napi_value RenderStep(napi_env env, napi_callback_info info) {
    napi_value a[1]; if (!get_args(env, info, 1, a)) return nullptr;
    pt_ctx* c = ctx_of(env, a[0]); if (!c) return nullptr;
    pt_step_stats st; int rc = pt_render_step(c, &st);
    if (rc < 0) return throw_pt(env, c, "pt_render_step", rc);
    napi_value o; napi_create_object(env, &o);
    set(env, o, "camera", (double)st.n_camera); set(env, o, "material", (double)st.n_material);
    set(env, o, "rays", (double)st.n_rays); set(env, o, "step", (double)st.step_index);
    return o;
}
// tolerance mode: what the exact re-walks of the sampled rays found (all 0 in the reference-exact mode)
static void set_check(napi_env env, napi_value o, const pt_run_stats& rs) {
    set(env, o, "checkedRays", (double)rs.checked_rays); set(env, o, "mismatchedRays", (double)rs.mismatched_rays); set(env, o, "flippedRays", (double)rs.flipped_rays);
    set(env, o, "maxRelDistDiff", rs.max_rel_dist_diff); set(env, o, "beyondToleranceRays", (double)rs.beyond_tolerance_rays); set(env, o, "lostHitRays", (double)rs.lost_hit_rays);
}
// setTraversal(ctx, mode, checkPeriod) -> mode in use once a scene is built, else -1     [pt_set_traversal_mode, pt_set_traversal_check, pt_get_traversal_mode]
static napi_value SetTraversal(napi_env env, napi_callback_info info) {
    napi_value a[3]; if (!get_args(env, info, 3, a)) return nullptr;
    pt_ctx* c = ctx_of(env, a[0]); if (!c) return nullptr;
    int rc = pt_set_traversal_mode(c, (int)i64(env, a[1]));
    if (rc < 0) return throw_pt(env, c, "pt_set_traversal_mode", rc);
    if (i64(env, a[2]) >= 0 && (rc = pt_set_traversal_check(c, (int)i64(env, a[2]))) < 0) return throw_pt(env, c, "pt_set_traversal_check", rc);
    int used = -1; (void)pt_get_traversal_mode(c, &used);
    napi_value v; napi_create_int32(env, used, &v);
    return v;
}
// renderRun(ctx, maxSteps, flags) -> stats                                  [pt_render_run]
static napi_value RenderRun(napi_env env, napi_callback_info info) {
    napi_value a[3]; if (!get_args(env, info, 3, a)) return nullptr;
    pt_ctx* c = ctx_of(env, a[0]); if (!c) return nullptr;
    pt_run_stats rs; int rc = pt_render_run(c, i64(env, a[1]), (int)i64(env, a[2]), &rs);
    if (rc < 0) return throw_pt(env, c, "pt_render_run", rc);
    napi_value o; napi_create_object(env, &o);
    set(env, o, "steps", (double)rs.steps); set(env, o, "totalRays", (double)rs.total_rays); set(env, o, "totalSamples", (double)rs.total_samples);
    set(env, o, "stackOverflows", (double)rs.stack_overflows); set(env, o, "extendMs", rs.extend_ms);
    set(env, o, "extendLaunches", (double)rs.extend_launches); set(env, o, "traversalBytes", (double)rs.traversal_bytes); set(env, o, "wastedRays", (double)rs.wasted_rays);
    set_check(env, o, rs);
    return o;
}
// renderRunMany([ctx, ...], maxSteps, flags) -> [stats, ...]                [pt_render_run_many]
static napi_value RenderRunMany(napi_env env, napi_callback_info info) {
    napi_value a[3]; if (!get_args(env, info, 3, a)) return nullptr;
    uint32_t n = 0; napi_get_array_length(env, a[0], &n);
    if (n < 1 || n > 64) { napi_throw_type_error(env, nullptr, "expected an array of 1..64 contexts"); return nullptr; }
    pt_ctx* cs[64]; pt_run_stats rs[64];
    for (uint32_t i = 0; i < n; i++) { napi_value e; napi_get_element(env, a[0], i, &e); cs[i] = ctx_of(env, e); if (!cs[i]) return nullptr; }
    int rc = pt_render_run_many(cs, (int)n, i64(env, a[1]), (int)i64(env, a[2]), rs);
    if (rc < 0) return throw_pt(env, cs[0], "pt_render_run_many", rc);
    napi_value arr; napi_create_array_with_length(env, n, &arr);
    for (uint32_t i = 0; i < n; i++) {
        napi_value o; napi_create_object(env, &o);
        set(env, o, "steps", (double)rs[i].steps); set(env, o, "totalRays", (double)rs[i].total_rays); set(env, o, "totalSamples", (double)rs[i].total_samples);
        set(env, o, "stackOverflows", (double)rs[i].stack_overflows); set(env, o, "extendMs", rs[i].extend_ms);
        set(env, o, "extendLaunches", (double)rs[i].extend_launches); set(env, o, "traversalBytes", (double)rs[i].traversal_bytes); set(env, o, "wastedRays", (double)rs[i].wasted_rays);
        set_check(env, o, rs[i]);
        napi_set_element(env, arr, i, o);
    }
    return arr;
}
// imageRead(ctx, Float32Array rgba)                                         [pt_image_read]
static napi_value ImageRead(napi_env env, napi_callback_info info) {
    napi_value a[2]; if (!get_args(env, info, 2, a)) return nullptr;
    pt_ctx* c = ctx_of(env, a[0]); if (!c) return nullptr;
    void* d; size_t nb; if (!ta(env, a[1], &d, &nb)) return nullptr;
    if (!d) { napi_throw_type_error(env, nullptr, "imageRead: expected a Float32Array"); return nullptr; }
    int rc = pt_image_read(c, (float*)d, (int64_t)(nb / 4));
    if (rc < 0) return throw_pt(env, c, "pt_image_read", rc);
    return nullptr;
}
// tonemap(ctx, Uint8Array rgba8)                                            [pt_tonemap_rgba8 on the ctx image]
static napi_value Tonemap(napi_env env, napi_callback_info info) {
    napi_value a[2]; if (!get_args(env, info, 2, a)) return nullptr;
    pt_ctx* c = ctx_of(env, a[0]); if (!c) return nullptr;
    void* d; size_t nb; if (!ta(env, a[1], &d, &nb)) return nullptr;
    int32_t w = 0, h = 0; int rs = pt_image_size(c, &w, &h);
    if (rs < 0) return throw_pt(env, c, "pt_image_size", rs);
    if (!d || nb != (size_t)w * (size_t)h * 4) { napi_throw_range_error(env, nullptr, "tonemap: the Uint8Array must hold width * height * 4 bytes of the current render"); return nullptr; }
    int rc = pt_tonemap_rgba8(c, nullptr, 0, 0, (uint8_t*)d);
    if (rc < 0) return throw_pt(env, c, "pt_tonemap_rgba8", rc);
    return nullptr;
}

// commCreate([ctx, ...]) -> comm handle: one rank per GPU, in array order                       [pt_comm_create]
static void finalize_comm(napi_env, void* data, void*) { pt_comm** h = (pt_comm**)data; if (*h) pt_comm_destroy(*h); delete h; }
static napi_value CommCreate(napi_env env, napi_callback_info info) {
    napi_value a[1]; if (!get_args(env, info, 1, a)) return nullptr;
    uint32_t n = 0; napi_get_array_length(env, a[0], &n);
    if (n < 1 || n > 64) { napi_throw_type_error(env, nullptr, "expected an array of 1..64 contexts"); return nullptr; }
    pt_ctx* cs[64];
    for (uint32_t i = 0; i < n; i++) { napi_value e; napi_get_element(env, a[0], i, &e); cs[i] = ctx_of(env, e); if (!cs[i]) return nullptr; }
    pt_comm* comm = nullptr; int rc = pt_comm_create(cs, (int)n, &comm);
    if (rc < 0) return throw_pt(env, cs[0], "pt_comm_create", rc);
    pt_comm** h = new pt_comm*(comm);
    napi_value ext; NAPI_OK(napi_create_external(env, h, finalize_comm, nullptr, &ext));
    return ext;
}
// commReduceImage(comm, root): sum of the ranks' accumulation buffers onto rank root's (RCCL)  [pt_comm_reduce_image]
static napi_value CommReduceImage(napi_env env, napi_callback_info info) {
    napi_value a[2]; if (!get_args(env, info, 2, a)) return nullptr;
    void* p = nullptr;
    if (napi_get_value_external(env, a[0], &p) != napi_ok || !p || !*(pt_comm**)p) { napi_throw_type_error(env, nullptr, "expected a live comm handle"); return nullptr; }
    int rc = pt_comm_reduce_image(*(pt_comm**)p, (int)i64(env, a[1]));
    if (rc < 0) { napi_throw_error(env, nullptr, ("pt_comm_reduce_image failed (" + std::to_string(rc) + ")").c_str()); return nullptr; }
    return nullptr;
}
// commDestroy(comm)                                                                            [pt_comm_destroy]
static napi_value CommDestroy(napi_env env, napi_callback_info info) {
    napi_value a[1]; if (!get_args(env, info, 1, a)) return nullptr;
    void* p = nullptr; napi_get_value_external(env, a[0], &p);
    if (p && *(pt_comm**)p) { pt_comm_destroy(*(pt_comm**)p); *(pt_comm**)p = nullptr; }
    return nullptr;
}

static napi_value Init(napi_env env, napi_value exports) {
    struct { const char* name; napi_callback fn; } fns[] = {
        {"create", Create}, {"createSharingScene", CreateSharingScene}, {"renderRunMany", RenderRunMany}, {"destroy", Destroy}, {"meshUpload", MeshUpload}, {"sceneBuild", SceneBuild}, {"sceneInfo", SceneInfo},
        {"readBuffer", ReadBuffer}, {"traceRays", TraceRays}, {"renderInit", RenderInit}, {"renderReset", RenderReset},
        {"renderSetExtensions", RenderSetExtensions}, {"setTraversal", SetTraversal},
        {"renderStep", RenderStep}, {"renderRun", RenderRun}, {"imageRead", ImageRead}, {"tonemap", Tonemap},
        {"commCreate", CommCreate}, {"commReduceImage", CommReduceImage}, {"commDestroy", CommDestroy}};
    for (auto& f : fns) {
        napi_value v; napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &v);
        napi_set_named_property(env, exports, f.name, v);
    }
    return exports;
}
NAPI_MODULE(pt_amd, Init)
