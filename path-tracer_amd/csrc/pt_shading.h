// pt_shading.h -- the per-slot bodies of the logic, camera and material kernels (logic.js:62-152, camera.js:60-118,
// material.js:75-120).  Each only reads and writes its own slot's PathState (plus the image), which is what lets the
// launches in pt_kernels_render.hip / pt_kernels_extend.hip call them from whichever thread holds the slot.
#pragma once
#include "pt_internal.h"
#include "pt_math.h"

#define ALBEDO0_POW 0.217637640824031f   // pow(0.5, 2.2): material.js:107 through :72
#define ALBEDO1_POW 0.0289911870530680f  // pow(0.2, 2.2): material.js:110 through :72

__device__ __forceinline__ f3 F3(P3 v) { return F3(v.x, v.y, v.z); }
__device__ __forceinline__ P3 mk3(float x, float y, float z) { P3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ P3 mk3(f3 v) { return mk3(v.x, v.y, v.z); }

// code bits: 1 = camera queue, 2 = material queue, 4 = wants a pixel ticket, 8 = finished a sample this step
#define PT_CODE_CAMERA 1u
#define PT_CODE_MATERIAL 2u
#define PT_CODE_TICKET 4u
#define PT_CODE_FINISHED 8u

// ============================================================================================
// generate (camera.js:60-118)
// ============================================================================================
// Runs in the extend kernel, for the entries of the camera queue (the ray goes straight into the traversal; the slot's
// origin and direction are still stored, the material sample of the next step reads them).
__device__ __forceinline__ void generate_one(const PathState& ps, const RenderConst& rc, int path, f3& ray_o, f3& ray_d) {
    int px = ps.pixel_index[path];
    float seed = ps.random_seed[path];
    float2 r2 = rand2(seed);
    ps.random_seed[path] = seed + 2.0f;
    float cx = (float)(px % rc.width) + r2.x, cy = (float)(px / rc.width) + r2.y;
    float sx = cx / (float)rc.width, sy = cy / (float)rc.height;
    sx = sx * 2.0f - 1.0f; sy = sy * 2.0f - 1.0f; sy *= -1.0f;
    // the camera frame is the same for every slot: camera_frame() below evaluates it once per render (same expressions)
    const CameraFrame& cf = rc.frame;
    float aspect = cf.aspect, sn = cf.fov_sin;
    f3 local = F3(aspect * sx * sn, 1.0f, sy * sn);
    f3 pos = F3(rc.cam_pos[0], rc.cam_pos[1], rc.cam_pos[2]);
    f3 fwd = F3(cf.fwd[0], cf.fwd[1], cf.fwd[2]), right = F3(cf.right[0], cf.right[1], cf.right[2]), up = F3(cf.up[0], cf.up[1], cf.up[2]);
    f3 w = normalize3(local);
    f3 d = (right * w.x + fwd * w.y) + up * w.z;
    ps.path_o[path] = mk3(pos);
    ps.path_d[path] = mk3(d);
    ray_o = pos; ray_d = d;
}

// camera.js:96-111, the part that does not depend on the pixel
__device__ __forceinline__ CameraFrame camera_frame(const RenderConst& rc) {
    CameraFrame cf;
    cf.aspect = (float)rc.width / (float)rc.height;
    float sn, cs; pt_sincos(rc.fov_deg * 3.14159265358979323846f / 180.0f, sn, cs);
    cf.fov_sin = sn;
    f3 pos = F3(rc.cam_pos[0], rc.cam_pos[1], rc.cam_pos[2]);
    f3 fwd = normalize3(F3(rc.cam_look_at[0], rc.cam_look_at[1], rc.cam_look_at[2]) - pos);
    f3 right = normalize3(F3(fwd.y, -fwd.x, 0.0f));
    f3 up = cross3(right, fwd);
    cf.fwd[0] = fwd.x; cf.fwd[1] = fwd.y; cf.fwd[2] = fwd.z;
    cf.right[0] = right.x; cf.right[1] = right.y; cf.right[2] = right.z;
    cf.up[0] = up.x; cf.up[1] = up.y; cf.up[2] = up.z;
    return cf;
}

// ============================================================================================
// shade (material.js:75-120; scene.js:204-258 get_triangle_hit_info)
// ============================================================================================
__device__ __forceinline__ void shade_one(const PathState& ps, const SceneView& sc, int path, int hit_obj, int hit_tri) {
    f3 o = F3(ps.path_o[path]), d = F3(ps.path_d[path]);
    const float4* ob = sc.objects + 8 * (int64_t)hit_obj;
    f3 lo, ld; int mat;
    {
        float4 inf = ob[3], w0 = ob[4], w1 = ob[5], w2 = ob[6];
        lo = xform(w0, w1, w2, o, 1.0f); ld = xform(w0, w1, w2, d, 0.0f);
        mat = __float_as_int(inf.y);
    }
    f3 ln, lhit;
    {
        const float4* tr = sc.tris + 3 * (int64_t)hit_tri;
        float4 t0 = tr[0], t1 = tr[1], t2 = tr[2];
        f3 v0 = F3(t0), v0v1 = F3(t1), v0v2 = F3(t2);             // SceneView::tris: edges and unit normal are stored
        f3 pvec = cross3(ld, v0v2);
        float det = dot3(v0v1, pvec), i_det = 1.0f / det;
        f3 tvec = lo - v0;
        f3 qvec = cross3(tvec, v0v1);
        ln = F3(t0.w, t1.w, t2.w);
        float ldist = dot3(v0v2, qvec) * i_det;
        lhit = lo + ld * ldist;
    }
    float4 l0 = ob[0], l1 = ob[1], l2 = ob[2];
    f3 whit = xform(l0, l1, l2, lhit, 1.0f);
    f3 n = normalize3(xform(l0, l1, l2, ln, 0.0f));
    float dist = length3(whit - o);
    f3 hit_pos = o + d * dist;
    f3 ov = __builtin_fabsf(n.x) > __builtin_fabsf(n.y) ? F3(-n.y, n.x, 0.0f) : F3(0.0f, -n.z, n.y);   // material.js:166-171
    f3 o1 = normalize3(ov), o2 = normalize3(cross3(o1, n));
    float seed = ps.random_seed[path];
    float4 bp = make_float4(0.0f, 0.0f, 0.0f, 0.0f); f3 wi = F3(0.0f, 0.0f, 0.0f);
    if (mat == 0 || mat == 1) {                                                  // material.js:106-111
        float2 r2 = rand2(seed); seed += 2.0f;
        float rad = __builtin_sqrtf(fmax_(r2.x, 0.0f)), sn, cs;                    // material.js:141-151
        pt_sincos(6.283185307179586f * r2.y, sn, cs);
        float dx = rad * cs, dy = rad * sn;
        float z = __builtin_sqrtf(fmax_(0.0f, (1.0f - dx * dx) - dy * dy));
        wi = F3(dx, dy, z);
        float a = mat == 0 ? ALBEDO0_POW : ALBEDO1_POW;
        bp = make_float4(a, a, a, wi.z);
    }
    f3 nd = (o1 * wi.x + o2 * wi.y) + n * wi.z;
    unsigned flag = 1u; f3 off_n = n;
    if (mat >= 2 && mat - 2 < sc.n_ext_mats) {                                  // extension materials: rules defined in oracle/pt_oracle.h
        float4 m0 = sc.ext_mats[2 * (mat - 2)], m1 = sc.ext_mats[2 * (mat - 2) + 1];
        int type = __float_as_int(m0.w);
        if (type == 0) {                                                        // Lambert, rgb albedo
            float2 r2 = rand2(seed); seed += 2.0f;
            float rad = __builtin_sqrtf(fmax_(r2.x, 0.0f)), sn, cs;
            pt_sincos(6.283185307179586f * r2.y, sn, cs);
            float dx = rad * cs, dy = rad * sn;
            float z = __builtin_sqrtf(fmax_(0.0f, (1.0f - dx * dx) - dy * dy));
            wi = F3(dx, dy, z);
            bp = make_float4(m0.x, m0.y, m0.z, wi.z);
            nd = (o1 * wi.x + o2 * wi.y) + n * wi.z;
        } else if (type == 1) {                                                 // emitter
            bp = make_float4(m1.x, m1.y, m1.z, 0.0f); nd = n; flag = 3u;
        } else if (type == 2) {                                                 // mirror
            float k = 2.0f * dot3(d, n);
            nd = d - n * k;
            bp = make_float4(m0.x, m0.y, m0.z, 1.0f);
        } else if (type == 3) {                                                 // glossy: a mirror blurred by roughness = emission[0]
            float2 r2 = rand2(seed); seed += 2.0f;
            float rad = __builtin_sqrtf(fmax_(r2.x, 0.0f)), sn, cs;
            pt_sincos(6.283185307179586f * r2.y, sn, cs);
            float dx = rad * cs, dy = rad * sn;
            float z = __builtin_sqrtf(fmax_(0.0f, (1.0f - dx * dx) - dy * dy));
            float dn = dot3(d, n);
            f3 ns = dn < 0.0f ? n : F3(-n.x, -n.y, -n.z);
            f3 refl = d - n * (2.0f * dn);
            f3 dd = (o1 * dx + o2 * dy) + ns * z;
            float rough = fmin_(fmax_(m1.x, 0.0f), 1.0f), keep = 1.0f - rough;
            nd = normalize3(refl * keep + dd * rough);
            bp = make_float4(m0.x, m0.y, m0.z, 1.0f);
            off_n = ns;
        }
    }
    f3 no = hit_pos + off_n * 0.0001f;
    ps.material_throughput_pdf[path] = bp;
    ps.state[path] |= flag;
    ps.random_seed[path] = seed;
    ps.path_o[path] = mk3(no);
    ps.path_d[path] = mk3(nd);
}

// ============================================================================================
// logic (logic.js:62-152) for a slot whose ray has been traced: state machine, image accumulation, decision.
// Image accumulation: a slot owns its pixel for the 8 samples of a ticket (logic.js:6,129), so the samples are
// summed in the slot's ticket accumulator and reach the image with ONE group of float atomics per ticket instead
// of one per sample (scattered float atomics are the slowest memory operation on this chip).  The reference's
// own add order is a race (logic.js:222-233); sample counts stay exact.
// ============================================================================================
__device__ __forceinline__ unsigned logic_one(const PathState& ps, float* __restrict__ image, int i, bool b_hit, int max_nb) {
    unsigned code = 0; bool finished = false;
    const uint32_t st = ps.state[i];
    if ((st & PT_ST_RETIRED) != 0u) return 0u;                                   // logic.js:66 a slot past the last ticket
    f3  tp = F3(ps.path_throughput[i]);
    int nb = (int)PT_ST_NB(st) + 1, sip = (int)PT_ST_SIP(st);
    uint32_t flags = PT_ST_FLAGS(st);
    f3 contrib = F3(0.0f, 0.0f, 0.0f);
    if (flags & 2u) {                                                   // extension: the ray hit an emitter (pt_amd.h pt_material)
        contrib = tp * F3(ps.material_throughput_pdf[i]); tp = F3(0.0f, 0.0f, 0.0f);
    } else if (flags & 1u) {                                            // logic.js:94-109
        tp = tp * F3(ps.material_throughput_pdf[i]);
        if (nb > 3) {
            float seed = ps.random_seed[i];
            float2 r2 = rand2(seed);
            ps.random_seed[i] = seed + 2.0f;
            float q = fmin_(fmax_(0.1f, 1.0f - tp.y), 0.7f);
            if (r2.x < q) tp = F3(0.0f, 0.0f, 0.0f); else tp = tp / (1.0f - q);
        }
    }
    if (nb > max_nb) tp = F3(0.0f, 0.0f, 0.0f);                          // logic.js:111-113 (max_nb = 20)
    if (!b_hit) { contrib = F3(contrib.x + 2.0f * tp.x, contrib.y + 2.0f * tp.y, contrib.z + 2.0f * tp.z); tp = F3(0.0f, 0.0f, 0.0f); }
    if (tp.x == 0.0f && tp.y == 0.0f && tp.z == 0.0f) {                  // logic.js:121-133
        finished = true; code |= 1u; nb = 0; tp = F3(1.0f, 1.0f, 1.0f);
        if (sip >= PT_SAMPLES_PER_TICKET - 1) code |= 4u; else sip += 1;
    } else { code |= 2u; nb += 1; }
    ps.path_throughput[i] = mk3(tp);
    ps.state[i] = PT_ST_PACK(0u, sip, nb);                               // flags = 0 (logic.js:142)
    if (finished) {                                                      // logic.js:145-150
        float4 acc = ps.ticket_acc[i];
        acc.x += contrib.x; acc.y += contrib.y; acc.z += contrib.z; acc.w += 1.0f;
        if (code & 4u) {                                                 // ticket complete: flush to the image
            float* px = image + 4 * (int64_t)ps.pixel_index[i];
            unsafeAtomicAdd(px + 0, acc.x); unsafeAtomicAdd(px + 1, acc.y);
            unsafeAtomicAdd(px + 2, acc.z); unsafeAtomicAdd(px + 3, acc.w);
            acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        ps.ticket_acc[i] = acc;
    }
    return code | (finished ? PT_CODE_FINISHED : 0u);
}
