/*
 * pt_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see pt_oracle.h for the contract
 * and the "parity unpinned" statement).  Every function cites the reference file:line it
 * restates (paths relative to /root/reference).  Nothing here is shipped or called by
 * the product path.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 */
#include "pt_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * vector helpers: the arithmetic contract of pt_oracle.h
 * ---------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3    V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3    v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3    v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3    v3_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3    v3_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3    v3_divs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3    v3_cross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }
static inline v3    v3_normalize(v3 a) { return v3_divs(a, v3_length(a)); }
static inline v3    v3_min(v3 a, v3 b) { return V3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
static inline v3    v3_max(v3 a, v3 b) { return V3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
static inline v3    v3_ld(const float* p) { return V3(p[0], p[1], p[2]); }

/* rows[3][4] * (v, w): the WGSL mat4x4f(col0..col3) * vec4f built at scene.js:208-216,337-345 */
static inline v3 xform(const float m[3][4], v3 v, float w) {
    return V3(((m[0][0] * v.x + m[0][1] * v.y) + m[0][2] * v.z) + m[0][3] * w,
              ((m[1][0] * v.x + m[1][1] * v.y) + m[1][2] * v.z) + m[1][3] * w,
              ((m[2][0] * v.x + m[2][1] * v.y) + m[2][2] * v.z) + m[2][3] * w);
}

uint32_t orc_f32_to_u32(float x) { /* WGSL u32(f32): truncate, saturate; NaN defined as 0 */
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}

/* ------------------------------------------------------------------------------------------
 * math spec: sin/cos
 * ---------------------------------------------------------------------------------------- */
void orc_sincos(float theta, float* s_out, float* c_out) {
    /* valid for theta in [0, 8): quadrant by truncation, 3-term Cody-Waite, Cephes polynomials */
    int   k  = (int)(theta * 0.636619772367581343f + 0.5f);
    float kf = (float)k;
    float r  = ((theta - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
    float z  = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    switch (k & 3) {
        case 0:  *s_out = ps;  *c_out = pc;  break;
        case 1:  *s_out = pc;  *c_out = -ps; break;
        case 2:  *s_out = -ps; *c_out = -pc; break;
        default: *s_out = -pc; *c_out = ps;  break;
    }
}

/* material.js:154-158 (= logic.js:235-239 = camera.js:124-128) */
uint32_t orc_base_hash(uint32_t px, uint32_t py) {
    uint32_t p2x = 1103515245u * ((px >> 1) ^ py);
    uint32_t p2y = 1103515245u * ((py >> 1) ^ px);
    uint32_t h32 = 1103515245u * (p2x ^ (p2y >> 3));
    return h32 ^ (h32 >> 16);
}

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* material.js:159-163: f32(0x7fffffff) rounds to 2^31 */
void orc_rand2(float seed, float* r0, float* r1) {
    uint32_t n  = orc_base_hash(f2u(seed + 1.0f), f2u(seed + 2.0f));
    uint32_t rx = n & 0x7fffffffu, ry = (n * 48271u) & 0x7fffffffu;
    *r0 = (float)rx / 2147483648.0f;
    *r1 = (float)ry / 2147483648.0f;
}

/* ------------------------------------------------------------------------------------------
 * LBVH build
 * ---------------------------------------------------------------------------------------- */
/* aabb-z-idx.js:202-209 */
static inline uint32_t split_3(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
/* aabb-z-idx.js:191-198 */
static uint32_t morton_of_box(v3 mn, v3 mx, v3 bmin, v3 bmax) {
    v3 cen = v3_scale(v3_add(mx, mn), 0.5f);
    v3 num = v3_sub(cen, bmin), den = v3_sub(bmax, bmin);
    v3 rel = V3(num.x / den.x, num.y / den.y, num.z / den.z);
    uint32_t ux = orc_f32_to_u32(rel.x * 1023.99f), uy = orc_f32_to_u32(rel.y * 1023.99f),
             uz = orc_f32_to_u32(rel.z * 1023.99f);
    return split_3(ux) | (split_3(uy) << 1) | (split_3(uz) << 2);
}

/* aabb-z-idx.js:176-195 */
void orc_tri_aabb_morton(const orc_tri* tris, int64_t n, const float bmin[3], const float bmax[3],
                         orc_aabb* aabb_out, uint32_t* key_out) {
    v3 lo = v3_ld(bmin), hi = v3_ld(bmax);
    for (int64_t i = 0; i < n; i++) {
        v3 a = v3_ld(tris[i].v0), b = v3_ld(tris[i].v1), c = v3_ld(tris[i].v2);
        v3 mn = v3_min(a, v3_min(b, c)), mx = v3_max(a, v3_max(b, c));
        orc_aabb bx = { { mn.x, mn.y, mn.z }, 0.0f, { mx.x, mx.y, mx.z }, 0.0f };
        aabb_out[i] = bx;
        key_out[i]  = morton_of_box(mn, mx, lo, hi);
    }
}
/* aabb2z-idx.js:157-170 */
void orc_obj_morton(const orc_aabb* aabbs, int64_t n, const float bmin[3], const float bmax[3], uint32_t* key_out) {
    v3 lo = v3_ld(bmin), hi = v3_ld(bmax);
    for (int64_t i = 0; i < n; i++) key_out[i] = morton_of_box(v3_ld(aabbs[i].mn), v3_ld(aabbs[i].mx), lo, hi);
}

/* radix-sort.js:162-500: net effect = stable sort by the full 32-bit key carrying idx[i]=i.
 * Restated as a 4-pass byte LSD counting sort (stable, so ties keep ascending original index). */
void orc_stable_sort(uint32_t* keys, int32_t* idx_out, int64_t n) {
    uint32_t* k2 = (uint32_t*)malloc((size_t)n * 4 + 4);
    int32_t*  i2 = (int32_t*)malloc((size_t)n * 4 + 4);
    for (int64_t i = 0; i < n; i++) idx_out[i] = (int32_t)i;
    uint32_t* ks = keys; uint32_t* kd = k2; int32_t* is = idx_out; int32_t* id = i2;
    for (int pass = 0; pass < 4; pass++) {
        int64_t cnt[257]; memset(cnt, 0, sizeof cnt);
        int sh = 8 * pass;
        for (int64_t i = 0; i < n; i++) cnt[((ks[i] >> sh) & 255u) + 1]++;
        for (int b = 0; b < 256; b++) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < n; i++) { int64_t p = cnt[(ks[i] >> sh) & 255u]++; kd[p] = ks[i]; id[p] = is[i]; }
        uint32_t* tk = ks; ks = kd; kd = tk; int32_t* ti = is; is = id; id = ti;
    }
    /* 4 passes: result is back in keys / idx_out */
    free(k2); free(i2);
}

/* radix-tree.js:160-185.  clz(k1^k2) == dif(); equal keys fall back to 32 + dif(indices) */
static inline int dif32(uint32_t a, uint32_t b) { uint32_t x = a ^ b; return x ? __builtin_clz(x) : -1; }
static inline int del_(const uint32_t* keys, int64_t n, int64_t i1, int64_t i2) {
    if (i1 >= n || i2 >= n || i1 < 0 || i2 < 0) return -1;
    int kd = dif32(keys[i1], keys[i2]);
    if (kd == -1) kd = 32 + dif32((uint32_t)i1, (uint32_t)i2);
    return kd;
}
/* radix-tree.js:188-238 compute_child_index; the split search is restated as Karras' loop
 * (t = ceil(l/2), ceil(l/4), ..., 1), which is the reference's t sequence without its
 * overflow-terminated tail of no-op t=1 iterations (SURVEY.md "Restatement traps"). */
static void child_index(const uint32_t* keys, int64_t n, int64_t i, int32_t* cl, int32_t* cr) {
    int a = del_(keys, n, i, i + 1) - del_(keys, n, i, i - 1);
    int64_t d = (a > 0) - (a < 0);
    int del_min = del_(keys, n, i, i - d);
    int64_t l_max = 2;
    while (del_(keys, n, i, i + l_max * d) > del_min) l_max *= 2;
    int64_t l = 0;
    for (int64_t t = l_max / 2; t > 0; t /= 2)
        if (del_(keys, n, i, i + (l + t) * d) > del_min) l += t;
    int64_t j = i + l * d;
    int del_node = del_(keys, n, i, j);
    int64_t s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (del_(keys, n, i, i + (s + t) * d) > del_node) s += t;
    } while (t > 1);
    int64_t gamma = i + s * d + (d < 0 ? d : 0);
    int64_t lo = i < j ? i : j, hi = i < j ? j : i;
    *cl = (int32_t)(lo == gamma ? -gamma - 1 : gamma);
    *cr = (int32_t)(hi == gamma + 1 ? -(gamma + 1) - 1 : gamma + 1);
}
/* radix-tree.js:136-157 */
void orc_radix_tree(const uint32_t* keys, int64_t n, int32_t* parents) {
    memset(parents, 0, (size_t)(2 * n) * 4);
    for (int64_t i = 0; i < n - 1; i++) {
        int32_t cl, cr;
        child_index(keys, n, i, &cl, &cr);
        if (cl >= 0) parents[cl] = (int32_t)i; else parents[n + -(cl + 1)] = (int32_t)i;
        if (cr >= 0) parents[cr] = (int32_t)i; else parents[n + -(cr + 1)] = (int32_t)i;
    }
}

/* bvh-up-pass.js:199-280, executed serially thread 0..n-1, each to completion: the canonical
 * order (first arrival = Karras left child).  f_1 keeps its arrival count (excluded from parity). */
void orc_up_pass(const int32_t* idx, const orc_aabb* aabb, const int32_t* par, int64_t n, orc_node* nodes) {
    memset(nodes, 0, (size_t)n * sizeof(orc_node));
    for (int64_t t = 0; t < n; t++) {
        v3 mn = v3_ld(aabb[idx[t]].mn), mx = v3_ld(aabb[idx[t]].mx);
        /* :209-210 */
        mn.x -= (mn.x == 0.0f ? 1.0f : 0.0f) * 1e-8f; mn.y -= (mn.y == 0.0f ? 1.0f : 0.0f) * 1e-8f;
        mn.z -= (mn.z == 0.0f ? 1.0f : 0.0f) * 1e-8f;
        mx.x += (mx.x == 0.0f ? 1.0f : 0.0f) * 1e-8f; mx.y += (mx.y == 0.0f ? 1.0f : 0.0f) * 1e-8f;
        mx.z += (mx.z == 0.0f ? 1.0f : 0.0f) * 1e-8f;
        int64_t c_idx = t; int32_t w_idx = (int32_t)(-(t + 1)); int level = 0; int skipped = 0;
        while ((w_idx != 0 || level == 0) && !skipped) {
            int32_t   p_idx = level == 0 ? par[c_idx + n] : par[c_idx];
            orc_node* nd = &nodes[p_idx];
            int32_t   sibling = nd->f_1++;
            if (sibling == 0) {
                nd->lmin[0] = mn.x; nd->lmin[1] = mn.y; nd->lmin[2] = mn.z;
                nd->lmax[0] = mx.x; nd->lmax[1] = mx.y; nd->lmax[2] = mx.z;
                nd->l_child = w_idx;
                skipped = 1;
            } else {
                nd->rmin[0] = mn.x; nd->rmin[1] = mn.y; nd->rmin[2] = mn.z;
                nd->rmax[0] = mx.x; nd->rmax[1] = mx.y; nd->rmax[2] = mx.z;
                nd->r_child = w_idx;
                v3 lmn = v3_ld(nd->lmin), lmx = v3_ld(nd->lmax);
                if (lmn.x == 0.0f || lmn.y == 0.0f || lmn.z == 0.0f || lmx.x == 0.0f || lmx.y == 0.0f || lmx.z == 0.0f) {
                    fprintf(stderr, "orc_up_pass: serial order would spin at node %d\n", p_idx); abort();
                }
                mn = v3_min(mn, lmn); mx = v3_max(mx, lmx);
                c_idx = p_idx; w_idx = p_idx; level += 1;
            }
        }
    }
}

/* builders.js:174-192 */
void orc_build_bvh(const orc_aabb* aabb, uint32_t* keys, int64_t n, int32_t* idx_out, orc_node* nodes_out) {
    int32_t* parents = (int32_t*)malloc((size_t)(2 * n) * 4 + 8);
    orc_stable_sort(keys, idx_out, n);
    orc_radix_tree(keys, n, parents);
    orc_up_pass(idx_out, aabb, parents, n, nodes_out);
    free(parents);
}

/* ------------------------------------------------------------------------------------------
 * scene
 * ---------------------------------------------------------------------------------------- */
typedef struct { orc_tri* tris; int64_t n; float bounds[6]; int built; orc_node* nodes; orc_tri* sorted;
                 int32_t* perm; int64_t tri_offset; int used; } orc_mesh;

struct orc_scene {
    orc_mesh* meshes; int n_meshes, cap_meshes;
    int64_t n_tris; orc_node* blas_nodes; orc_tri* blas_tris;
    int n_objects; orc_node* tlas_nodes; orc_object* objects; int32_t* tlas_perm;
};

orc_scene* orc_scene_create(void) { return (orc_scene*)calloc(1, sizeof(orc_scene)); }
static void scene_free_built(orc_scene* s) {
    free(s->blas_nodes); free(s->blas_tris); free(s->tlas_nodes); free(s->objects); free(s->tlas_perm);
    s->blas_nodes = NULL; s->blas_tris = NULL; s->tlas_nodes = NULL; s->objects = NULL; s->tlas_perm = NULL;
}
void orc_scene_destroy(orc_scene* s) {
    if (!s) return;
    for (int i = 0; i < s->n_meshes; i++) { free(s->meshes[i].tris); free(s->meshes[i].nodes); free(s->meshes[i].sorted); free(s->meshes[i].perm); }
    free(s->meshes); scene_free_built(s); free(s);
}
int orc_scene_add_mesh(orc_scene* s, const float* tris12, int64_t n, const float bounds[6]) {
    if (s->n_meshes == s->cap_meshes) {
        s->cap_meshes = s->cap_meshes ? 2 * s->cap_meshes : 8;
        s->meshes = (orc_mesh*)realloc(s->meshes, (size_t)s->cap_meshes * sizeof(orc_mesh));
    }
    orc_mesh* m = &s->meshes[s->n_meshes];
    memset(m, 0, sizeof *m);
    m->n = n; m->tris = (orc_tri*)malloc((size_t)n * sizeof(orc_tri) + 16);
    memcpy(m->tris, tris12, (size_t)n * sizeof(orc_tri));
    memcpy(m->bounds, bounds, 24);
    return s->n_meshes++;
}

/* builders.js:153-214 buildMeshBVH */
static void build_mesh(orc_mesh* m) {
    int64_t n = m->n;
    orc_aabb* aabb = (orc_aabb*)malloc((size_t)n * sizeof(orc_aabb) + 16);
    uint32_t* keys = (uint32_t*)malloc((size_t)n * 4 + 4);
    m->perm = (int32_t*)malloc((size_t)n * 4 + 4);
    m->nodes = (orc_node*)malloc((size_t)n * sizeof(orc_node) + 16);
    m->sorted = (orc_tri*)malloc((size_t)n * sizeof(orc_tri) + 16);
    orc_tri_aabb_morton(m->tris, n, m->bounds, m->bounds + 3, aabb, keys);
    orc_build_bvh(aabb, keys, n, m->perm, m->nodes);
    for (int64_t i = 0; i < n; i++) m->sorted[i] = m->tris[m->perm[i]]; /* rearrange.js:162-169 */
    free(aabb); free(keys);
    m->built = 1;
}

/* scene.js:15-73 */
int orc_scene_build(orc_scene* s, const orc_instance* inst, int n_inst) {
    scene_free_built(s);
    if (n_inst <= 0) return -1;
    for (int i = 0; i < s->n_meshes; i++) s->meshes[i].used = 0;
    for (int i = 0; i < n_inst; i++) {
        if (inst[i].mesh < 0 || inst[i].mesh >= s->n_meshes) return -2;
        s->meshes[inst[i].mesh].used = 1;
    }
    /* :17-29 utilizedMeshOrder: ascending mesh id, running triangle offset; BLAS cached per mesh */
    int64_t off = 0;
    for (int i = 0; i < s->n_meshes; i++) {
        orc_mesh* m = &s->meshes[i];
        if (!m->used) continue;
        m->tri_offset = off;
        if (!m->built) build_mesh(m);
        off += m->n;
    }
    s->n_tris = off;
    /* pack-t-mesh.js:218-245 (all n node slots of each mesh are rebased, the unused last one too) */
    s->blas_nodes = (orc_node*)malloc((size_t)off * sizeof(orc_node) + 16);
    s->blas_tris  = (orc_tri*)malloc((size_t)off * sizeof(orc_tri) + 16);
    for (int i = 0; i < s->n_meshes; i++) {
        orc_mesh* m = &s->meshes[i];
        if (!m->used) continue;
        int32_t o = (int32_t)m->tri_offset;
        for (int64_t k = 0; k < m->n; k++) {
            s->blas_tris[o + k] = m->sorted[k];
            orc_node nd = m->nodes[k];
            if (nd.l_child < 0) nd.l_child -= o; else nd.l_child += o;
            if (nd.r_child < 0) nd.r_child -= o; else nd.r_child += o;
            s->blas_nodes[o + k] = nd;
        }
    }
    /* scene.js:42-61 scene bounds; builders.js:97-151 buildTLAS */
    float bmin[3] = { 1e30f, 1e30f, 1e30f }, bmax[3] = { -1e30f, -1e30f, -1e30f };
    orc_aabb* aabbs = (orc_aabb*)calloc((size_t)n_inst, sizeof(orc_aabb));
    for (int i = 0; i < n_inst; i++)
        for (int c = 0; c < 3; c++) {
            aabbs[i].mn[c] = inst[i].aabb_min[c]; aabbs[i].mx[c] = inst[i].aabb_max[c];
            bmin[c] = fminf(bmin[c], inst[i].aabb_min[c]); bmax[c] = fmaxf(bmax[c], inst[i].aabb_max[c]);
        }
    uint32_t* keys = (uint32_t*)malloc((size_t)n_inst * 4);
    s->tlas_perm  = (int32_t*)malloc((size_t)n_inst * 4);
    s->tlas_nodes = (orc_node*)malloc((size_t)n_inst * sizeof(orc_node));
    orc_obj_morton(aabbs, n_inst, bmin, bmax, keys);
    orc_build_bvh(aabbs, keys, n_inst, s->tlas_perm, s->tlas_nodes);
    free(keys); free(aabbs);
    /* builders.js:25-67 packObjects (TLAS-sorted order) */
    s->n_objects = n_inst;
    s->objects = (orc_object*)calloc((size_t)n_inst, sizeof(orc_object));
    for (int i = 0; i < n_inst; i++) {
        const orc_instance* in = &inst[s->tlas_perm[i]];
        orc_object* ob = &s->objects[i];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) { ob->l2w[r][c] = in->l2w[4 * c + r]; ob->w2l[r][c] = in->w2l[4 * c + r]; }
        ob->bvh_offset = (int32_t)s->meshes[in->mesh].tri_offset;
        ob->material = in->material;
    }
    return 0;
}
int64_t orc_scene_num_tris(const orc_scene* s) { return s->n_tris; }
int     orc_scene_num_objects(const orc_scene* s) { return s->n_objects; }
const orc_node*   orc_scene_blas_nodes(const orc_scene* s) { return s->blas_nodes; }
const orc_tri*    orc_scene_blas_tris(const orc_scene* s) { return s->blas_tris; }
const orc_node*   orc_scene_tlas_nodes(const orc_scene* s) { return s->tlas_nodes; }
const orc_object* orc_scene_objects(const orc_scene* s) { return s->objects; }
const int32_t*    orc_scene_tlas_perm(const orc_scene* s) { return s->tlas_perm; }
const int32_t*    orc_scene_mesh_perm(const orc_scene* s, int mesh) {
    return (mesh >= 0 && mesh < s->n_meshes) ? s->meshes[mesh].perm : NULL;
}

/* ------------------------------------------------------------------------------------------
 * traversal
 * ---------------------------------------------------------------------------------------- */
/* scene.js:508-515 */
static inline float aabb_intersect(v3 low, v3 high, v3 o, v3 d) {
    v3 id = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    v3 f = v3_mul(v3_sub(high, o), id), n = v3_mul(v3_sub(low, o), id);
    v3 tmax = v3_max(f, n), tmin = v3_min(f, n);
    float t0 = fmaxf(tmin.x, fmaxf(tmin.y, tmin.z));
    float t1 = fminf(tmax.x, fminf(tmax.y, tmax.z));
    return (t1 >= t0) ? ((t1 < 0.0f) ? -1e30f : t0) : -1e30f;
}
/* scene.js:478-506 */
static inline float tri_intersect(v3 o, v3 d, const orc_tri* tr) {
    v3 v0 = v3_ld(tr->v0);
    v3 v0v1 = v3_sub(v3_ld(tr->v1), v0), v0v2 = v3_sub(v3_ld(tr->v2), v0);
    v3 pvec = v3_cross(d, v0v2);
    float det = v3_dot(v0v1, pvec);
    if (fabsf(det) < 1e-10f) return -1.0f;
    float i_det = 1.0f / det;
    v3 tvec = v3_sub(o, v0);
    float u = v3_dot(tvec, pvec) * i_det;
    if (u < 0.0f || u > 1.0f) return -1.0f;
    v3 qvec = v3_cross(tvec, v0v1);
    float v = v3_dot(d, qvec) * i_det;
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    return v3_dot(v0v2, qvec) * i_det;
}

typedef struct { int32_t hit_obj, hit_tri; float hit_dis; int32_t n_inner, n_leaf, n_enter, n_leave; int overflow; int max_sp; } hit_res;

/* scene.js:330-355 / :357-379: map o and the point o+d*hit_dist, renormalise d, recompute hit_dist */
static inline void switch_space(const float m[3][4], v3* o, v3* d, float* hit_dist) {
    v3 hit_pos = v3_add(*o, v3_scale(*d, *hit_dist));
    v3 new_o = xform(m, *o, 1.0f), new_h = xform(m, hit_pos, 1.0f);
    v3 nd = v3_sub(new_h, new_o);
    *hit_dist = v3_length(nd);
    *d = v3_divs(nd, *hit_dist);
    *o = new_o;
}

/* scene.js:313-475 intersect_bvh.  Differences, both guards the WGSL gets from its runtime:
 * stack has ORC_STACK_SIZE entries (reference 32) and a push onto a full stack ends the traversal
 * with the hit found so far (WGSL: out-of-bounds store is clamped/discarded, then garbage). */
/* rec (optional, tools/ only): one byte per trip through the loop -- 1 inner node, 2 triangle, 4 object entered, +8 when the
 * trip began by leaving an object; a final lone 8 is the leave after the loop.  At most cap bytes; *n_rec gets the true count. */
static hit_res intersect_bvh_rec(const orc_scene* s, v3 o, v3 d, uint8_t* rec, int cap, int* n_rec) {
    int32_t stack[ORC_STACK_SIZE]; int nr = 0;
    float hit_dist = 1e6f; int32_t hit_obj = -1, hit_tri = -1;
    int32_t stack_ptr = 0, node_idx = 0, obj_idx = -1, switch_pt = -1;
    hit_res R; memset(&R, 0, sizeof R);
    int64_t iter = 0;
    while (stack_ptr >= 0) {
        /* guard: the degenerate N=1 tree (r_child = 0 self pointer, SURVEY "Degenerate N=1") can loop */
        if (++iter > ORC_MAX_ITERS) { R.overflow = 1; break; }
        uint8_t kind = 0;
        if (stack_ptr < switch_pt) {
            switch_space(s->objects[obj_idx].l2w, &o, &d, &hit_dist);
            switch_pt = -1; obj_idx = -1; R.n_leave++; kind = 8;
        }
        kind |= node_idx >= 0 ? 1 : (obj_idx < 0 ? 4 : 2);
        if (rec) { if (nr < cap) rec[nr] = kind; nr++; }
        if (node_idx < 0) {
            if (obj_idx < 0) {
                obj_idx = -(node_idx + 1);
                const orc_object* ob = &s->objects[obj_idx];
                node_idx = ob->bvh_offset; switch_pt = stack_ptr;
                switch_space(ob->w2l, &o, &d, &hit_dist);
                R.n_enter++;
            } else {
                float dist = tri_intersect(o, d, &s->blas_tris[-(node_idx + 1)]);
                if (dist > 0.0f && dist < hit_dist) { hit_dist = dist; hit_tri = -(node_idx + 1); hit_obj = obj_idx; }
                R.n_leaf++;
                stack_ptr -= 1;
                node_idx = stack_ptr >= 0 ? stack[stack_ptr] : 0;
            }
        } else {
            const orc_node* nd = obj_idx < 0 ? &s->tlas_nodes[node_idx] : &s->blas_nodes[node_idx];
            R.n_inner++;
            float l_dist = aabb_intersect(v3_ld(nd->lmin), v3_ld(nd->lmax), o, d);
            float r_dist = aabb_intersect(v3_ld(nd->rmin), v3_ld(nd->rmax), o, d);
            int l_valid = l_dist != -1e30f && l_dist < hit_dist;
            int r_valid = r_dist != -1e30f && r_dist < hit_dist;
            if (l_valid && r_valid) {
                int32_t f_idx, c_idx;
                if (l_dist < r_dist) { c_idx = nd->l_child; f_idx = nd->r_child; }
                else                 { c_idx = nd->r_child; f_idx = nd->l_child; }
                if (stack_ptr >= ORC_STACK_SIZE) { R.overflow = 1; break; }
                stack[stack_ptr] = f_idx; stack_ptr += 1; node_idx = c_idx;
                if (stack_ptr > R.max_sp) R.max_sp = stack_ptr;
            } else if (l_valid) node_idx = nd->l_child;
            else if (r_valid)   node_idx = nd->r_child;
            else { stack_ptr -= 1; node_idx = stack_ptr >= 0 ? stack[stack_ptr] : 0; }
        }
    }
    if (obj_idx >= 0) { /* :442-466 */
        switch_space(s->objects[obj_idx].l2w, &o, &d, &hit_dist);
        R.n_leave++;
        if (rec) { if (nr < cap) rec[nr] = 8; nr++; }
    }
    if (n_rec) *n_rec = nr;
    R.hit_obj = hit_obj; R.hit_tri = hit_tri; R.hit_dis = hit_dist;
    return R;
}
static hit_res intersect_bvh(const orc_scene* s, v3 o, v3 d) { return intersect_bvh_rec(s, o, d, NULL, 0, NULL); }
/* the kind of every trip of every ray (see intersect_bvh_rec): kinds[i * cap ..], len[i] = trips of ray i (may exceed cap) */
void orc_trace_kinds(const orc_scene* s, const float* o3, const float* d3, int64_t n, uint8_t* kinds, int cap, int32_t* len, int threads) {
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
    for (int64_t i = 0; i < n; i++) {
        int nr = 0;
        intersect_bvh_rec(s, v3_ld(o3 + 3 * i), v3_ld(d3 + 3 * i), kinds + i * (int64_t)cap, cap, &nr);
        len[i] = nr;
    }
}

static int g_last_max_depth = 0;   /* deepest stack of the last orc_trace_rays call (tests: is the spill exercised?) */
int64_t orc_trace_rays(const orc_scene* s, const float* o3, const float* d3, int64_t n, int32_t* hit_obj,
                       int32_t* hit_tri, float* hit_dist, int32_t* counts4, int threads) {
    int64_t overflow = 0; int max_sp = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads) reduction(+ : overflow) reduction(max : max_sp)
    for (int64_t i = 0; i < n; i++) {
        hit_res r = intersect_bvh(s, v3_ld(o3 + 3 * i), v3_ld(d3 + 3 * i));
        if (r.max_sp > max_sp) max_sp = r.max_sp;
        if (hit_obj) hit_obj[i] = r.hit_obj;
        if (hit_tri) hit_tri[i] = r.hit_tri;
        if (hit_dist) hit_dist[i] = r.hit_dis;
        if (counts4) { counts4[4 * i] = r.n_inner; counts4[4 * i + 1] = r.n_leaf; counts4[4 * i + 2] = r.n_enter; counts4[4 * i + 3] = r.n_leave; }
        overflow += r.overflow;
    }
    g_last_max_depth = max_sp;
    return overflow;
}
int orc_trace_max_depth(void) { return g_last_max_depth; }

/* Independent pin (no BVH, no hit_dist round trips): for every object in TLAS-sorted order map
 * the ray to object space exactly as an "enter" from a fresh ray would, test every triangle of
 * its mesh, and compare hits by WORLD distance |l2w(local hit) - o|.  Not bit-identical to the
 * traversal by construction (no carried hit_dist), so tests compare with a tolerance on distance
 * and exact index agreement away from ties. */
void orc_brute_force(const orc_scene* s, const float* o3, const float* d3, int64_t n, int32_t* hit_obj,
                     int32_t* hit_tri, float* hit_t, int threads) {
    if (threads < 1) threads = 1;
    /* mesh extent per object: [bvh_offset, next offset) */
    int64_t* ends = (int64_t*)malloc((size_t)s->n_objects * 8);
    for (int k = 0; k < s->n_objects; k++) {
        int64_t e = s->n_tris;
        for (int m = 0; m < s->n_meshes; m++)
            if (s->meshes[m].used && s->meshes[m].tri_offset == s->objects[k].bvh_offset) e = s->meshes[m].tri_offset + s->meshes[m].n;
        ends[k] = e;
    }
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
    for (int64_t i = 0; i < n; i++) {
        v3 o = v3_ld(o3 + 3 * i), d = v3_ld(d3 + 3 * i);
        double best = 1e300; int32_t bo = -1, bt = -1;
        for (int k = 0; k < s->n_objects; k++) {
            const orc_object* ob = &s->objects[k];
            v3 lo = o, ld = d; float hd = 1e6f;
            switch_space(ob->w2l, &lo, &ld, &hd);
            for (int64_t t = ob->bvh_offset; t < ends[k]; t++) {
                float dist = tri_intersect(lo, ld, &s->blas_tris[t]);
                if (!(dist > 0.0f)) continue;
                v3 lh = v3_add(lo, v3_scale(ld, dist));
                v3 wh = xform(ob->l2w, lh, 1.0f);
                double dx = (double)wh.x - o.x, dy = (double)wh.y - o.y, dz = (double)wh.z - o.z;
                double wd = sqrt(dx * dx + dy * dy + dz * dz);
                if (wd < best) { best = wd; bo = k; bt = (int32_t)t; }
            }
        }
        hit_obj[i] = bo; hit_tri[i] = bt; if (hit_t) hit_t[i] = bo >= 0 ? (float)best : 1e6f;
    }
    free(ends);
}

/* ------------------------------------------------------------------------------------------
 * render: path-tracer.js:49-123 step(); logic.js, camera.js, material.js, ray-trace.js kernels
 * ---------------------------------------------------------------------------------------- */
#define SAMPLES_PER_PIXEL_PER_PATH 8 /* logic.js:6 */
#define ALBEDO0_POW 0.217637640824031f /* pow(0.5, 2.2) material.js:107,72 */
#define ALBEDO1_POW 0.0289911870530680f /* pow(0.2, 2.2) material.js:110,72 */

struct orc_render {
    const orc_scene* scene; orc_render_params p; int64_t num_steps, num_logic, n_live;
    int64_t pixel_counter, ticket_limit;
    int32_t *pixel_index, *num_bounces, *hit_obj, *hit_tri, *samples_in_pixel; uint32_t* flags;
    float *random_seed, *path_throughput, *material_throughput_pdf, *path_o, *path_d;
    float* image; float* contrib; uint8_t* code;
    int32_t *camera_queue, *material_queue; int64_t n_camera, n_material;
    int64_t total_rays, traversal_bytes;
    orc_render_ext ext;
};

orc_render* orc_render_create(const orc_scene* s, const orc_render_params* p) {
    orc_render* r = (orc_render*)calloc(1, sizeof *r);
    r->scene = s; r->p = *p;
    size_t n = (size_t)p->num_paths;
    /* WebGPU zero-fills: relied on by num_bounces / flags / samples_in_pixel */
    r->pixel_index = calloc(n, 4); r->num_bounces = calloc(n, 4); r->hit_obj = calloc(n, 4); r->hit_tri = calloc(n, 4);
    r->samples_in_pixel = calloc(n, 4); r->flags = calloc(n, 4); r->random_seed = calloc(n, 4);
    r->path_throughput = calloc(n, 16); r->material_throughput_pdf = calloc(n, 16); r->path_o = calloc(n, 16); r->path_d = calloc(n, 16);
    r->image = calloc((size_t)p->width * p->height, 16); r->contrib = calloc(n, 16); r->code = calloc(n, 1);
    r->camera_queue = calloc(n, 4); r->material_queue = calloc(n, 4);
    /* logic.js:177 ticket limit; multi-GPU split gives each rank [ticket_begin, ticket_end) */
    int64_t total = ((int64_t)p->width * p->height * p->spp + SAMPLES_PER_PIXEL_PER_PATH - 1) / SAMPLES_PER_PIXEL_PER_PATH;
    r->pixel_counter = p->ticket_end >= 0 ? p->ticket_begin : 0;
    r->ticket_limit  = p->ticket_end >= 0 ? p->ticket_end : total;
    return r;
}
void orc_render_set_extensions(orc_render* r, const orc_render_ext* e) { r->ext = *e; }
void orc_render_destroy(orc_render* r) {
    if (!r) return;
    free(r->pixel_index); free(r->num_bounces); free(r->hit_obj); free(r->hit_tri); free(r->samples_in_pixel); free(r->flags);
    free(r->random_seed); free(r->path_throughput); free(r->material_throughput_pdf); free(r->path_o); free(r->path_d);
    free(r->image); free(r->contrib); free(r->code); free(r->camera_queue); free(r->material_queue); free(r);
}

/* logic.js:62-152 for one slot.  Writes the slot's image contribution to contrib[] and its
 * decisions to code[] (bit0 camera queue, bit1 material queue, bit2 wants a pixel ticket). */
static void logic_slot(orc_render* r, int64_t i, int first_sample) {
    uint8_t code = 0; float* c = r->contrib + 4 * i; c[0] = c[1] = c[2] = c[3] = 0.0f;
    if ((r->flags[i] & (1u << 31)) != 0u) { r->code[i] = 0; return; }
    if (first_sample) {
        uint32_t gi = (uint32_t)(r->p.slot_base + i) + r->ext.seed * 0x9E3779B1u;   /* ext: user seed (0 = reference) */
        r->random_seed[i] = (float)orc_base_hash(gi, gi + 1u) / 4294967296.0f + 0.008f; /* :77 */
        r->path_throughput[4 * i] = r->path_throughput[4 * i + 1] = r->path_throughput[4 * i + 2] = 1.0f;
        code = 1 | 4;
    } else {
        v3 tp = v3_ld(r->path_throughput + 4 * i);
        int32_t nb = r->num_bounces[i] + 1;
        int b_hit = r->hit_obj[i] >= 0;
        uint32_t flags = r->flags[i];
        if (flags & 2u) {                /* ext: the ray hit an emitter -- add throughput * emission, end the path */
            v3 e = v3_mul(tp, v3_ld(r->material_throughput_pdf + 4 * i));
            c[0] += e.x; c[1] += e.y; c[2] += e.z; tp = V3(0, 0, 0);
        } else if (flags & 1u) {
            tp = v3_mul(tp, v3_ld(r->material_throughput_pdf + 4 * i));
            if (nb > 3) {
                float r0, r1; orc_rand2(r->random_seed[i], &r0, &r1); (void)r1;
                r->random_seed[i] += 2.0f;
                float q = fminf(fmaxf(0.1f, 1.0f - tp.y), 0.7f);
                if (r0 < q) tp = V3(0, 0, 0); else tp = v3_divs(tp, 1.0f - q);
            }
        }
        if (nb > (r->ext.max_nb > 0 ? r->ext.max_nb : 20)) tp = V3(0, 0, 0);   /* :111; ext: max_nb */
        if (!b_hit) { c[0] += 2.0f * tp.x; c[1] += 2.0f * tp.y; c[2] += 2.0f * tp.z; tp = V3(0, 0, 0); }
        if (tp.x == 0.0f && tp.y == 0.0f && tp.z == 0.0f) {
            c[3] = 1.0f; code |= 1; nb = 0; tp = V3(1, 1, 1);
            if (r->samples_in_pixel[i] >= SAMPLES_PER_PIXEL_PER_PATH - 1) code |= 4; else r->samples_in_pixel[i] += 1;
        } else { code |= 2; nb += 1; }
        r->path_throughput[4 * i] = tp.x; r->path_throughput[4 * i + 1] = tp.y; r->path_throughput[4 * i + 2] = tp.z;
        r->num_bounces[i] = nb; r->flags[i] = 0u;
    }
    r->code[i] = code;
}

/* camera.js:60-118 for one queued path */
static void camera_slot(orc_render* r, int32_t path) {
    const orc_render_params* p = &r->p;
    int32_t px = r->pixel_index[path];
    float r0, r1; orc_rand2(r->random_seed[path], &r0, &r1);
    float cx = (float)(px % p->width) + r0, cy = (float)(px / p->width) + r1;
    r->random_seed[path] += 2.0f;
    float sx = cx / (float)p->width, sy = cy / (float)p->height;
    sx = sx * 2.0f - 1.0f; sy = sy * 2.0f - 1.0f; sy *= -1.0f;
    float aspect = (float)p->width / (float)p->height;
    float sn, cs; orc_sincos(p->fov_deg * 3.14159265358979323846f / 180.0f, &sn, &cs); (void)cs;
    v3 local = V3(aspect * sx * sn, 1.0f, sy * sn);
    v3 pos = v3_ld(p->cam_pos);
    v3 fwd = v3_normalize(v3_sub(v3_ld(p->cam_look_at), pos));
    v3 right = v3_normalize(V3(fwd.y, -fwd.x, 0.0f));
    v3 up = v3_cross(right, fwd);
    v3 w = v3_normalize(local);
    v3 d = v3_add(v3_add(v3_scale(right, w.x), v3_scale(fwd, w.y)), v3_scale(up, w.z));
    float* po = r->path_o + 4 * path; float* pd = r->path_d + 4 * path;
    po[0] = pos.x; po[1] = pos.y; po[2] = pos.z; pd[0] = d.x; pd[1] = d.y; pd[2] = d.z;
}

/* scene.js:239-258 helper_intersect_triangle (no rejection tests) + :204-237 get_triangle_hit_info */
static void triangle_hit_info(const orc_scene* s, v3 o, v3 d, int32_t hit_obj, int32_t hit_tri, v3* normal, float* dist) {
    const orc_object* ob = &s->objects[hit_obj];
    v3 lo = xform(ob->w2l, o, 1.0f), ld = xform(ob->w2l, d, 0.0f);
    const orc_tri* tr = &s->blas_tris[hit_tri];
    v3 v0 = v3_ld(tr->v0), v0v1 = v3_sub(v3_ld(tr->v1), v0), v0v2 = v3_sub(v3_ld(tr->v2), v0);
    v3 pvec = v3_cross(ld, v0v2);
    float det = v3_dot(v0v1, pvec), i_det = 1.0f / det;
    v3 tvec = v3_sub(lo, v0);
    v3 qvec = v3_cross(tvec, v0v1);
    v3 ln = v3_normalize(v3_cross(v0v1, v0v2));
    float ldist = v3_dot(v0v2, qvec) * i_det;
    v3 lhit = v3_add(lo, v3_scale(ld, ldist));
    v3 whit = xform(ob->l2w, lhit, 1.0f), wn = xform(ob->l2w, ln, 0.0f);
    *normal = v3_normalize(wn);
    *dist = v3_length(v3_sub(whit, o));
}

/* material.js:75-120 for one queued path */
static void material_slot(orc_render* r, int32_t path) {
    const orc_scene* s = r->scene;
    v3 o = v3_ld(r->path_o + 4 * path), d = v3_ld(r->path_d + 4 * path);
    int32_t hit_obj = r->hit_obj[path], hit_tri = r->hit_tri[path];
    v3 n; float dist; triangle_hit_info(s, o, d, hit_obj, hit_tri, &n, &dist);
    v3 hit_pos = v3_add(o, v3_scale(d, dist));
    /* :166-171 ortho */
    v3 ov = fabsf(n.x) > fabsf(n.y) ? V3(-n.y, n.x, 0.0f) : V3(0.0f, -n.z, n.y);
    v3 o1 = v3_normalize(ov), o2 = v3_normalize(v3_cross(o1, n));
    float seed = r->random_seed[path];
    float bp[4] = { 0, 0, 0, 0 }; v3 wi = V3(0, 0, 0);
    int32_t mat = s->objects[hit_obj].material;
    if (mat == 0 || mat == 1) {
        float r0, r1; orc_rand2(seed, &r0, &r1); seed += 2.0f;
        /* :141-151 cosineSampleHemisphere / uniformSampleDisk */
        float rad = sqrtf(fmaxf(r0, 0.0f)), theta = 6.283185307179586f * r1, sn, cs;
        orc_sincos(theta, &sn, &cs);
        float dx = rad * cs, dy = rad * sn;
        float z = sqrtf(fmaxf(0.0f, (1.0f - dx * dx) - dy * dy));
        wi = V3(dx, dy, z);
        float a = mat == 0 ? ALBEDO0_POW : ALBEDO1_POW;
        bp[0] = bp[1] = bp[2] = a; bp[3] = wi.z;
    }
    v3 nd = v3_add(v3_add(v3_scale(o1, wi.x), v3_scale(o2, wi.y)), v3_scale(n, wi.z));
    uint32_t flag = 1u; v3 off_n = n;
    if (mat >= 2 && mat - 2 < r->ext.n_materials) {              /* ext materials (pt_oracle.h); nothing of the reference below */
        const orc_material* m = &r->ext.materials[mat - 2];
        if (m->type == ORC_MAT_LAMBERT) {
            float r0, r1; orc_rand2(seed, &r0, &r1); seed += 2.0f;
            float rad = sqrtf(fmaxf(r0, 0.0f)), theta = 6.283185307179586f * r1, sn, cs;
            orc_sincos(theta, &sn, &cs);
            float dx = rad * cs, dy = rad * sn;
            float z = sqrtf(fmaxf(0.0f, (1.0f - dx * dx) - dy * dy));
            wi = V3(dx, dy, z);
            bp[0] = m->albedo[0]; bp[1] = m->albedo[1]; bp[2] = m->albedo[2]; bp[3] = wi.z;
            nd = v3_add(v3_add(v3_scale(o1, wi.x), v3_scale(o2, wi.y)), v3_scale(n, wi.z));
        } else if (m->type == ORC_MAT_EMITTER) {
            bp[0] = m->emission[0]; bp[1] = m->emission[1]; bp[2] = m->emission[2]; bp[3] = 0.0f;
            nd = n; flag = 3u;                                   /* the ray is traced but its result is ignored */
        } else if (m->type == ORC_MAT_MIRROR) {
            float k = 2.0f * v3_dot(d, n);
            nd = v3_sub(d, v3_scale(n, k));
            bp[0] = m->albedo[0]; bp[1] = m->albedo[1]; bp[2] = m->albedo[2]; bp[3] = 1.0f;
        } else if (m->type == ORC_MAT_GLOSSY) {                  /* rule: pt_oracle.h */
            float r0, r1; orc_rand2(seed, &r0, &r1); seed += 2.0f;
            float rad = sqrtf(fmaxf(r0, 0.0f)), theta = 6.283185307179586f * r1, sn, cs;
            orc_sincos(theta, &sn, &cs);
            float dx = rad * cs, dy = rad * sn;
            float z = sqrtf(fmaxf(0.0f, (1.0f - dx * dx) - dy * dy));
            float dn = v3_dot(d, n);
            v3 ns = dn < 0.0f ? n : V3(-n.x, -n.y, -n.z);
            v3 refl = v3_sub(d, v3_scale(n, 2.0f * dn));
            v3 dd = v3_add(v3_add(v3_scale(o1, dx), v3_scale(o2, dy)), v3_scale(ns, z));
            float rough = fminf(fmaxf(m->emission[0], 0.0f), 1.0f), keep = 1.0f - rough;
            nd = v3_normalize(v3_add(v3_scale(refl, keep), v3_scale(dd, rough)));
            bp[0] = m->albedo[0]; bp[1] = m->albedo[1]; bp[2] = m->albedo[2]; bp[3] = 1.0f;
            off_n = ns;
        }
    }
    memcpy(r->material_throughput_pdf + 4 * path, bp, 16);
    r->flags[path] |= flag;
    r->random_seed[path] = seed;
    v3 no = v3_add(hit_pos, v3_scale(off_n, 0.0001f));           /* material.js:118 */
    float* po = r->path_o + 4 * path; float* pd = r->path_d + 4 * path;
    po[0] = no.x; po[1] = no.y; po[2] = no.z; pd[0] = nd.x; pd[1] = nd.y; pd[2] = nd.z;
}

/* One step() of the reference is four dispatches: logic, camera, material, ray-trace (path-tracer.js:49-123).  They are
 * kept as separately callable phases because the camera and the material phase touch disjoint slots (code bit 0 / bit 1)
 * and therefore commute: an implementation may run "material" right behind "logic" for the same slot and hand out
 * tickets / camera rays afterwards.  orc_render_step runs them in the reference's order. */
static void phase_logic(orc_render* r, int threads) {           /* logic.js:62-152 + image adds :145-150,222-233 */
    int64_t n = r->p.num_paths; int first = r->num_logic == 0;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < n; i++) logic_slot(r, i, first);
    for (int64_t i = 0; i < n; i++) {                           /* canonical serial order of the adds */
        const float* c = r->contrib + 4 * i;
        if (c[0] != 0.0f || c[1] != 0.0f || c[2] != 0.0f || c[3] != 0.0f) {
            float* px = r->image + 4 * (int64_t)r->pixel_index[i];
            px[0] += c[0]; px[1] += c[1]; px[2] += c[2]; px[3] += c[3];
        }
    }
    r->num_logic++;
}
static void phase_queues_camera(orc_render* r, int threads) {   /* logic.js:155-219 (tickets, queues), camera.js */
    int64_t n = r->p.num_paths, npix = (int64_t)r->p.width * r->p.height, n_live = 0;
    r->n_camera = r->n_material = 0;
    for (int64_t i = 0; i < n; i++) {                           /* canonical serial order: ascending slot */
        uint8_t code = r->code[i];
        if (code & 4) {
            int64_t ticket = r->pixel_counter++;
            if (ticket >= r->ticket_limit) r->flags[i] |= (1u << 31);
            else { r->pixel_index[i] = (int32_t)(ticket % npix); r->samples_in_pixel[i] = 0; }
        }
        if (code & 1) r->camera_queue[r->n_camera++] = (int32_t)i;
        if (code & 2) r->material_queue[r->n_material++] = (int32_t)i;
        if (!(r->flags[i] & (1u << 31))) n_live++;
    }
    r->n_live = n_live;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t q = 0; q < r->n_camera; q++) camera_slot(r, r->camera_queue[q]);
}
static void phase_material(orc_render* r, int threads) {        /* material.js over the slots logic sent on (code bit 1) */
    int64_t n = r->p.num_paths;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < n; i++) if (r->code[i] & 2) material_slot(r, (int32_t)i);
}
static void phase_trace(orc_render* r, int threads) {           /* ray-trace.js:53-69 over camera ++ material */
    int64_t n_rays = r->n_camera + r->n_material, bytes = 0;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads) reduction(+ : bytes)
    for (int64_t q = 0; q < n_rays; q++) {
        int32_t path = q < r->n_camera ? r->camera_queue[q] : r->material_queue[q - r->n_camera];
        hit_res h = intersect_bvh(r->scene, v3_ld(r->path_o + 4 * path), v3_ld(r->path_d + 4 * path));
        r->hit_obj[path] = h.hit_obj; r->hit_tri[path] = h.hit_tri;
        bytes += 44 + 64 * (int64_t)h.n_inner + 48 * (int64_t)h.n_leaf + 128 * (int64_t)(h.n_enter + h.n_leave);
    }
    r->total_rays += n_rays; r->traversal_bytes += bytes; r->num_steps++;
}
void orc_render_phase(orc_render* r, int phase, orc_step_stats* out, int threads) {
    if (threads < 1) threads = 1;
    if (phase == ORC_PHASE_LOGIC) phase_logic(r, threads);
    else if (phase == ORC_PHASE_QUEUES_CAMERA) phase_queues_camera(r, threads);
    else if (phase == ORC_PHASE_MATERIAL) phase_material(r, threads);
    else if (phase == ORC_PHASE_TRACE) phase_trace(r, threads);
    if (out) { out->n_camera = r->n_camera; out->n_material = r->n_material; out->n_rays = r->n_camera + r->n_material; out->n_live = r->n_live; }
}
void orc_render_step(orc_render* r, orc_step_stats* out, int threads) {
    orc_render_phase(r, ORC_PHASE_LOGIC, NULL, threads);
    orc_render_phase(r, ORC_PHASE_QUEUES_CAMERA, NULL, threads);
    orc_render_phase(r, ORC_PHASE_MATERIAL, NULL, threads);
    orc_render_phase(r, ORC_PHASE_TRACE, out, threads);
}

int64_t orc_render_run(orc_render* r, int64_t max_steps, int64_t* total_rays, int64_t* traversal_bytes, int threads) {
    int64_t steps = 0; orc_step_stats st;
    while (steps < max_steps) {
        orc_render_step(r, &st, threads); steps++;
        if (st.n_camera + st.n_material == 0) break;
    }
    if (total_rays) *total_rays = r->total_rays;
    if (traversal_bytes) *traversal_bytes = r->traversal_bytes;
    return steps;
}
const float* orc_render_image(const orc_render* r) { return r->image; }
const void* orc_render_field(const orc_render* r, const char* name, size_t* bytes) {
    size_t n = (size_t)r->p.num_paths;
#define F(nm, sz) if (!strcmp(name, #nm)) { if (bytes) *bytes = n * (sz); return r->nm; }
    F(pixel_index, 4) F(num_bounces, 4) F(random_seed, 4) F(path_throughput, 16) F(material_throughput_pdf, 16)
    F(path_o, 16) F(path_d, 16) F(hit_obj, 4) F(hit_tri, 4) F(flags, 4) F(samples_in_pixel, 4)
#undef F
    if (!strcmp(name, "pixel_counter")) { if (bytes) *bytes = 8; return &r->pixel_counter; }
    return NULL;
}
const int32_t* orc_render_queue(const orc_render* r, int which, int64_t* n) {
    if (which == 0) { *n = r->n_camera; return r->camera_queue; }
    *n = r->n_material; return r->material_queue;
}

/* display.js:98-113; canvas unorm8 store = round-to-nearest of clamp(x,0,1)*255, NaN -> 0 */
void orc_tonemap_rgba8(const float* image, int64_t n_pixels, uint8_t* out) {
    for (int64_t i = 0; i < n_pixels; i++) {
        const float* p = image + 4 * i;
        float c[3] = { p[0] / p[3], p[1] / p[3], p[2] / p[3] };
        float lum = (c[0] * 0.2126f + c[1] * 0.7152f) + c[2] * 0.0722f;
        for (int k = 0; k < 3; k++) {
            float v = powf(c[k] / (1.0f + lum), 1.0f / 2.2f);
            if (!(v > 0.0f)) v = 0.0f;
            if (v > 1.0f) v = 1.0f;
            out[4 * i + k] = (uint8_t)(v * 255.0f + 0.5f);
        }
        out[4 * i + 3] = 255;
    }
}
