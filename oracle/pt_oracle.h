/*
 * pt_oracle.h -- CPU oracle for the path-tracer hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C, strict-IEEE-754 binary32 restatement of the reference's WGSL/JS
 * algorithms (file:line citations are relative to /root/reference and sit on every
 * function in pt_oracle.c).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product (path-tracer_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" for every device-side row -- the reference ships no
 * tests, golden vectors or fixtures and its WGSL cannot execute here (no WebGPU
 * adapter).  The host-side JS (parse-obj.js, transform-utils.js) IS pinned: fixtures in
 * tests/golden/ were produced by running those two reference files under Node 12
 * (tests/golden/make_host_fixtures.js).  Device-side pins are self-made: brute force,
 * std-sort agreement, tree invariants, analytic radiance (see tests/).
 *
 * Arithmetic contract (shared, as a specification, with the HIP kernels):
 *   - binary32, round-to-nearest-even, no FMA contraction (-ffp-contract=off),
 *     correctly rounded / and sqrt, subnormals kept;
 *   - dot(a,b)   = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   - cross(a,b) = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
 *   - length(v)  = sqrt(dot(v,v));  normalize(v) = v / length(v) (three divides)
 *   - M*(v,w)    = ((r.x*v.x + r.y*v.y) + r.z*v.z) + r.w*w   per matrix row r
 *   - min/max    = fminf/fmaxf (a NaN operand is ignored)
 *   - f32->u32   = truncation, saturating, NaN -> 0
 *   - sin/cos    = pt_sincos below (Cody-Waite by pi/2 + Cephes minimax polynomials);
 *     WGSL leaves sin/cos/pow precision to the backend, so any accurate definition
 *     conforms; this one is chosen so that CPU and GPU agree bit for bit.
 *   - pow(albedo,2.2) for the two constant albedos = correctly rounded constants.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_STACK_SIZE 64 /* reference: 32 (scene.js:305); see DESIGN.md "stack" */
#define ORC_MAX_ITERS (1 << 20) /* traversal loop guard (reference: none) */

typedef struct { float lmin[3]; int32_t l_child; float lmax[3]; int32_t f_1;
                 float rmin[3]; int32_t r_child; float rmax[3]; int32_t f_2; } orc_node;   /* scene.js:265-274 */
typedef struct { float v0[3], p0, v1[3], p1, v2[3], p2; } orc_tri;                          /* scene.js:294-298 */
typedef struct { float l2w[3][4]; int32_t bvh_offset, material, f_2, f_3;
                 float w2l[3][4]; int32_t f_4, f_5, f_6, f_7; } orc_object;                /* scene.js:277-292 */
typedef struct { float mn[3], p0, mx[3], p1; } orc_aabb;                                    /* aabb-z-idx.js:159-162 */

/* instance as the JS host hands it over (column-major 4x4, already rounded to f32) */
typedef struct { float l2w[16]; float w2l[16]; float aabb_min[3]; float aabb_max[3];
                 int32_t mesh; int32_t material; } orc_instance;

/* ---- LBVH build stages (each separately testable) ---- */
void orc_tri_aabb_morton(const orc_tri* tris, int64_t n, const float bmin[3], const float bmax[3],
                         orc_aabb* aabb_out, uint32_t* key_out);
void orc_obj_morton(const orc_aabb* aabbs, int64_t n, const float bmin[3], const float bmax[3], uint32_t* key_out);
void orc_stable_sort(uint32_t* keys, int32_t* idx_out, int64_t n);
void orc_radix_tree(const uint32_t* sorted_keys, int64_t n, int32_t* parents /* 2n, zero-filled here */);
void orc_up_pass(const int32_t* idx, const orc_aabb* aabb, const int32_t* parents, int64_t n, orc_node* nodes /* n */);
void orc_build_bvh(const orc_aabb* aabb, uint32_t* keys /* in: unsorted, out: sorted */, int64_t n,
                   int32_t* idx_out, orc_node* nodes_out);

/* ---- scene ---- */
typedef struct orc_scene orc_scene;
orc_scene* orc_scene_create(void);
void       orc_scene_destroy(orc_scene*);
/* scene.js:518 registerMesh (+ parse result: builders.js:155) */
int  orc_scene_add_mesh(orc_scene*, const float* tris12, int64_t n_tris, const float bounds[6]);
/* scene.js:15-73 build() */
int  orc_scene_build(orc_scene*, const orc_instance* inst, int n_inst);
int64_t orc_scene_num_tris(const orc_scene*);
int     orc_scene_num_objects(const orc_scene*);
const orc_node*   orc_scene_blas_nodes(const orc_scene*);
const orc_tri*    orc_scene_blas_tris(const orc_scene*);
const orc_node*   orc_scene_tlas_nodes(const orc_scene*);
const orc_object* orc_scene_objects(const orc_scene*);
const int32_t*    orc_scene_tlas_perm(const orc_scene*);
const int32_t*    orc_scene_mesh_perm(const orc_scene*, int mesh);   /* per-mesh sorted->original */

/* ---- traversal ---- */
/* counts[4] per ray (optional): n_inner, n_leaf, n_enter, n_leave ; returns #stack overflows */
int64_t orc_trace_rays(const orc_scene*, const float* o3, const float* d3, int64_t n,
                       int32_t* hit_obj, int32_t* hit_tri, float* hit_dist, int32_t* counts4, int threads);
int orc_trace_max_depth(void);   /* deepest traversal stack seen by the last orc_trace_rays call */
/* tools/ only (wave-scheduling models): the kind of every trip of every ray's walk -- 1 inner node, 2 triangle, 4 object entered,
 * +8 when the trip began by leaving an object, a final lone 8 = the leave after the loop.  kinds: n x cap bytes; len[i] may exceed cap. */
void orc_trace_kinds(const orc_scene*, const float* o3, const float* d3, int64_t n, uint8_t* kinds, int cap, int32_t* len, int threads);
/* independent pin: test every instance x every triangle, no BVH. ties -> first in (obj,tri) order */
void orc_brute_force(const orc_scene*, const float* o3, const float* d3, int64_t n,
                     int32_t* hit_obj, int32_t* hit_tri, float* hit_t_world, int threads);

/* ---- render (wavefront integrator) ---- */
typedef struct {
    int32_t width, height;
    float   cam_pos[3], cam_look_at[3], fov_deg;
    int32_t spp;
    int32_t num_paths;          /* path-tracer.js:4 (1,000,000) */
    int64_t slot_base;          /* multi-GPU: global slot index of local slot 0 */
    int64_t ticket_begin, ticket_end; /* multi-GPU: this rank's share of pixel tickets; end<0 => all */
} orc_render_params;

typedef struct { int64_t n_camera, n_material, n_rays, n_live; } orc_step_stats;

/* ---- extensions: things the configs ask for that the reference does not have.  All default-off: with
 * max_nb = 0, seed = 0 and no extension materials every result is the reference's.  There is nothing in the reference
 * to follow here; the oracle DEFINES these rules and the HIP kernels must reproduce them.
 *   max_nb   replaces the constant of logic.js:111 (num_bounces advances by 2 per bounce there; 0 -> 20)
 *   seed     added (times 0x9E3779B1, mod 2^32) to the global slot index hashed at logic.js:77
 *   materials[k] describes material id k + 2 (ids 0 and 1 stay the reference's two grey Lambert albedos, material.js:106-111;
 *   an id with no entry keeps the reference's behaviour for unknown ids: zero BRDF, the path dies)
 *     type 0 Lambert, rgb albedo already in linear space (same cosine sampling as the reference)
 *     type 1 emitter: the path ends at the hit and adds throughput * emission
 *     type 2 mirror: wi = d - 2 (d.n) n, throughput *= albedo
 *     type 3 glossy: a mirror blurred by `roughness` = emission[0] clamped to [0, 1] (0 = the mirror, 1 = the Lambert lobe):
 *            ns = the normal turned towards the incoming ray; dd = a cosine-sampled direction about ns (rand2(seed), seed += 2, the
 *            reference's sampler and basis); wi = normalize(reflect(d, n) * (1 - roughness) + dd * roughness); throughput *= albedo;
 *            the new origin is offset along ns (every other material keeps the reference's offset along n, material.js:118) */
enum { ORC_MAT_LAMBERT = 0, ORC_MAT_EMITTER = 1, ORC_MAT_MIRROR = 2, ORC_MAT_GLOSSY = 3, ORC_MAX_EXT_MATERIALS = 14 };
typedef struct { float albedo[3]; int32_t type; float emission[3]; int32_t pad; } orc_material;       /* 32 B */
typedef struct { int32_t max_nb; uint32_t seed; int32_t n_materials, pad; orc_material materials[ORC_MAX_EXT_MATERIALS]; } orc_render_ext;

typedef struct orc_render orc_render;
orc_render* orc_render_create(const orc_scene*, const orc_render_params*);
void        orc_render_destroy(orc_render*);
void        orc_render_set_extensions(orc_render*, const orc_render_ext*);   /* before the first step */
void        orc_render_step(orc_render*, orc_step_stats* out, int threads);
/* the four dispatches of a step, one at a time (step = LOGIC, QUEUES_CAMERA, MATERIAL, TRACE; QUEUES_CAMERA and MATERIAL
 * touch disjoint slots and may be swapped) */
enum { ORC_PHASE_LOGIC = 0, ORC_PHASE_QUEUES_CAMERA = 1, ORC_PHASE_MATERIAL = 2, ORC_PHASE_TRACE = 3 };
void        orc_render_phase(orc_render*, int phase, orc_step_stats* out, int threads);
/* run until both stage-2 queues are empty; returns number of steps */
int64_t     orc_render_run(orc_render*, int64_t max_steps, int64_t* total_rays, int64_t* traversal_bytes, int threads);
const float*    orc_render_image(const orc_render*);             /* W*H*4 */
/* path-state read-back for parity (field names = PathState members, path-tracer.js:131-160) */
const void*     orc_render_field(const orc_render*, const char* name, size_t* bytes);
const int32_t*  orc_render_queue(const orc_render*, int which /*0 cam,1 mat*/, int64_t* n);
void        orc_tonemap_rgba8(const float* image, int64_t n_pixels, uint8_t* out);   /* display.js:98-113 */

/* ---- math spec (exported so tests can sweep it against libm) ---- */
void     orc_sincos(float theta, float* s, float* c);
uint32_t orc_base_hash(uint32_t px, uint32_t py);
void     orc_rand2(float seed, float* r0, float* r1);
uint32_t orc_f32_to_u32(float x);

#ifdef __cplusplus
}
#endif
#endif
