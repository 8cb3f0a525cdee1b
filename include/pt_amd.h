/*
 * pt_amd.h -- C ABI of the MI355X-native path-tracer hot path (libpt_amd.so).
 *
 * Every entry point replaces one interface of the reference (AddisonPrairie/Path-Tracer,
 * paths relative to /root/reference).  Plain pointers and sizes only; no torch / HIP types.
 * All functions return 0 on success and a negative pt_status on failure; pt_last_error()
 * gives the message.  A pt_ctx is bound to ONE GPU and is single-caller (not thread-safe),
 * like the reference's single JS thread (ray-trace.js:38-39: every dispatch is awaited).
 *
 * Ownership: the ctx owns all device memory it allocates; the caller owns every host
 * pointer it passes in or out.  "dev" pointers are device addresses on the ctx's GPU.
 */
#ifndef PT_AMD_H
#define PT_AMD_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_ctx pt_ctx;

enum pt_status {
    PT_OK = 0, PT_ERR_ARG = -1, PT_ERR_HIP = -2, PT_ERR_STATE = -3, PT_ERR_OOM = -4, PT_ERR_NOGPU = -5
};

/* ---- records shared with the reference's WGSL structs (bit-exact layouts) ---- */
/* scene.js:265-274 BVHNode.  child >= 0: inner node index; child < 0: leaf -(child+1).
 * f_1 / f_2 are excluded from parity (reference: race-dependent arrival counter / 0). */
typedef struct { float l_min[3]; int32_t l_child; float l_max[3]; int32_t f_1;
                 float r_min[3]; int32_t r_child; float r_max[3]; int32_t f_2; } pt_bvh_node;   /* 64 B */
/* scene.js:294-298 Triangle (3 x vec3f padded to 16 B; parse-obj.js:64-68 pads with 3.1415) */
typedef struct { float v0[3], pad0, v1[3], pad1, v2[3], pad2; } pt_triangle;                     /* 48 B */
/* scene.js:277-292 Object, as packed by builders.js:28-66 */
typedef struct { float l2w[3][4]; int32_t bvh_offset, material, f_2, f_3;
                 float w2l[3][4]; int32_t f_4, f_5, f_6, f_7; } pt_object;                       /* 128 B */

/* One instanceMesh() call after the host computed its matrices (scene.js:523-529,
 * transform-utils.js:45-59 column-major, rounded to f32 as builders.js:33-58 / :99-107 do). */
typedef struct {
    float   l2w[16];        /* localToWorld, column-major */
    float   w2l[16];        /* worldToLocal, column-major */
    float   aabb_min[3];    /* world AABB, transform-utils.js:11-43 */
    float   aabb_max[3];
    int32_t mesh;           /* handle returned by pt_mesh_upload */
    int32_t material;       /* material.js:104-111: 0 -> albedo .5, 1 -> albedo .2 */
} pt_instance;

/* ---- context ---- */
/* replaces: navigator.gpu.requestAdapter()/requestDevice() (main.js:3-4) + initScene(device) (scene.js:1) */
int         pt_create(int device_id, pt_ctx** out);
/* PT_ERR_STATE (and nothing is freed) while contexts made by pt_create_sharing_scene still borrow this one's scene. */
int         pt_destroy(pt_ctx* ctx);
const char* pt_last_error(const pt_ctx* ctx);
/* Optional: run all work on a caller-owned hipStream_t (e.g. torch's current stream). NULL = own stream. */
int         pt_set_stream(pt_ctx* ctx, void* hip_stream);
int         pt_sync(pt_ctx* ctx);
/* "0000:c1:00.0"-style PCI bus id of the ctx's GPU (hipDeviceGetPCIBusId): lets a multi-GPU host print which physical device each rank drives. */
int         pt_device_pci_bus_id(pt_ctx* ctx, char* out, size_t cap);

/* ---- scene (scripts/scene) ---- */
/* replaces: scene.registerMesh({file}) (scene.js:518-521) after parseObj (builders.js:155):
 * tris12 = parseObj().triangleArray (12 f32 per triangle), bounds = {min xyz, max xyz} over all
 * `v` lines (parse-obj.js:33-38).  Returns the mesh handle (>= 0) or a negative status.
 * The BLAS is built on the device on first use by pt_scene_build and cached (scene.js:25-27). */
int pt_mesh_upload(pt_ctx* ctx, const float* tris12, int64_t n_tris, const float bounds[6]);
/* replaces: await scene.build() (scene.js:15-146): per-mesh LBVH (builders.js:153-214), packMeshes
 * (pack-t-mesh.js), TLAS (builders.js:97-151), packObjects (builders.js:25-95). */
int pt_scene_build(pt_ctx* ctx, const pt_instance* instances, int n_instances);

typedef struct { int64_t n_tris; int32_t n_objects; int32_t n_meshes_used; } pt_scene_info;
int pt_scene_get_info(pt_ctx* ctx, pt_scene_info* out);
/* Parity read-backs of the four scene buffers of the reference's bind group (scene.js:68-73)
 * and of the sort permutations (builders.js:26 reads the TLAS one back, too). n = element count. */
int pt_get_blas_nodes(pt_ctx* ctx, pt_bvh_node* out, int64_t n);
int pt_get_blas_tris(pt_ctx* ctx, pt_triangle* out, int64_t n);
int pt_get_tlas_nodes(pt_ctx* ctx, pt_bvh_node* out, int64_t n);
int pt_get_objects(pt_ctx* ctx, pt_object* out, int64_t n);
int pt_get_tlas_perm(pt_ctx* ctx, int32_t* out, int64_t n);
int pt_get_mesh_perm(pt_ctx* ctx, int mesh, int32_t* out, int64_t n);
/* Test hook, no reference counterpart: the most traversal-stack entries any ray can hold in the built scene (inner nodes
 * on the longest TLAS path + on the longest BLAS path), or INT32_MAX when no bound exists: a scene with a one-primitive tree (one
 * instance, or a one-triangle mesh), whose right child is the node itself (SURVEY "Degenerate N=1"), or more than 2^20 objects.  Within
 * the LDS share of a staged scene the extend kernel drops its spill path and its guards; within 64 the default kernel drops the guards. */
int pt_scene_get_stack_bound(pt_ctx* ctx, int32_t* bound);

/* Test hook over the extend kernel (ray-trace.js:53-69 + scene.js:313-475 intersect_bvh) for
 * caller-supplied rays.  o3/d3: n x 3 floats (host).  Outputs (host, any may be NULL):
 * hit_obj (TLAS-sorted object index or -1), hit_tri (packed triangle index or -1), hit_dist,
 * counts4 = per-ray {n_inner, n_leaf, n_enter, n_leave} visit counts. */
int pt_trace_rays(pt_ctx* ctx, const float* o3, const float* d3, int64_t n,
                  int32_t* hit_obj, int32_t* hit_tri, float* hit_dist, int32_t* counts4);
/* How many rays of the last pt_trace_rays call on this context were stopped by the traversal guards (a push onto a full
 * 64-entry stack or more than 2^20 loop trips: the WGSL runtime's array clamp has no other counterpart); their hits are
 * the ones found up to that point.  0 on every sane scene; parity tests assert it. */
int pt_trace_last_guard_hits(pt_ctx* ctx, int64_t* out);

/* pt_trace_rays for callers that cannot allocate -- NO counterpart in the reference (a WebGPU host cannot be in that position);
 * it exists for code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same kernel, same
 * results as pt_trace_rays, but: every pointer is a DEVICE address on the ctx's GPU (o3_dev / d3_dev: n x 3 packed floats;
 * outputs as in pt_trace_rays, any may be NULL; counts4_dev 16-byte aligned; guard_hits_dev: one uint64 = rays stopped by the
 * traversal guards); the call allocates and frees nothing, waits for nothing, and returns once the work is on the ctx's
 * stream (pt_sync or the caller's own stream ordering makes the outputs readable).  The traversal-stack spill and the guard
 * counter live in a file-scope device array of the library (10.5 MB on every GPU that loads it, for the life of the process)
 * sized for chunks of 65,536 rays; a call works through its rays chunk by chunk.  At most PT_TRACE_NOALLOC_MAX_RAYS rays per
 * call: more is PT_ERR_ARG.  All contexts of one GPU share that array: calls that overlap on one GPU (two contexts, two
 * streams, two threads) race on it -- issue them one after the other. */
#define PT_TRACE_NOALLOC_MAX_RAYS 4194304
int pt_trace_rays_noalloc(pt_ctx* ctx, const float* o3_dev, const float* d3_dev, int64_t n,
                          int32_t* hit_obj_dev, int32_t* hit_tri_dev, float* hit_dist_dev, int32_t* counts4_dev,
                          uint64_t* guard_hits_dev);

/* ---- render (scripts/render) ---- */
typedef struct {
    int32_t width, height;           /* params.image (path-tracer.js:57-58) */
    float   cam_pos[3];              /* params.camera.position (path-tracer.js:64-66) */
    float   cam_look_at[3];          /* params.camera.lookAt   (path-tracer.js:68-70) */
    float   fov_deg;                 /* params.camera.fov      (path-tracer.js:62) */
    int32_t spp;                     /* params.settings.samples (path-tracer.js:19) */
    int32_t num_paths;               /* NUM_PATHS (path-tracer.js:4); 0 -> 1,000,000 */
    /* multi-GPU spp split (no reference counterpart; all 0 / -1 reproduces the reference) */
    int64_t slot_base;               /* global slot index of local slot 0 (RNG stream id, logic.js:77) */
    int64_t ticket_begin, ticket_end;/* this GPU's pixel-ticket range; ticket_end < 0 -> all tickets */
    void*   image_dev;               /* optional caller-owned W*H*4 f32 device buffer (zeroed by init) */
} pt_render_params;

/* replaces: initPathTracer({device, scene, image, camera, settings}) (path-tracer.js:1-47,207-329) */
int pt_render_init(pt_ctx* ctx, const pt_render_params* params);

/* Start the same render again without reallocating: every buffer pt_render_init created is zeroed
 * again, as a second initPathTracer() call would find them (new GPUBuffers are zero-filled). */
int pt_render_reset(pt_ctx* ctx);

/* ---- extensions (NO reference counterpart: the reference has two grey Lambert materials, a hard-coded bounce
 * cut and no seed -- material.js:106-111, logic.js:111, logic.js:77).  Default-off: without this call, or with an all-zero
 * pt_render_ext, every result is the reference's.  The rules are defined by the oracle (oracle/pt_oracle.h) and reproduced
 * bit for bit by the kernels.  Call after pt_render_init (or pt_render_reset) and before the first step; it survives
 * pt_render_reset.
 *   max_nb     replaces the 20 of logic.js:111 (num_bounces advances by 2 per bounce there); 0 keeps 20; at most 4194304
 *   seed       decorrelates the random streams: added (times 0x9E3779B1 mod 2^32) to the slot index hashed at logic.js:77
 *   materials  materials[k] describes material id k + 2; ids 0 and 1 stay the reference's; an id without an entry keeps the
 *              reference's unknown-id behaviour (zero BRDF)
 *                PT_MAT_LAMBERT  rgb albedo (linear), the reference's cosine sampling
 *                PT_MAT_EMITTER  the path ends at the hit and adds throughput * emission
 *                PT_MAT_MIRROR   wi = d - 2 (d.n) n, throughput *= albedo
 *                PT_MAT_GLOSSY   a mirror blurred by roughness = emission[0] in [0, 1] towards the cosine lobe about the normal that
 *                                faces the ray: wi = normalize(reflect * (1 - roughness) + cosine_sample * roughness), throughput *= albedo */
enum { PT_MAT_LAMBERT = 0, PT_MAT_EMITTER = 1, PT_MAT_MIRROR = 2, PT_MAT_GLOSSY = 3, PT_MAX_EXT_MATERIALS = 14 };
typedef struct { float albedo[3]; int32_t type; float emission[3]; int32_t pad; } pt_material;   /* 32 B */
typedef struct { int32_t max_nb; uint32_t seed; int32_t n_materials, pad; pt_material materials[PT_MAX_EXT_MATERIALS]; } pt_render_ext;
int pt_render_set_extensions(pt_ctx* ctx, const pt_render_ext* ext);

typedef struct {
    int64_t n_camera;     /* stage_2_queue_size[0] after logic (path-tracer.js:174) */
    int64_t n_material;   /* stage_2_queue_size[1] */
    int64_t n_rays;       /* stage_3_queue_size[0]: rays traced this step */
    int64_t step_index;
} pt_step_stats;
/* replaces: await pt.step() (path-tracer.js:49-123): logic -> camera -> material -> ray-trace.
 * stats may be NULL (then no host sync happens). */
int pt_render_step(pt_ctx* ctx, pt_step_stats* stats);

typedef struct {
    int64_t steps;              /* step() calls issued */
    int64_t total_rays;         /* sum of stage-3 queue sizes */
    int64_t total_samples;      /* finished camera samples (= sum of image .w) */
    int64_t stack_overflows;    /* rays that hit the traversal guards (0 on sane scenes) */
    double  extend_ms;          /* sum of the durations (HIP events) of the timed extend launches: one launch in sixteen, or all */
    int64_t extend_launches;    /* how many launches were timed */
    int64_t traversal_bytes;    /* algorithmic bytes if counted, else 0 */
    int64_t wasted_rays;        /* part of total_rays: rays traced for slots that ticket exhaustion had just retired
                                   (logic.js:177-190 pushes them to the camera queue regardless); one per retired slot */
    /* tolerance mode only (PT_TRAVERSAL_WIDE; all 0 otherwise): what the sampled exact re-walks found (pt_set_traversal_check) */
    int64_t checked_rays;       /* rays of this render that were walked a second time, exactly as the reference walks them */
    int64_t mismatched_rays;    /* of those: the two walks returned a different (hit_obj, hit_tri) */
    int64_t flipped_rays;       /* of the mismatched: one walk hit and the other missed */
    double  max_rel_dist_diff;  /* largest |hit_dist - exact hit_dist| / exact hit_dist over the checked rays that both walks hit */
    int64_t beyond_tolerance_rays; /* checked rays whose two hit_dist differ by more than 1e-6 relative, flipped rays included: north_star
                                      allows differing indices "within 1e-6"; these are the rays outside that allowance */
    int64_t lost_hit_rays;      /* of those: the tolerance mode's hit is the FARTHER one (or a miss): a hit the reference finds was lost.
                                   The others found a closer hit than the reference walk, whose own box test had culled it */
} pt_run_stats;
enum { PT_RUN_TIME_EXTEND = 1, PT_RUN_COUNT_BYTES = 2,
       PT_RUN_TIME_EXTEND_ALL = 4 /* events around every extend launch, not one in sixteen: for measuring the kernel, costs the render ~5 % */ };
/* The reference's requestAnimationFrame loop (main.js:61-72) made headless: step until both
 * stage-2 queues are empty (or max_steps).  flags: PT_RUN_* . */
int pt_render_run(pt_ctx* ctx, int64_t max_steps, int flags, pt_run_stats* out);

/* ---- several path pools on one GPU (NO reference counterpart: SURVEY 8e's sample split, applied inside one device) ----
 * pt_create_sharing_scene: a second context on src's device that borrows src's built scene (no copy); it has its own
 * stream and its own render state (pt_render_init with its own slot_base / ticket range, like one GPU of a multi-GPU
 * render; give every context the same image_dev to accumulate one image).  It cannot upload meshes, build a scene or
 * change the traversal mode.  While it lives src keeps count of it: pt_scene_build, pt_set_traversal_mode and pt_destroy on
 * src return PT_ERR_STATE, and so does pt_render_init on src while a borrower still accumulates into src's own image.
 * pt_render_run_many: pt_render_run over n contexts at once from one host thread -- their batches of steps are
 * issued in turn on their own streams, so that one pool's compaction launches run in the tail of another's extend
 * launch.  The contexts may also sit on different GPUs (each with its own scene replica and its own share: the
 * single-process form of the multi-GPU render, see pt_comm_create).  out: n entries (or NULL).  Every pool is bit-identical to the oracle run with the same share; overlapping
 * pools only change when launches run. */
int pt_create_sharing_scene(pt_ctx* src, pt_ctx** out);
int pt_render_run_many(pt_ctx** ctxs, int n, int64_t max_steps, int flags, pt_run_stats* out);

/* replaces: pt.getImageBuffer() (path-tracer.js:125-127) + a read-back the reference lacks.
 * rgba: W*H*4 floats (rgb = sum of radiance, a = finished samples; logic.js:146-149). */
int pt_image_read(pt_ctx* ctx, float* rgba, int64_t n_floats);
int pt_image_dev_ptr(pt_ctx* ctx, void** dev_ptr);
/* Adds the partial sums of unfinished pixel tickets to the image (on the ctx's stream, not waited for): what pt_image_read does
 * before it copies.  For callers that consume the device buffer themselves in the middle of a render. */
int pt_image_flush(pt_ctx* ctx);
int pt_image_size(pt_ctx* ctx, int32_t* width, int32_t* height);   /* params.image of the current render */
/* Parity read-back of one PathState member (path-tracer.js:131-160) by name; vec3f/vec4f members
 * come back as 4 floats per slot.  "pixel_counter" returns one int64. */
int pt_path_state_read(pt_ctx* ctx, const char* field, void* out, size_t bytes);
/* stage-2 queues of the last step: which = 0 camera_queue, 1 material_queue. *n_io in: capacity, out: count */
int pt_queue_read(pt_ctx* ctx, int which, int32_t* out, int64_t* n_io);

/* ---- multi-GPU (NO reference counterpart: the reference is a single-device browser app) ----
 * Samples are independent, so a render shards with no exchange on the data path: every GPU holds the whole scene and renders
 * its own pixel-ticket range with its own RNG streams (pt_render_params.slot_base / ticket_begin / ticket_end), into its own
 * accumulation buffer.  The ONE collective is the sum of those buffers: pt_comm_reduce_image = RCCL ncclReduce(sum, float32,
 * W*H*4) over xGMI onto rank `root`'s image, in place, ordered behind the work on the contexts' streams, waited for.
 * The image is sums of radiance and sample counts (logic.js:146-149), so the sum has the single-GPU image's form.
 * The reduce MOVES the samples: afterwards the root's image holds every rank's samples and every other rank's image is zero, so it
 * may be called again later in the same render (a progressive refresh) and adds only what has been accumulated since -- nothing is
 * counted twice.  It knows only the communicator's own contexts: pools made by pt_create_sharing_scene that add into the same images
 * must have finished (pt_render_run / pt_render_run_many return finished; after pt_render_step: pt_sync) and, in the middle of a
 * render, have had their open pixel tickets flushed (pt_image_flush + pt_sync) before the call.
 *   pt_comm_create        one process drives several GPUs: ctxs[i] is rank i (one context per GPU; ncclCommInitAll)
 *   pt_comm_unique_id +   one process per GPU: rank 0 makes the 128-byte id, the host passes it to the other ranks by its own
 *   pt_comm_create_rank   means (environment, file, torch.distributed store), every rank joins with its context (ncclCommInitRank)
 * librccl is loaded on first use; without it these return PT_ERR_STATE. */
typedef struct pt_comm pt_comm;
#define PT_COMM_ID_BYTES 128
int  pt_comm_create(pt_ctx** ctxs, int n, pt_comm** out);
int  pt_comm_unique_id(void* id128);
int  pt_comm_create_rank(pt_ctx* ctx, const void* id128, int rank, int world, pt_comm** out);
int  pt_comm_reduce_image(pt_comm* comm, int root);
void pt_comm_destroy(pt_comm* comm);

/* ---- tolerance mode (NO reference counterpart, default off) ----
 * BASELINE.json north_star: "per-ray closest-hit triangle indices must be bit-exact, except for counted grazing/edge rays within
 * 1e-6".  PT_TRAVERSAL_REFERENCE (default) is the bit-exact walk of scene.js:313-475.  PT_TRAVERSAL_WIDE uses that allowance:
 * inside an object the extend kernel walks a compressed 4-wide collapse of the reference's binary BVH (each inner child replaced by
 * its two children: up to four outward-rounded boxes per 64-B fetch, about half the dependent fetches per ray); triangles, the
 * triangle test and the TLAS are the reference's.  The closest hit is the same except between triangles at (nearly) equal distance
 * along the ray -- shared edges, grazing rays -- where the other visit order may keep the other one; the low bits of hit_dist differ
 * too (the reference carries hit_dist through its object-space round trips, which then happen at other moments), so path state is no
 * longer bit-comparable with the oracle: sample counts stay exact, images agree statistically.  The differing rays are COUNTED: one
 * 64-ray block of every step's ray queue in `period` (pt_set_traversal_check; default 256, 1 = every ray, 0 = none; which blocks rotates
 * with the step) is walked a second time with the reference walk, by extra workgroups of the next step's launch, and pt_run_stats reports how many of those returned another (hit_obj, hit_tri) and the
 * largest relative hit_dist difference.  PT_TRAVERSAL_AUTO picks WIDE when the scene has at least 2^18 packed triangles per object
 * -- long walks inside few objects, where halving the dependent fetches pays (+15 ... +31 % on a one-mesh scene of 0.5 ... 10 M
 * triangles) -- and REFERENCE otherwise (many objects or small meshes lose 2 ... 26 % to the decode); pt_get_traversal_mode tells
 * which walk the built scene uses.  The reference-exact buffers (pt_get_*) are unchanged; the collapse
 * costs about 22 B per triangle of extra device memory.  Contexts made by pt_create_sharing_scene inherit mode and period. */
enum { PT_TRAVERSAL_REFERENCE = 0, PT_TRAVERSAL_WIDE = 1, PT_TRAVERSAL_AUTO = 2 };
int pt_set_traversal_mode(pt_ctx* ctx, int mode);
int pt_get_traversal_mode(pt_ctx* ctx, int* mode_in_use);      /* PT_TRAVERSAL_REFERENCE or PT_TRAVERSAL_WIDE; needs a built scene */
int pt_set_traversal_check(pt_ctx* ctx, int period);            /* takes effect when the next render starts (pt_render_init / pt_render_reset) */

/* ---- measurement hooks (no reference counterpart; used by tools/ and bench.py only) ---- */
/* The extend stage exists in several variants that differ in ray-to-lane mapping and stack placement, never in
 * results.  pt_extend_variants returns how many there are and copies the name of `variant`. */
int pt_extend_variants(pt_ctx* ctx, int variant, char* name_out, size_t name_cap);
int pt_set_extend_variant(pt_ctx* ctx, int variant);      /* -1 (default): the library picks per scene; also: environment PT_EXTEND_VARIANT at pt_create */
/* Re-run the extend kernel `reps` times on the ray queue of the last step (idempotent) and report its mean
 * duration in microseconds: A/B of variants on identical work inside one process. */
int pt_extend_replay(pt_ctx* ctx, int variant, int reps, double* avg_us);
/* One more replay of `variant` in which every wave logs 4 x uint64: {start, end} in 100 MHz ticks, XCC_ID << 32 | HW_ID,
 * block << 32 | class << 24 | trips (class 0 camera, 1 bounce).  out: cap_waves * 4 uint64.  tools/wave_timeline.py */
int pt_extend_wave_log(pt_ctx* ctx, int variant, uint64_t* out, int64_t cap_waves, int64_t* n_waves);

/* After a pt_render_run with PT_RUN_COUNT_BYTES: out12[0..3] = passes of the kind-scheduled traversal loop by body (binary inner node,
 * 4-wide node, triangle, object switch), summed over the waves; out12[4..7] = lane visits (n_inner -- 64-B node records of either kind --,
 * n_leaf, n_enter, n_leave); out12[8..11] = node-body passes (inner or 4-wide) by the lanes they served: <= 16, <= 32, <= 48, <= 64.
 * Lane use of a body = its lane visits / (64 x its passes).  out12 holds 12 entries. */
int pt_extend_pass_counts(pt_ctx* ctx, int64_t* out12);

/* ---- display (scripts/display) ---- */
/* replaces: initDisplay().draw() fragment shader (display.js:98-113) + canvas unorm8 store.
 * image_dev NULL -> the ctx's image.  rgba8: W*H*4 bytes (host), row 0 = top. */
int pt_tonemap_rgba8(pt_ctx* ctx, const void* image_dev, int32_t width, int32_t height, uint8_t* rgba8);

#ifdef __cplusplus
}
#endif
#endif
