// pt_internal.h -- shared between the kernel translation units and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define PT_BLOCK 256                 // threads per workgroup: 4 wave64
#define PT_STEP_BLOCK 1024           // logic / compaction workgroups: 16 wave64, one count record per 1024 slots
#define PT_STACK_SIZE 64             // traversal stack entries per ray (reference: 32, scene.js:305)
#define PT_LDS_STACK 24              // pt_trace_rays (256-thread workgroups): the first PT_LDS_STACK entries live in LDS, the rest
                                     // in an HBM spill; the render's extend variants choose their own split (pt_kernels_extend.hip)
#define PT_SCENE_LDS_QUADS 512       // a scene of at most this many 16-B quads (12 per object + 7 per triangle) is staged in LDS
#define PT_STAGED_LDS_STACK 12       // LDS stack entries per ray of the staged extend variant
#define PT_WIDE_LDS_STACK 14         // ... of the kind-scheduled wide walk: 14 + 5 parked words per lane = 4864 B per wave, 33 waves per CU's 160 KB (32 run)
#define PT_TOP_LEVELS 7              // a mesh's tree-top block holds at most its first 7 levels (127 nodes, 8 KB); a scene uses as many
#define PT_TOP_SLOTS 127             // levels as fit PT_SCENE_LDS_QUADS together with its TLAS and objects (pt_api.hip)
#define PT_MAX_ITERS (1 << 20)       // traversal loop guard (degenerate N=1 trees can self-loop)
#define PT_SAMPLES_PER_TICKET 8      // logic.js:6 SAMPLES_PER_PIXEL_PER_PATH

// ---- device views ---------------------------------------------------------------------------
// The four scene buffers of the reference's bind group (scene.js:68-73), each addressed as
// 16-byte quads: a 64-B node = 4 quads, a 48-B triangle = 3, a 128-B object = 8.
struct SceneView {
    const float4* nodes;     // ONE node array for the walk (no reference counterpart; read-backs return the reference's two buffers):
                             //   ids [0, n_tris)                      the packed BLAS, inner nodes in depth-first order (k_reorder_nodes;
                             //                                        reference order with PT_TRAVERSAL_ORDER=0)
                             //   ids [n_tris, n_tris + n_objects)     the TLAS, its inner links moved up by n_tris (k_tlas_rebase): a walk
                             //                                        starts at tlas_root and never asks which tree a node id belongs to
                             //   ids [top_base, top_base + top_nodes) tree-top blocks (below)
    const float4* soa;       // extend variant 7 only (A/B of a structure-of-arrays node layout), else null: quad k of node i of `nodes` at soa[k * soa_stride + i]
    int32_t soa_stride;      // = n_tris + n_objects
    const float4* objects;   // pt_object[n_objects]
    const float4* tris;      // per triangle {v0, nx}{v1 - v0, ny}{v2 - v0, nz}: the reference's pt_triangle with what every ray
                             // would recompute from it -- the two edges (scene.js:479-481) and the unit normal
                             // (scene.js:233) -- evaluated once at build time by the same f32 operations (k_tri_records)
    const float4* wide;      // tolerance mode only (pt_set_traversal_mode): compressed 4-wide collapse of the BLAS (PT_WIDE_QUADS per node,
                             // wide ids in depth-first order of the wide tree), else null
    const float4* objects_wide;  // `objects` with every root replaced by the root of that mesh's wide tree (what a wide walk enters through)
    const float4* ext_mats;  // extension materials (pt_material = 2 quads each) for material ids 2.., null by default
    int32_t n_ext_mats;
    int32_t stack_bound;         // most stack entries any ray can hold in this scene (TLAS depth + deepest BLAS depth), or
                                 // INT32_MAX when unknown: only computed for scenes small enough to be staged in LDS
    int32_t n_objects, n_tris;   // record counts of the four buffers (the extend kernel stages tiny scenes in LDS)
    int32_t tlas_root;           // = n_tris: the node id a walk starts at
    int32_t sched_wi, sched_wl, sched_ws;   // kind-scheduled walk: weights of the inner-node, triangle and switch bodies in the choice of a pass
    int32_t sched_ww;                       // ... and of the 4-wide node body (tolerance mode)
    // Tree-top blocks (k_top_block): node ids top_base .. top_base + top_nodes - 1 of `nodes` hold the first levels of every mesh's
    // tree by heap slot; objects_top = `objects` with the roots pointing there.  A launch that stages them in LDS walks from
    // objects_top; every other launch walks from `objects` and never sees them.  top_nodes = 0: the scene has none.
    const float4* objects_top;
    int32_t top_base, top_nodes;
};

// PathState (path-tracer.js:131-160) as separate device arrays.  The vec3f members are stored as 12-byte records, not
// at the reference's 16-byte stride: the step streams the whole pool through the caches every iteration, and the pad
// words were 12 of its 108 bytes per slot (pt_path_state_read hands them back at the reference's stride).
// num_bounces, samples_in_pixel and flags share one word per slot (PT_ST_*): the logic pass reads and writes all three
// every step, and one load + one store replace three of each.  hit_obj / hit_tri are one int2 for the same reason.
struct P3 { float x, y, z; };
#define PT_ST_RETIRED 0x80000000u                                  // flags bit 31 (logic.js:72,179)
#define PT_ST_FLAGS(s) ((s) & 3u)                                  // flags bits 0-1 (material.js:118; bit 1: emitter extension)
#define PT_ST_SIP(s) (((s) >> 2) & 7u)                             // samples_in_pixel, 0..7 (logic.js:129-133)
#define PT_ST_NB(s) (((s) >> 8) & 0x7fffffu)                       // num_bounces (<= max_nb + 2; pt_render_set_extensions bounds max_nb)
#define PT_ST_PACK(flags, sip, nb) ((uint32_t)(flags) | ((uint32_t)(sip) << 2) | ((uint32_t)(nb) << 8))
#define PT_MAX_NB_LIMIT (1 << 22)
struct PathState {
    int32_t*  pixel_index;
    uint32_t* state;             // PT_ST_*: flags | samples_in_pixel | num_bounces
    float*    random_seed;
    P3*       path_throughput;
    float4*   material_throughput_pdf;
    P3*       path_o;
    P3*       path_d;
    int2*     hit;               // {hit_obj, hit_tri}
    float4*   ticket_acc;        // not in the reference: radiance sum + sample count of the slot's current pixel ticket
};

// Device-resident step header: queue sizes (QueuesStage*.stage_N_queue_size, path-tracer.js:173-204),
// the pixel ticket counter (PathState.pixel_counter) and run totals.
struct StepHeader {
    int32_t  n_camera, n_material, n_rays, first_empty_step;   // first_empty_step: -1 until both queues are empty
    int64_t  pixel_counter;      // next ticket to hand out
    int64_t  ticket_base;        // pixel_counter at the start of the current step
    int64_t  ticket_next[2];     // [step & 1]: first ticket of that step (double-buffered: workgroup 0 of the compaction
                                 // launch writes the next step's while the other workgroups still read this step's)
    int64_t  total_rays;
    int64_t  total_samples;
    int64_t  total_retired;      // slots retired by ticket exhaustion so far; each traced one wasted ray in the step that retired it
                                 // (logic.js:177-190: the slot is still pushed to the camera queue)
    unsigned long long overflow; // rays stopped by the stack / iteration guards
    unsigned long long visits[64][4];   // n_inner, n_leaf, n_enter, n_leave (counting variant only), spread over 64 slots
                                        // so that one word does not serialise every wave's atomic; summed by the host
    unsigned long long passes[4];       // counting variant only: passes of the kind-scheduled loop by body -- inner node, 4-wide node, triangle, switch -- summed over the waves
    unsigned long long lane_hist[4];    // counting variant only: node-body passes (inner or wide) by active lanes: <= 16, <= 32, <= 48, <= 64
    int32_t  step_index, pad;
    // Dispatch order of the extend kernel (performance only, results do not depend on it): sampled waves report their trip
    // count (the longest lane's) per ray class, the scan of the next step puts the class that held the longest wave first.
    uint32_t cls_max[32][2];             // {camera, bounce} maximum trips of a wave, spread over 32 cells
    int32_t  extend_order, pad2;         // 0 = camera rays first (queue order), 1 = bounce rays first
    // Tolerance mode (PT_TRAVERSAL_WIDE): the sampled workgroups of an extend launch walk their rays a second time, exactly as the
    // reference does, and add up how the two walks differ (RenderConst::check_period; pt_run_stats).
    unsigned long long chk_rays, chk_mismatch, chk_flips;   // rays walked twice; with another (hit_obj, hit_tri); of those, hit against miss
    unsigned long long chk_beyond, chk_lost;                // hit_dist apart by more than 1e-6 relative (flips included); of those, the wide walk's hit is the farther one (or a miss)
    uint32_t chk_max_rel_bits; int32_t prev_n_rays;   // prev_n_rays: n_rays of the step before (the rays the front workgroups re-walk)
    //     // largest |hit_dist(wide) - hit_dist(exact)| / hit_dist(exact) over the rays both walks hit (f32 bits; atomicMax)
};

struct CameraFrame { float fwd[3], right[3], up[3], fov_sin, aspect; };   // camera.js:96-111, evaluated once per render on the device
struct RenderConst {
    int32_t width, height, num_paths, spp;
    float   cam_pos[3], cam_look_at[3], fov_deg;
    int64_t slot_base, ticket_limit;
    CameraFrame frame;
    int32_t max_nb;      // logic.js:111 constant (20) unless the max_nb extension is set
    uint32_t seed_mix;   // 0 unless the seed extension is set
    // tolerance mode: the rays of one queue block in check_period are walked a second time, exactly (0 = none, 1 = all), by chk_cap
    // workgroups in front of the next step's extend launch; chk_buf: 2 x chk_cap x 9 x 64 words (ray, hit of the wide walk), chk_spill: those
    // workgroups' share of the traversal-stack spill.  Fixed for a render (set when it starts).
    int32_t xcd_chunk, pad0;     // > 0: the extend launch gives an XCD runs of xcd_chunk consecutive queue blocks (workgroup i runs on XCD i % 8), 0: block i to workgroup i
    int32_t check_period, chk_cap;
    float* chk_buf; int32_t* chk_spill;
};

// ---- launchers (pt_kernels_build.hip) ---------------------------------------------------------
void ptk_tri_aabb_morton(const float4* tris, int64_t n, const float bounds[6], float4* aabb, uint32_t* keys, hipStream_t s);
void ptk_obj_morton(const float4* aabb, int64_t n, const float bounds[6], uint32_t* keys, hipStream_t s);
void ptk_radix_tree(const uint32_t* sorted_keys, int64_t n, int32_t* parents /* 2n, zeroed */, int32_t* first /* n or null */, hipStream_t s);
// traversal-order (depth-first preorder) numbering of a tree's inner nodes and the reordered copy of a packed mesh's nodes
void ptk_dfs_index(const int32_t* parents, const int32_t* first, int64_t n, int32_t* dfs /* n */, int32_t* top_nodes /* PT_TOP_SLOTS, preset to -1, or null */,
                   uint8_t* depth /* n: links between node and root, or null */, hipStream_t s);
void ptk_top_block(const float4* packed_nodes, const int32_t* dfs, const int32_t* top_nodes, int64_t offset, int64_t n, int levels, int top_id, float4* trav_nodes, hipStream_t s);
void ptk_tlas_rebase(const float4* tlas_nodes, int n, int delta, float4* out /* the TLAS part of SceneView::nodes */, hipStream_t s);
void ptk_objects_top(const float4* objects, int n, const int32_t* mesh_root, const int32_t* top_root, int n_meshes, float4* out, hipStream_t s);
void ptk_soa_split(const float4* nodes, int64_t n, float4* soa /* 4 x n quads */, hipStream_t s);
void ptk_reorder_nodes(const float4* packed_nodes, const int32_t* dfs, int64_t offset, int64_t n, float4* trav_nodes, hipStream_t s);
void ptk_up_pass(const int32_t* idx, const float4* aabb, const int32_t* parents, int64_t n, float4* nodes /* zeroed */, hipStream_t s);
void ptk_rearrange(const float4* tris_in, const int32_t* idx, int64_t n, float4* tris_out, hipStream_t s);
void ptk_tri_records(const float4* tris, int64_t n, float4* records, hipStream_t s);   // SceneView::tris from pt_triangle[n]
void ptk_pack_mesh(const float4* src_nodes, const float4* src_tris, int64_t n, int32_t offset, float4* dst_nodes, float4* dst_tris, hipStream_t s);
// Tolerance mode (north_star: hit indices exact "except for counted grazing/edge rays within 1e-6"): a wide node = a
// binary node at an even depth of its mesh's tree with each inner child replaced by that child's two children -- up to four boxes per
// fetch -- and the boxes quantised to 8 bits per coordinate in a frame slightly larger than their union, rounded outwards (a compressed
// wide BVH): the traversal is bound by the number of dependent fetches and of 16-B lane requests it sends to L1, and this node is 4 quads
// for four children instead of 4 quads for two.  Wide nodes are numbered in depth-first order of the wide tree, the meshes one after the
// other (ptk_wide_count gives a mesh's count; only the binary nodes at even depths have one, a third of the triangle count).
//   quad 0: frame origin x, y, z (f32) | step.x (f32; the three steps are powers of two)
//   quad 1: child ref[4] (>= 0 wide node, < 0 triangle leaf, PT_WIDE_EMPTY = unused slot)
//   quad 2: lo.x[4] | lo.y[4] | lo.z[4] | hi.x[4]   (one byte per child)        quad 3: hi.y[4] | hi.z[4] | step.y | step.z
// child box = origin + q * step.
#define PT_WIDE_EMPTY 0x7fffffff
#define PT_WIDE_QUADS 4
// flags[dfs[i]] = 1 for the binary nodes i of one mesh that become wide nodes; scan = exclusive prefix sum of flags in place, *total = their
// number (three launches; tile_sums: (n + 1023) / 1024 + 1 ints).
void ptk_wide_flags(const int32_t* dfs, const uint8_t* depth, int64_t n, int32_t* flags, hipStream_t s);
void ptk_exclusive_scan(int32_t* data, int64_t n, int32_t* tile_sums, int32_t* total, hipStream_t s);
// one packed mesh (binary nodes [offset, offset + n) of blas_nodes) -> its wide nodes at wide ids wide_base + rank[dfs[i]]
void ptk_collapse_wide(const float4* blas_nodes, int64_t offset, int64_t n, const int32_t* dfs, const uint8_t* depth, const int32_t* rank,
                       int32_t wide_base, float4* wide, hipStream_t s);
void ptk_pack_objects(const void* instances /* pt_instance[n] */, const int32_t* tlas_perm, const int32_t* mesh_tri_offset,
                      int n, float4* objects, hipStream_t s);
// pt_sort.hip: stable LSD radix sort of u32 keys; vout[i] = original position of the key that lands at i; kin is left alone.
// tmp == null: only *tmp_bytes (two key-sized buffers + the digit histograms) is set.
hipError_t pt_sort_keys_index(void* tmp, size_t* tmp_bytes, const uint32_t* kin, uint32_t* kout, int32_t* vout, size_t n, hipStream_t s);

// ---- launchers (pt_kernels_render.hip) --------------------------------------------------------
int  ptk_step_blocks(int num_paths);   // count records per step; the code array holds PT_STEP_BLOCK bytes per record, zero padded
// blk_counts: per PT_STEP_BLOCK slots two 64-bit count records {#camera | #material << 16 | #tickets << 32 | #finished << 48},
// [step & 1] read by that step's compaction, the other filled by its extend launch (first_logic writes step 0's)
void ptk_first_logic(const PathState& ps, const RenderConst& rc, uint8_t* code, int4* blk_counts, hipStream_t s);   // before step 0 only
// counts -> queues, tickets, camera rays, StepHeader.  parity = step & 1 (StepHeader::ticket_next)
void ptk_count_compact_generate_shade(const PathState& ps, const RenderConst& rc, const SceneView& sc, const uint8_t* code, int4* blk_counts, int parity, StepHeader* hdr,
                                int32_t* cam_q, int32_t* mat_q, int fat /* 0: four 256-thread workgroups per count record; 1: one (for steps with few live slots); 5: one of
                                128 threads (pools that overlap); 3, 4, 6, 7: other shapes, A/B only */, hipStream_t s);
// Extend over the step's ray queue (camera ++ material), followed per slot by the next step's logic and material sample
// (code != null; they write the decision codes and count records the next ptk_count_compact_generate_shade reads).  code == null: traversal only.
// count != 0 selects the visit-counting variant.
void ptk_extend_queue(const PathState& ps, const RenderConst& rc, const SceneView& sc, StepHeader* hdr,
                      const int32_t* cam_q, const int32_t* mat_q, int32_t* spill, int count, int variant,
                      unsigned long long* wave_log /* null unless measuring */, float* image, uint8_t* code,
                      unsigned long long* rec_next /* the next step's count records, used with code */,
                      int grid_rays /* an upper bound of the queue length; the launch holds this many threads */, hipStream_t s);   // uses rc.max_nb
int64_t ptk_extend_waves(int num_paths, int variant);   // waves the launch of that variant holds (wave_log entries)
int ptk_extend_num_variants();
const char* ptk_extend_variant_name(int v);
// Extend over n caller rays (o / d: `stride` floats per ray, 4 = float4 records, 3 = packed).  Any output may be null.
// spill: ptk_spill_ints(n) ints (sized for the smallest LDS share of any variant, so more than this kernel's 24-entry share needs).
void ptk_extend_direct(const SceneView& sc, const float* o, const float* d, int stride, int n, int32_t* hit_obj,
                       int32_t* hit_tri, float* hit_dist, int4* counts4, int32_t* spill, StepHeader* hdr, hipStream_t s);
void ptk_camera_frame(const RenderConst& rc, CameraFrame* out_dev, hipStream_t s);   // one thread; the host copies the result into rc.frame
void ptk_flush_tickets(const PathState& ps, int num_paths, float* image, hipStream_t s);
void ptk_tonemap(const float4* image, int64_t n_pixels, uint32_t* rgba8, hipStream_t s);
size_t ptk_spill_ints(int64_t n_rays);   // ints needed in the spill buffer for a launch over n_rays
