// pt_api.hip -- host side of libpt_amd.so: context, device memory, launch sequencing, C ABI.
// Declarations and the reference interface each entry point replaces: include/pt_amd.h.
#include "pt_amd.h"
#include "pt_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

static_assert(sizeof(pt_bvh_node) == 64 && sizeof(pt_triangle) == 48 && sizeof(pt_object) == 128, "record layouts");
static_assert(sizeof(pt_instance) == 160, "pt_instance is 40 words (k_pack_objects)");

// ---- scratch of pt_trace_rays_noalloc: device memory that belongs to the code object, so the entry point allocates nothing ----
// [StepHeader (guard counter), padded to PT_NOALLOC_HDR_BYTES | traversal-stack spill of one chunk of PT_NOALLOC_CHUNK rays]
// Every GPU that loads the library holds its own copy of this array (10.5 MB) until the process ends.  In host code the name
// `scratch` is only a handle: the device address is asked for once per device (scratch_address, called by pt_create with that
// device current) and everything below uses that pointer.
#define PT_NOALLOC_CHUNK 65536
#define PT_NOALLOC_HDR_BYTES 4096
#define PT_NOALLOC_SCRATCH_BYTES (PT_NOALLOC_HDR_BYTES + (size_t)PT_NOALLOC_CHUNK * (PT_STACK_SIZE - PT_LDS_STACK) * 4)
static_assert(sizeof(StepHeader) <= PT_NOALLOC_HDR_BYTES && PT_NOALLOC_CHUNK % 256 == 0, "scratch layout");
static __device__ __align__(256) unsigned char scratch[PT_NOALLOC_SCRATCH_BYTES];

namespace {

thread_local std::string g_err;   // errors raised before a ctx exists

constexpr int PT_MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[PT_MAX_DEVICES];   // device address of `scratch` on each GPU, null until asked for

// The current device must be `device`.
hipError_t scratch_address(int device, void** out) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

struct Mesh {
    int64_t n = 0; float bounds[6] = {0, 0, 0, 0, 0, 0};
    float4* d_tris_in = nullptr;      // upload order (parse-obj.js order)
    float4* d_nodes = nullptr;        // per-mesh BLAS, child pointers relative to the mesh
    float4* d_tris_sorted = nullptr;  // Morton order
    int32_t* d_perm = nullptr;        // sorted position -> upload position
    int32_t* d_dfs = nullptr;         // inner node -> its number in depth-first preorder (traversal-order copy of the BLAS);
                                      // behind these n entries: PT_TOP_SLOTS + 1 ints, the nodes of the first levels by heap slot (k_dfs_index)
    uint8_t* d_depth = nullptr;       // inner node -> links between it and the root (the even depths become the wide nodes of the tolerance mode)
    bool built = false, used = false; int64_t tri_offset = 0;
};

}  // namespace

struct pt_ctx {
    int device = 0; int extend_variant = 0; hipStream_t stream = nullptr; bool own_stream = false; std::string err;
    std::vector<Mesh> meshes;
    // scene
    bool scene_borrowed = false;   // pt_create_sharing_scene: the scene buffers below belong to another context
    pt_ctx* lender = nullptr;      // ... namely this one, which lists us in its `borrowers` until we are destroyed
    std::vector<pt_ctx*> borrowers;
    int64_t last_guard_hits = 0;   // pt_trace_rays: rays stopped by the traversal guards in the last call
    void* d_scratch = nullptr;     // this GPU's copy of the file-scope `scratch` array (pt_trace_rays_noalloc)
    bool scene_built = false; int64_t n_tris = 0; int n_objects = 0; int n_meshes_used = 0; int32_t stack_bound = INT32_MAX;
    float4 *d_blas_nodes = nullptr, *d_blas_tris = nullptr, *d_tlas_nodes = nullptr, *d_objects = nullptr;
    float4* d_tri_records = nullptr;                    // SceneView::tris (edges + normal per packed triangle); d_blas_tris stays the reference's buffer
    float4* d_objects_top = nullptr; int32_t top_nodes = 0;   // tree-top blocks at node ids n_tris .. of d_blas_trav (SceneView::objects_top)
    float4* d_soa = nullptr;                            // extend variant 7 only: d_blas_trav's BLAS and TLAS records as four arrays of quads (built on demand)
    float4* d_blas_trav = nullptr;                      // SceneView::blas: the packed BLAS with inner nodes in depth-first order; d_blas_nodes stays the reference's
    int32_t* d_tlas_perm = nullptr;
    // render
    bool render_ready = false; RenderConst rc{}; PathState ps{};
    float* d_image = nullptr; bool own_image = false; uint8_t* d_code = nullptr;
    int4 *d_blk_counts = nullptr; int32_t *d_cam_q = nullptr, *d_mat_q = nullptr, *d_spill = nullptr;
    StepHeader* d_hdr = nullptr; StepHeader* h_hdr = nullptr; int nblocks = 0; int64_t step = 0;
    bool overlapped = false;    // the last pt_render_run_many drove this context together with others
    int64_t known_rays = -1;    // queue size in the last header the host has seen (-1: none yet); picks the compaction kernel's shape
    int64_t rays_cap = INT64_MAX;   // an upper bound of every coming step's queue length: the pool size, and once a header shows the
                                    // tickets used up, that header's queue length (retired slots never come back: the queue only shrinks)
    float4* d_ext_mats = nullptr; int n_ext_mats = 0;   // extension materials (pt_render_set_extensions)
    float* d_chk_buf = nullptr; int32_t* d_chk_spill = nullptr; int chk_cap_alloc = 0;   // tolerance mode: side buffers of the exact re-walk (RenderConst::chk_*)
    float4* d_wide = nullptr; float4* d_objects_wide = nullptr; int64_t n_wide = 0;   // tolerance mode: 4-wide collapse of the packed BLAS, objects that enter through it
    int traversal_mode = 0;        // what pt_set_traversal_mode was given (PT_TRAVERSAL_AUTO included)
    int check_period = 256;        // tolerance mode: one extend workgroup in check_period re-walks its rays exactly (pt_set_traversal_check)
    std::vector<std::pair<void*, size_t>> render_allocs; std::vector<hipEvent_t> events; int64_t ticket_begin = 0;
};

namespace {

int fail(pt_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; else g_err = msg; return code; }

#define HIPOK(c, expr)                                                                                          \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess)                                                                                   \
            return fail(c, e_ == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP,                                 \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                     \
    } while (0)

template <typename T> int dalloc(pt_ctx* c, T** p, size_t count, bool zero) {
    *p = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    HIPOK(c, hipMalloc((void**)p, bytes));
    if (zero) HIPOK(c, hipMemsetAsync(*p, 0, bytes, c->stream));
    return PT_OK;
}
void dfree(void* p) { if (p) (void)hipFree(p); }

// Device temporaries of one call: released on every exit path, error returns included.
struct Tmp {
    std::vector<void*> held;
    ~Tmp() { for (void* p : held) (void)hipFree(p); }
    template <typename T> int alloc(pt_ctx* c, T** p, size_t count, bool zero) {
        int rc = dalloc(c, p, count, zero);
        if (rc == PT_OK) held.push_back(*p);
        return rc;
    }
};

int read_back(pt_ctx* c, void* dst, const void* src, size_t bytes);

// Most stack entries a ray can hold at once: every entry is the far child of an inner node on the path from the root to
// the node being visited (TLAS part below switch_pt, BLAS part above it), so the bound is the number of inner nodes on the
// longest root-to-leaf path of the TLAS plus that of the deepest BLAS.  Only evaluated (on the host, from the built nodes)
// for scenes small enough for the LDS-staged extend variant, whose kernel then drops the spill path and the overflow guard.
// A tree the walk cannot finish on (the degenerate one-primitive tree can point at itself) leaves the bound unknown.
int tree_path_inner_nodes(const std::vector<pt_bvh_node>& nodes, int64_t first, int64_t count, int64_t root) {
    std::vector<std::pair<int64_t, int>> todo{{root, 1}};
    int best = 0; int64_t visits = 0;
    while (!todo.empty()) {
        auto [n, depth] = todo.back(); todo.pop_back();
        if (n < first || n >= first + count || ++visits > count) return INT32_MAX;
        best = std::max(best, depth);
        if (nodes[(size_t)n].l_child >= 0) todo.push_back({nodes[(size_t)n].l_child, depth + 1});
        if (nodes[(size_t)n].r_child >= 0) todo.push_back({nodes[(size_t)n].r_child, depth + 1});
    }
    return best;
}
int scene_stack_bound(pt_ctx* c) {
    c->stack_bound = INT32_MAX;
    if (12 * (int64_t)c->n_objects + 7 * c->n_tris > PT_SCENE_LDS_QUADS) return PT_OK;
    std::vector<pt_bvh_node> tlas((size_t)c->n_objects), blas((size_t)c->n_tris); std::vector<pt_object> objs((size_t)c->n_objects);
    int r;
    if ((r = read_back(c, tlas.data(), c->d_tlas_nodes, tlas.size() * sizeof(pt_bvh_node))) || (r = read_back(c, blas.data(), c->d_blas_nodes, blas.size() * sizeof(pt_bvh_node))) ||
        (r = read_back(c, objs.data(), c->d_objects, objs.size() * sizeof(pt_object)))) return r;
    int64_t bound = tree_path_inner_nodes(tlas, 0, c->n_objects, 0), deepest = 0;
    for (const pt_object& o : objs) deepest = std::max<int64_t>(deepest, tree_path_inner_nodes(blas, 0, c->n_tris, o.bvh_offset));
    c->stack_bound = (int32_t)std::min<int64_t>(bound + deepest, INT32_MAX);
    return PT_OK;
}

// 8 waves/SIMD, class of the longest wave first, a tiny scene staged in LDS, kind-scheduled walk: +7...13 % over variant 3 on scenes
// with 2 to 101 objects and, with the inner body's weight at 3 (scene_view), +2.5 % on a one-object scene (profiles/r02_sched_weight_sweep.txt).
// The kind-scheduled bodies address records through 32-bit byte offsets (pt_kernels_extend.hip rec32): scenes with 2^26 or more records in
// one array (4 GB of nodes) take the per-lane loop of variant 2, which addresses with 64 bits.
bool fits_32bit_offsets(const pt_ctx* c) { return c->n_tris + c->n_objects + c->top_nodes < (1ll << 26) && c->n_wide < (1ll << 26); }
int extend_variant_of(const pt_ctx* c) {
    const int v = c->extend_variant >= 0 ? c->extend_variant : 4;
    return fits_32bit_offsets(c) ? v : std::min(v, 2);
}

// PT_TRAVERSAL_AUTO: the wide walk pays where the binary nodes do not stay in the caches -- they are 64 B per triangle, and the chip
// has 8 x 4 MB of L2 in front of a 256 MB memory-side cache (measured, profiles/r03_mode_crossover.txt: 32 MB of nodes, 100 instances of
// 500 k triangles, lose 20 % to the decode; 640 MB, the 10 M-triangle grid, gain).
constexpr int64_t PT_AUTO_WIDE_NODE_BYTES = 256ll << 20;
bool wide_wanted(const pt_ctx* c) {
    return c->traversal_mode == PT_TRAVERSAL_WIDE || (c->traversal_mode == PT_TRAVERSAL_AUTO && c->n_tris * 64 > PT_AUTO_WIDE_NODE_BYTES);
}
bool wide_in_use(const pt_ctx* c) { return c->d_wide && wide_wanted(c); }

SceneView scene_view(const pt_ctx* c) {
    SceneView v; v.nodes = c->d_blas_trav; v.objects = c->d_objects; v.tris = c->d_tri_records;
    v.wide = wide_in_use(c) ? c->d_wide : nullptr; v.objects_wide = v.wide ? c->d_objects_wide : nullptr;
    v.soa = c->d_soa; v.soa_stride = (int32_t)std::min<int64_t>(c->n_tris + c->n_objects, INT32_MAX);
    v.ext_mats = c->d_ext_mats; v.n_ext_mats = c->n_ext_mats; v.stack_bound = c->stack_bound;
    v.n_objects = c->n_objects; v.n_tris = (int32_t)std::min<int64_t>(c->n_tris, INT32_MAX);
    v.tlas_root = v.n_tris;
    v.sched_wi = c->n_objects > 1 ? 6 : 3; v.sched_wl = 9; v.sched_ws = 5;                   // profiles/r02_sched_weight_sweep.txt
    v.sched_ww = 3;                                                                          // profiles/r03_wide_weight_sweep.txt
    static const struct EnvWeights {                                                         // measurement override: PT_SCHED_W="inner,triangle,switch"
        int w[4] = {0, 0, 0, 0};
        EnvWeights() {
            const char* e = getenv("PT_SCHED_W"); int n = e ? sscanf(e, "%d,%d,%d,%d", &w[0], &w[1], &w[2], &w[3]) : 0;
            if (n < 3 || w[0] <= 0 || w[1] <= 0 || w[2] <= 0) w[0] = 0;
            if (n < 4 || w[3] <= 0) w[3] = 0;
        }
    } env;
    if (env.w[0] > 0) { v.sched_wi = env.w[0]; v.sched_wl = env.w[1]; v.sched_ws = env.w[2]; if (env.w[3] > 0) v.sched_ww = env.w[3]; }
    const bool top = c->d_objects_top && !v.wide;
    v.objects_top = top ? c->d_objects_top : nullptr; v.top_base = top ? v.n_tris + v.n_objects : INT32_MAX; v.top_nodes = top ? c->top_nodes : 0;
    return v;
}

// builders.js:174-192: sort -> radix tree -> up pass.  aabb: 2 quads per primitive.
int build_bvh(pt_ctx* c, const float4* d_aabb, uint32_t* d_keys, int64_t n, int32_t* d_perm_out, float4* d_nodes_out, int32_t* d_dfs_out = nullptr, uint8_t* d_depth_out = nullptr) {
    uint32_t* d_keys_sorted = nullptr; int32_t *d_parents = nullptr, *d_first = nullptr; char* d_tmp = nullptr; size_t tmp_bytes = 0;
    int rc = PT_OK; Tmp tmp;
    if ((rc = tmp.alloc(c, &d_keys_sorted, (size_t)n, false))) return rc;
    if ((rc = tmp.alloc(c, &d_parents, (size_t)(2 * n), true))) return rc;
    if (d_dfs_out && (rc = tmp.alloc(c, &d_first, (size_t)n, true))) return rc;
    HIPOK(c, hipMemsetAsync(d_nodes_out, 0, (size_t)n * 64, c->stream));   // WebGPU zero-fills BVH_BUFFER; f_1 is the arrival counter
    // radix-sort.js:162-500: stable by key, carrying the position (its init_idx pass is folded into the first radix pass)
    HIPOK(c, pt_sort_keys_index(nullptr, &tmp_bytes, d_keys, d_keys_sorted, d_perm_out, (size_t)n, c->stream));
    if ((rc = tmp.alloc(c, &d_tmp, std::max<size_t>(tmp_bytes, 16), false))) return rc;
    HIPOK(c, pt_sort_keys_index(d_tmp, &tmp_bytes, d_keys, d_keys_sorted, d_perm_out, (size_t)n, c->stream));
    ptk_radix_tree(d_keys_sorted, n, d_parents, d_first, c->stream);
    ptk_up_pass(d_perm_out, d_aabb, d_parents, n, d_nodes_out, c->stream);
    if (d_dfs_out) {
        HIPOK(c, hipMemsetAsync(d_dfs_out + n, 0xff, (PT_TOP_SLOTS + 1) * 4, c->stream));
        ptk_dfs_index(d_parents, d_first, n, d_dfs_out, d_dfs_out + n, d_depth_out, c->stream);
    }
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

// builders.js:153-214 buildMeshBVH
int build_mesh(pt_ctx* c, Mesh& m) {
    float4* d_aabb = nullptr; uint32_t* d_keys = nullptr; int rc; Tmp tmp;
    dfree(m.d_nodes); dfree(m.d_tris_sorted); dfree(m.d_perm); dfree(m.d_dfs); dfree(m.d_depth); m.d_nodes = m.d_tris_sorted = nullptr; m.d_perm = m.d_dfs = nullptr; m.d_depth = nullptr;   // a failed earlier attempt
    if ((rc = tmp.alloc(c, &d_aabb, (size_t)m.n * 2, false))) return rc;
    if ((rc = tmp.alloc(c, &d_keys, (size_t)m.n, false))) return rc;
    if ((rc = dalloc(c, &m.d_nodes, (size_t)m.n * 4, false))) return rc;
    if ((rc = dalloc(c, &m.d_tris_sorted, (size_t)m.n * 3, false))) return rc;
    if ((rc = dalloc(c, &m.d_perm, (size_t)m.n, false))) return rc;
    if ((rc = dalloc(c, &m.d_dfs, (size_t)m.n + PT_TOP_SLOTS + 1, false))) return rc;
    if ((rc = dalloc(c, &m.d_depth, (size_t)m.n, false))) return rc;
    ptk_tri_aabb_morton(m.d_tris_in, m.n, m.bounds, d_aabb, d_keys, c->stream);
    if ((rc = build_bvh(c, d_aabb, d_keys, m.n, m.d_perm, m.d_nodes, m.d_dfs, m.d_depth))) return rc;
    ptk_rearrange(m.d_tris_in, m.d_perm, m.n, m.d_tris_sorted, c->stream);
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipStreamSynchronize(c->stream));
    m.built = true;
    return PT_OK;
}

void free_scene(pt_ctx* c) {
    if (c->scene_borrowed) {
        c->d_wide = c->d_objects_wide = nullptr; c->n_wide = 0; c->d_soa = nullptr;
        c->d_blas_nodes = c->d_blas_tris = c->d_tlas_nodes = c->d_objects = c->d_tri_records = c->d_blas_trav = nullptr; c->d_tlas_perm = nullptr; c->scene_built = false;
        c->d_objects_top = nullptr; c->top_nodes = 0;
        return;
    }
    dfree(c->d_wide); dfree(c->d_objects_wide); c->d_wide = c->d_objects_wide = nullptr; c->n_wide = 0;
    dfree(c->d_soa); c->d_soa = nullptr;
    dfree(c->d_blas_nodes); dfree(c->d_blas_tris); dfree(c->d_tlas_nodes); dfree(c->d_objects); dfree(c->d_tlas_perm); dfree(c->d_tri_records); dfree(c->d_blas_trav);
    dfree(c->d_objects_top); c->d_objects_top = nullptr; c->top_nodes = 0;
    c->d_blas_nodes = c->d_blas_tris = c->d_tlas_nodes = c->d_objects = c->d_tri_records = c->d_blas_trav = nullptr; c->d_tlas_perm = nullptr; c->scene_built = false;
}
void free_render(pt_ctx* c) {
    for (auto& p : c->render_allocs) dfree(p.first);
    c->render_allocs.clear();
    if (c->h_hdr) { (void)hipHostFree(c->h_hdr); c->h_hdr = nullptr; }
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    c->events.clear();
    if (c->d_ext_mats) { dfree(c->d_ext_mats); c->d_ext_mats = nullptr; }
    c->n_ext_mats = 0;
    dfree(c->d_chk_buf); dfree(c->d_chk_spill); c->d_chk_buf = nullptr; c->d_chk_spill = nullptr; c->chk_cap_alloc = 0;
    c->render_ready = false; c->d_image = nullptr;
}

template <typename T> int ralloc(pt_ctx* c, T** p, size_t count) {
    int rc = dalloc(c, p, count, true);
    if (rc == PT_OK) c->render_allocs.push_back({*p, std::max<size_t>(count, 1) * sizeof(T)});
    return rc;
}

int read_back(pt_ctx* c, void* dst, const void* src, size_t bytes) {
    HIPOK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

void enqueue_step(pt_ctx* c, int count_visits, hipEvent_t ev0, hipEvent_t ev1) {
    SceneView sv = scene_view(c);
    int parity = (int)(c->step & 1);
    if (c->step == 0) ptk_first_logic(c->ps, c->rc, c->d_code, c->d_blk_counts, c->stream);
    // One compaction workgroup per count record while the step has little to do -- the first steps (camera rays only) and
    // the ramp-down, as far as the host knows it (pt_render_run sees headers up to two batches late) -- else four.  When
    // several pools run together (pt_render_run_many) always one: the other pools' launches hide this one's latencies,
    // so the shape that does less work wins (+2 %).
    int fat = c->overlapped || c->step < 6 || (c->known_rays >= 0 && 2 * c->known_rays < c->rc.num_paths);
    static const int shape = getenv("PT_COMPACT_SHAPE") ? atoi(getenv("PT_COMPACT_SHAPE")) : 0;   // measurement: 1 = always four per record, 2 = always one
    if (shape) fat = shape == 2;
    ptk_count_compact_generate_shade(c->ps, c->rc, sv, c->d_code, c->d_blk_counts, parity, c->d_hdr, c->d_cam_q, c->d_mat_q, fat, c->stream);
    if (ev0) (void)hipEventRecord(ev0, c->stream);
    ptk_extend_queue(c->ps, c->rc, sv, c->d_hdr, c->d_cam_q, c->d_mat_q, c->d_spill, count_visits, extend_variant_of(c), nullptr, c->d_image, c->d_code,
                     (unsigned long long*)c->d_blk_counts + (parity ^ 1), (int)std::min<int64_t>(c->rays_cap, c->rc.num_paths), c->stream);
    if (ev1) (void)hipEventRecord(ev1, c->stream);
    c->step++;
}

// tolerance mode: the 4-wide collapse of every used mesh's packed BVH (built on demand, dropped with the scene) and the copy of
// the objects whose roots point into it
int build_wide(pt_ctx* c) {
    if (c->d_wide) return PT_OK;
    int rc; Tmp tmp;
    // pass 1: which binary nodes become wide nodes, and their depth-first rank inside their mesh
    std::vector<int32_t*> ranks; std::vector<int32_t> base; std::vector<int32_t> mesh_root, wide_root;
    int64_t total = 0;
    for (Mesh& m : c->meshes) {
        if (!m.used) continue;
        int32_t *d_rank = nullptr, *d_tiles = nullptr;
        if ((rc = tmp.alloc(c, &d_rank, (size_t)m.n, false)) || (rc = tmp.alloc(c, &d_tiles, (size_t)((m.n + 1023) / 1024 + 2), false))) return rc;
        const int64_t ntiles = (m.n + 1023) / 1024;
        ptk_wide_flags(m.d_dfs, m.d_depth, m.n, d_rank, c->stream);
        ptk_exclusive_scan(d_rank, m.n, d_tiles, d_tiles + ntiles, c->stream);
        HIPOK(c, hipGetLastError());
        int32_t count = 0;
        if ((rc = read_back(c, &count, d_tiles + ntiles, 4))) return rc;
        ranks.push_back(d_rank); base.push_back((int32_t)total);
        mesh_root.push_back((int32_t)m.tri_offset); wide_root.push_back((int32_t)total);      // a mesh's root has rank 0
        total += count;
    }
    if (total >= (1ll << 31)) return fail(c, PT_ERR_ARG, "tolerance mode: more than 2^31 wide nodes");
    if ((rc = dalloc(c, &c->d_wide, (size_t)std::max<int64_t>(total, 1) * PT_WIDE_QUADS, false))) return rc;
    if ((rc = dalloc(c, &c->d_objects_wide, (size_t)c->n_objects * 8, false))) return rc;
    size_t u = 0;
    for (Mesh& m : c->meshes)
        if (m.used) { ptk_collapse_wide(c->d_blas_nodes, m.tri_offset, m.n, m.d_dfs, m.d_depth, ranks[u], base[u], c->d_wide, c->stream); u++; }
    int32_t *d_mesh_root = nullptr, *d_wide_root = nullptr;
    if ((rc = tmp.alloc(c, &d_mesh_root, mesh_root.size(), false)) || (rc = tmp.alloc(c, &d_wide_root, wide_root.size(), false))) return rc;
    HIPOK(c, hipMemcpyAsync(d_mesh_root, mesh_root.data(), mesh_root.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPOK(c, hipMemcpyAsync(d_wide_root, wide_root.data(), wide_root.size() * 4, hipMemcpyHostToDevice, c->stream));
    ptk_objects_top(c->d_objects, c->n_objects, d_mesh_root, d_wide_root, (int)mesh_root.size(), c->d_objects_wide, c->stream);
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipStreamSynchronize(c->stream));
    c->n_wide = total;
    return PT_OK;
}

// extend variant 7: the structure-of-arrays copy of the walk's nodes, made when first asked for
int ensure_soa(pt_ctx* c) {
    if (c->d_soa || !c->scene_built || c->scene_borrowed) return PT_OK;
    const int64_t n = c->n_tris + c->n_objects; int rc;
    if ((rc = dalloc(c, &c->d_soa, (size_t)n * 4, false))) return rc;
    ptk_soa_split(c->d_blas_trav, n, c->d_soa, c->stream);
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

// what the host learns from a header it has read back
void note_header(pt_ctx* c, const StepHeader& h) {
    c->known_rays = h.n_rays;
    if (h.pixel_counter >= c->rc.ticket_limit) c->rays_cap = std::min<int64_t>(c->rays_cap, h.n_rays);
}

// fresh StepHeader: no step taken, ticket counter at this GPU's first ticket
int write_header(pt_ctx* c) {
    HIPOK(c, hipStreamSynchronize(c->stream));
    {   // tolerance mode: the re-walk period is fixed for the render that starts here
        RenderConst& rc = c->rc;
        rc.check_period = (wide_in_use(c) && fits_32bit_offsets(c)) ? c->check_period : 0;   // the re-walk is a kind-scheduled walk
        rc.chk_cap = rc.check_period > 0 ? ((rc.num_paths + 63) / 64 + rc.check_period - 1) / rc.check_period : 0;
        if (rc.chk_cap > c->chk_cap_alloc) {
            dfree(c->d_chk_buf); dfree(c->d_chk_spill); c->d_chk_buf = nullptr; c->d_chk_spill = nullptr; c->chk_cap_alloc = 0;
            int r;
            if ((r = dalloc(c, &c->d_chk_buf, (size_t)2 * rc.chk_cap * 9 * 64, true)) || (r = dalloc(c, &c->d_chk_spill, (size_t)rc.chk_cap * 64 * PT_STACK_SIZE, false))) return r;
            c->chk_cap_alloc = rc.chk_cap;
        }
        rc.chk_buf = c->d_chk_buf; rc.chk_spill = c->d_chk_spill;
    }
    StepHeader h; memset(&h, 0, sizeof h);
    h.first_empty_step = -1; h.pixel_counter = c->ticket_begin; h.ticket_next[0] = c->ticket_begin;
    *c->h_hdr = h;
    HIPOK(c, hipMemcpyAsync(c->d_hdr, c->h_hdr, sizeof h, hipMemcpyHostToDevice, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    c->step = 0; c->known_rays = -1; c->rays_cap = INT64_MAX;
    return PT_OK;
}

}  // namespace

// for pt_comm.hip (struct pt_ctx is private to this file)
int pt_ctx_device(const pt_ctx* c) { return c->device; }
hipStream_t pt_ctx_stream(const pt_ctx* c) { return c->stream; }
int pt_ctx_fail(pt_ctx* c, int code, const std::string& msg) { return fail(c, code, msg); }

extern "C" {

int pt_create(int device_id, pt_ctx** out) {
    if (!out) return fail(nullptr, PT_ERR_ARG, "pt_create: out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(nullptr, PT_ERR_NOGPU, "pt_create: no HIP device visible");
    if (device_id < 0 || device_id >= n || device_id >= PT_MAX_DEVICES) return fail(nullptr, PT_ERR_ARG, "pt_create: device_id out of range");
    pt_ctx* c = new pt_ctx();
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) {
        delete c; return fail(nullptr, PT_ERR_HIP, "pt_create: hipSetDevice/hipStreamCreate failed");
    }
    if (hipError_t e = scratch_address(device_id, &c->d_scratch); e != hipSuccess) {
        (void)hipStreamDestroy(c->stream); delete c;
        return fail(nullptr, PT_ERR_HIP, std::string("pt_create: hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    }
    c->own_stream = true;
    c->extend_variant = -1;  // automatic: variant 4 (kind-scheduled walk)
    if (const char* v = getenv("PT_EXTEND_VARIANT")) {       // measurement override; anything but a variant number is refused, not guessed at
        char* end = nullptr; long n = strtol(v, &end, 10);
        if (end == v || *end != 0 || n < -1 || n >= ptk_extend_num_variants()) {
            (void)hipStreamDestroy(c->stream); delete c;
            return fail(nullptr, PT_ERR_ARG, std::string("pt_create: PT_EXTEND_VARIANT=") + v + " is not a variant (-1 .. " + std::to_string(ptk_extend_num_variants() - 1) + ")");
        }
        c->extend_variant = (int)n;
    }
    *out = c;
    return PT_OK;
}

int pt_create_sharing_scene(pt_ctx* src, pt_ctx** out) {
    if (!src || !out) return fail(src, PT_ERR_ARG, "pt_create_sharing_scene: null argument");
    if (!src->scene_built) return fail(src, PT_ERR_STATE, "scene has not been built yet");
    int r = pt_create(src->device, out);
    if (r) return r;
    pt_ctx* c = *out;
    c->scene_borrowed = true; c->scene_built = true; c->extend_variant = src->extend_variant;
    c->n_tris = src->n_tris; c->n_objects = src->n_objects; c->n_meshes_used = src->n_meshes_used; c->stack_bound = src->stack_bound;
    c->d_blas_nodes = src->d_blas_nodes; c->d_blas_tris = src->d_blas_tris; c->d_tlas_nodes = src->d_tlas_nodes; c->d_objects = src->d_objects;
    c->d_tlas_perm = src->d_tlas_perm; c->d_tri_records = src->d_tri_records; c->d_blas_trav = src->d_blas_trav; c->d_soa = src->d_soa; c->d_objects_top = src->d_objects_top; c->top_nodes = src->top_nodes; c->d_wide = src->d_wide; c->d_objects_wide = src->d_objects_wide; c->n_wide = src->n_wide; c->traversal_mode = src->traversal_mode; c->check_period = src->check_period;
    c->lender = src; src->borrowers.push_back(c);
    return PT_OK;
}

int pt_destroy(pt_ctx* c) {
    if (!c) return PT_OK;
    if (!c->borrowers.empty()) return fail(c, PT_ERR_STATE, "pt_destroy: " + std::to_string(c->borrowers.size()) + " context(s) still borrow this context's scene; destroy them first");
    if (c->lender) { auto& b = c->lender->borrowers; b.erase(std::remove(b.begin(), b.end(), c), b.end()); }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_render(c); free_scene(c);
    for (Mesh& m : c->meshes) { dfree(m.d_tris_in); dfree(m.d_nodes); dfree(m.d_tris_sorted); dfree(m.d_perm); dfree(m.d_dfs); dfree(m.d_depth); }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PT_OK;
}

const char* pt_last_error(const pt_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }

int pt_set_stream(pt_ctx* c, void* hip_stream) {
    if (!c) return PT_ERR_ARG;
    HIPOK(c, hipSetDevice(c->device));
    HIPOK(c, hipStreamSynchronize(c->stream));
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    if (hip_stream) c->stream = (hipStream_t)hip_stream;
    else { HIPOK(c, hipStreamCreate(&c->stream)); c->own_stream = true; }
    return PT_OK;
}

int pt_device_pci_bus_id(pt_ctx* c, char* out, size_t cap) {
    if (!c || !out || cap < 16) return fail(c, PT_ERR_ARG, "pt_device_pci_bus_id: bad argument");
    HIPOK(c, hipDeviceGetPCIBusId(out, (int)cap, c->device));
    return PT_OK;
}

int pt_sync(pt_ctx* c) {
    if (!c) return PT_ERR_ARG;
    HIPOK(c, hipSetDevice(c->device));
    HIPOK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

// ------------------------------------------------------------------------------------------ scene
int pt_mesh_upload(pt_ctx* c, const float* tris12, int64_t n_tris, const float bounds[6]) {
    if (!c || !tris12 || !bounds) return fail(c, PT_ERR_ARG, "pt_mesh_upload: null argument");
    if (c->scene_borrowed) return fail(c, PT_ERR_STATE, "pt_mesh_upload: this context borrows another context's scene");
    if (n_tris < 1 || n_tris > (1ll << 30)) return fail(c, PT_ERR_ARG, "pt_mesh_upload: n_tris must be in [1, 2^30]");
    HIPOK(c, hipSetDevice(c->device));
    Mesh m; m.n = n_tris; memcpy(m.bounds, bounds, 24);
    int rc = dalloc(c, &m.d_tris_in, (size_t)n_tris * 3, false);
    if (rc) return rc;
    HIPOK(c, hipMemcpyAsync(m.d_tris_in, tris12, (size_t)n_tris * 48, hipMemcpyHostToDevice, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    c->meshes.push_back(m);
    return (int)c->meshes.size() - 1;
}

int pt_scene_build(pt_ctx* c, const pt_instance* inst, int n_inst) {
    if (!c || !inst) return fail(c, PT_ERR_ARG, "pt_scene_build: null argument");
    if (c->scene_borrowed) return fail(c, PT_ERR_STATE, "pt_scene_build: this context borrows another context's scene");
    if (n_inst < 1) return fail(c, PT_ERR_ARG, "pt_scene_build: the scene needs at least one instance");
    if (!c->borrowers.empty()) return fail(c, PT_ERR_STATE, "pt_scene_build: other contexts still borrow this context's scene (pt_create_sharing_scene); destroy them first");
    HIPOK(c, hipSetDevice(c->device));
    for (int i = 0; i < n_inst; i++)
        if (inst[i].mesh < 0 || inst[i].mesh >= (int)c->meshes.size())   // scene.js:524-526
            return fail(c, PT_ERR_ARG, "pt_scene_build: mesh with ID " + std::to_string(inst[i].mesh) + " does not exist");
    free_render(c); free_scene(c);
    // scene.js:17-29: used meshes in ascending id order, running triangle offset, BLAS cached per mesh
    for (Mesh& m : c->meshes) m.used = false;
    for (int i = 0; i < n_inst; i++) c->meshes[inst[i].mesh].used = true;
    int64_t off = 0; int used = 0; int rc;
    std::vector<int32_t> mesh_off(c->meshes.size(), 0);
    for (size_t k = 0; k < c->meshes.size(); k++) {
        Mesh& m = c->meshes[k];
        if (!m.used) continue;
        m.tri_offset = off; mesh_off[k] = (int32_t)off; used++;
        if (!m.built && (rc = build_mesh(c, m))) return rc;
        off += m.n;
    }
    if (off > (1ll << 30)) return fail(c, PT_ERR_ARG, "pt_scene_build: more than 2^30 packed triangles");
    c->n_tris = off; c->n_objects = n_inst; c->n_meshes_used = used;
    // pack-t-mesh.js: concatenate + rebase
    if ((rc = dalloc(c, &c->d_blas_nodes, (size_t)off * 4, false))) return rc;
    if ((rc = dalloc(c, &c->d_blas_tris, (size_t)off * 3, false))) return rc;
    for (Mesh& m : c->meshes)
        if (m.used) ptk_pack_mesh(m.d_nodes, m.d_tris_sorted, m.n, (int32_t)m.tri_offset, c->d_blas_nodes, c->d_blas_tris, c->stream);
    if ((rc = dalloc(c, &c->d_tri_records, (size_t)off * 3, false))) return rc;
    ptk_tri_records(c->d_blas_tris, off, c->d_tri_records, c->stream);
    static const bool trav_order = !(getenv("PT_TRAVERSAL_ORDER") && atoi(getenv("PT_TRAVERSAL_ORDER")) == 0);   // measurement: 0 = walk the reference-order buffer
    // Tree-top blocks: as many levels of every used mesh's tree as fit in PT_SCENE_LDS_QUADS together with the TLAS and the objects
    // (the extend variant that stages them in LDS); fewer than two levels are not worth a block.
    int top_levels = 0;
    if (trav_order)
        for (int k = PT_TOP_LEVELS; k >= 2 && !top_levels; k--)
            if (12 * (int64_t)n_inst + 4 * (int64_t)used * ((1 << k) - 1) <= PT_SCENE_LDS_QUADS) top_levels = k;
    const int top_slots = top_levels ? (1 << top_levels) - 1 : 0;
    std::vector<int32_t> mesh_root, top_root;
    {   // the walk's node array (SceneView::nodes): [BLAS | TLAS, rebased (further down, once it is built) | tree-top blocks]
        if (off + n_inst + (int64_t)used * top_slots >= (1ll << 31)) return fail(c, PT_ERR_ARG, "pt_scene_build: more than 2^31 nodes");
        c->top_nodes = top_levels ? used * top_slots : 0;
        if ((rc = dalloc(c, &c->d_blas_trav, (size_t)(off + n_inst + c->top_nodes) * 4, false))) return rc;
        if (!trav_order) HIPOK(c, hipMemcpyAsync(c->d_blas_trav, c->d_blas_nodes, (size_t)off * 64, hipMemcpyDeviceToDevice, c->stream));
        int u = 0;
        for (Mesh& m : c->meshes) {
            if (!m.used || !trav_order) continue;
            ptk_reorder_nodes(c->d_blas_nodes, m.d_dfs, m.tri_offset, m.n, c->d_blas_trav, c->stream);
            if (top_levels) {
                const int top_id = (int)(off + n_inst + (int64_t)u * top_slots);
                ptk_top_block(c->d_blas_nodes, m.d_dfs, m.d_dfs + m.n, m.tri_offset, m.n, top_levels, top_id, c->d_blas_trav, c->stream);
                mesh_root.push_back((int32_t)m.tri_offset); top_root.push_back(top_id);
            }
            u++;
        }
    }
    // scene.js:42-61 scene bounds; builders.js:97-151 buildTLAS
    float sb[6] = {1e30f, 1e30f, 1e30f, -1e30f, -1e30f, -1e30f};
    std::vector<float> aabbs((size_t)n_inst * 8, 0.0f);
    for (int i = 0; i < n_inst; i++)
        for (int k = 0; k < 3; k++) {
            aabbs[8 * i + k] = inst[i].aabb_min[k]; aabbs[8 * i + 4 + k] = inst[i].aabb_max[k];
            sb[k] = fminf(sb[k], inst[i].aabb_min[k]); sb[3 + k] = fmaxf(sb[3 + k], inst[i].aabb_max[k]);
        }
    float4* d_aabb = nullptr; uint32_t* d_keys = nullptr; void* d_inst = nullptr; int32_t* d_mesh_off = nullptr; Tmp tmp;
    if ((rc = tmp.alloc(c, &d_aabb, (size_t)n_inst * 2, false))) return rc;
    if ((rc = tmp.alloc(c, &d_keys, (size_t)n_inst, false))) return rc;
    if ((rc = dalloc(c, &c->d_tlas_nodes, (size_t)n_inst * 4, false))) return rc;
    if ((rc = dalloc(c, &c->d_tlas_perm, (size_t)n_inst, false))) return rc;
    if ((rc = dalloc(c, &c->d_objects, (size_t)n_inst * 8, false))) return rc;
    if ((rc = tmp.alloc(c, (char**)&d_inst, (size_t)n_inst * sizeof(pt_instance), false))) return rc;
    if ((rc = tmp.alloc(c, &d_mesh_off, mesh_off.size(), false))) return rc;
    HIPOK(c, hipMemcpyAsync(d_aabb, aabbs.data(), aabbs.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPOK(c, hipMemcpyAsync(d_inst, inst, (size_t)n_inst * sizeof(pt_instance), hipMemcpyHostToDevice, c->stream));
    HIPOK(c, hipMemcpyAsync(d_mesh_off, mesh_off.data(), mesh_off.size() * 4, hipMemcpyHostToDevice, c->stream));
    ptk_obj_morton(d_aabb, n_inst, sb, d_keys, c->stream);
    if ((rc = build_bvh(c, d_aabb, d_keys, n_inst, c->d_tlas_perm, c->d_tlas_nodes))) return rc;
    ptk_pack_objects(d_inst, c->d_tlas_perm, d_mesh_off, n_inst, c->d_objects, c->stream);
    ptk_tlas_rebase(c->d_tlas_nodes, n_inst, (int)off, c->d_blas_trav + 4 * off, c->stream);
    if (c->top_nodes) {
        int32_t *d_mesh_root = nullptr, *d_top_root = nullptr;
        if ((rc = tmp.alloc(c, &d_mesh_root, mesh_root.size(), false)) || (rc = tmp.alloc(c, &d_top_root, top_root.size(), false))) return rc;
        if ((rc = dalloc(c, &c->d_objects_top, (size_t)n_inst * 8, false))) return rc;
        HIPOK(c, hipMemcpyAsync(d_mesh_root, mesh_root.data(), mesh_root.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPOK(c, hipMemcpyAsync(d_top_root, top_root.data(), top_root.size() * 4, hipMemcpyHostToDevice, c->stream));
        ptk_objects_top(c->d_objects, n_inst, d_mesh_root, d_top_root, (int)mesh_root.size(), c->d_objects_top, c->stream);
    }
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipStreamSynchronize(c->stream));
    c->scene_built = true;
    if (c->extend_variant == 7 && (rc = ensure_soa(c))) return rc;
    if ((rc = scene_stack_bound(c))) return rc;
    return wide_wanted(c) ? build_wide(c) : PT_OK;
}

int pt_set_traversal_mode(pt_ctx* c, int mode) {
    if (!c || (mode != PT_TRAVERSAL_REFERENCE && mode != PT_TRAVERSAL_WIDE && mode != PT_TRAVERSAL_AUTO)) return fail(c, PT_ERR_ARG, "pt_set_traversal_mode: unknown mode");
    if (c->scene_borrowed) return fail(c, PT_ERR_STATE, "pt_set_traversal_mode: this context borrows another context's scene (set the mode there first)");
    if (!c->borrowers.empty() && mode != c->traversal_mode) return fail(c, PT_ERR_STATE, "pt_set_traversal_mode: other contexts still borrow this context's scene; destroy them first");
    HIPOK(c, hipSetDevice(c->device));
    HIPOK(c, hipStreamSynchronize(c->stream));
    c->traversal_mode = mode;
    return (c->scene_built && wide_wanted(c)) ? build_wide(c) : PT_OK;
}

int pt_get_traversal_mode(pt_ctx* c, int* mode_in_use) {
    if (!c || !mode_in_use) return fail(c, PT_ERR_ARG, "pt_get_traversal_mode: null argument");
    if (!c->scene_built) return fail(c, PT_ERR_STATE, "scene has not been built yet");
    *mode_in_use = wide_in_use(c) ? PT_TRAVERSAL_WIDE : PT_TRAVERSAL_REFERENCE;
    return PT_OK;
}

int pt_set_traversal_check(pt_ctx* c, int period) {
    if (!c || period < 0 || period > (1 << 20)) return fail(c, PT_ERR_ARG, "pt_set_traversal_check: period must be in [0, 2^20]");
    c->check_period = period;
    return PT_OK;
}

int pt_scene_get_info(pt_ctx* c, pt_scene_info* out) {
    if (!c || !out) return fail(c, PT_ERR_ARG, "pt_scene_get_info: null argument");
    if (!c->scene_built) return fail(c, PT_ERR_STATE, "scene has not been built yet");   // scene.js:149
    out->n_tris = c->n_tris; out->n_objects = c->n_objects; out->n_meshes_used = c->n_meshes_used;
    return PT_OK;
}

#define SCENE_GETTER(name, T, ptr, count)                                                                      \
    int name(pt_ctx* c, T* out, int64_t n) {                                                                    \
        if (!c || !out) return fail(c, PT_ERR_ARG, #name ": null argument");                                    \
        if (!c->scene_built) return fail(c, PT_ERR_STATE, "scene has not been built yet");                     \
        if (n != (int64_t)(count)) return fail(c, PT_ERR_ARG, #name ": n does not match the scene");            \
        HIPOK(c, hipSetDevice(c->device));                                                                      \
        return read_back(c, out, ptr, (size_t)n * sizeof(T));                                                   \
    }
SCENE_GETTER(pt_get_blas_nodes, pt_bvh_node, c->d_blas_nodes, c->n_tris)
SCENE_GETTER(pt_get_blas_tris, pt_triangle, c->d_blas_tris, c->n_tris)
SCENE_GETTER(pt_get_tlas_nodes, pt_bvh_node, c->d_tlas_nodes, c->n_objects)
SCENE_GETTER(pt_get_objects, pt_object, c->d_objects, c->n_objects)
SCENE_GETTER(pt_get_tlas_perm, int32_t, c->d_tlas_perm, c->n_objects)

int pt_scene_get_stack_bound(pt_ctx* c, int32_t* bound) {
    if (!c || !bound) return fail(c, PT_ERR_ARG, "pt_scene_get_stack_bound: null argument");
    if (!c->scene_built) return fail(c, PT_ERR_STATE, "scene has not been built yet");
    *bound = c->stack_bound;
    return PT_OK;
}
int pt_get_mesh_perm(pt_ctx* c, int mesh, int32_t* out, int64_t n) {
    if (!c || !out) return fail(c, PT_ERR_ARG, "pt_get_mesh_perm: null argument");
    if (mesh < 0 || mesh >= (int)c->meshes.size() || !c->meshes[mesh].built) return fail(c, PT_ERR_STATE, "pt_get_mesh_perm: mesh not built");
    if (n != c->meshes[mesh].n) return fail(c, PT_ERR_ARG, "pt_get_mesh_perm: n does not match the mesh");
    HIPOK(c, hipSetDevice(c->device));
    return read_back(c, out, c->meshes[mesh].d_perm, (size_t)n * 4);
}

int pt_trace_rays(pt_ctx* c, const float* o3, const float* d3, int64_t n, int32_t* hit_obj, int32_t* hit_tri, float* hit_dist,
                  int32_t* counts4) {
    if (!c || !o3 || !d3 || n < 0) return fail(c, PT_ERR_ARG, "pt_trace_rays: bad argument");
    if (!c->scene_built) return fail(c, PT_ERR_STATE, "scene has not been built yet");
    HIPOK(c, hipSetDevice(c->device));
    const int64_t CH = 1 << 20;
    int64_t cap = std::min<int64_t>(std::max<int64_t>(n, 1), CH);
    float4 *d_o = nullptr, *d_d = nullptr; int32_t *d_ho = nullptr, *d_ht = nullptr, *d_spill = nullptr; float* d_hd = nullptr; int4* d_cnt = nullptr;
    StepHeader* d_hdr = nullptr; int rc; Tmp tmp;
    if ((rc = tmp.alloc(c, &d_o, (size_t)cap, false)) || (rc = tmp.alloc(c, &d_d, (size_t)cap, false)) || (rc = tmp.alloc(c, &d_ho, (size_t)cap, false)) ||
        (rc = tmp.alloc(c, &d_ht, (size_t)cap, false)) || (rc = tmp.alloc(c, &d_hd, (size_t)cap, false)) || (rc = tmp.alloc(c, &d_cnt, (size_t)cap, false)) ||
        (rc = tmp.alloc(c, &d_spill, ptk_spill_ints(cap), false)) || (rc = tmp.alloc(c, &d_hdr, 1, true)))
        return rc;
    std::vector<float> stage((size_t)cap * 4);
    SceneView sv = scene_view(c);
    c->last_guard_hits = 0;
    for (int64_t base = 0; base < n; base += CH) {
        int64_t m = std::min(CH, n - base);
        for (int pass = 0; pass < 2; pass++) {
            const float* src = (pass ? d3 : o3) + 3 * base;
            for (int64_t i = 0; i < m; i++) { stage[4 * i] = src[3 * i]; stage[4 * i + 1] = src[3 * i + 1]; stage[4 * i + 2] = src[3 * i + 2]; stage[4 * i + 3] = 0.0f; }
            HIPOK(c, hipMemcpyAsync(pass ? d_d : d_o, stage.data(), (size_t)m * 16, hipMemcpyHostToDevice, c->stream));
            HIPOK(c, hipStreamSynchronize(c->stream));
        }
        ptk_extend_direct(sv, (const float*)d_o, (const float*)d_d, 4, (int)m, d_ho, d_ht, d_hd, counts4 ? d_cnt : nullptr, d_spill, d_hdr, c->stream);
        HIPOK(c, hipGetLastError());
        if (hit_obj) HIPOK(c, hipMemcpyAsync(hit_obj + base, d_ho, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        if (hit_tri) HIPOK(c, hipMemcpyAsync(hit_tri + base, d_ht, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        if (hit_dist) HIPOK(c, hipMemcpyAsync(hit_dist + base, d_hd, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
        if (counts4) HIPOK(c, hipMemcpyAsync(counts4 + 4 * base, d_cnt, (size_t)m * 16, hipMemcpyDeviceToHost, c->stream));
        HIPOK(c, hipStreamSynchronize(c->stream));
    }
    unsigned long long guard_hits = 0;   // the kernel adds every ray its stack / iteration guards stopped to StepHeader::overflow
    if ((rc = read_back(c, &guard_hits, &d_hdr->overflow, sizeof guard_hits))) return rc;
    c->last_guard_hits = (int64_t)guard_hits;
    return PT_OK;
}

// No hipMalloc / hipFree / allocating call below this line of the function: the spill and the guard counter live in `scratch`
// (through c->d_scratch, this GPU's address of it), inputs and outputs are the caller's device memory.  Nothing is waited for.
int pt_trace_rays_noalloc(pt_ctx* c, const float* o3_dev, const float* d3_dev, int64_t n, int32_t* hit_obj_dev, int32_t* hit_tri_dev,
                          float* hit_dist_dev, int32_t* counts4_dev, uint64_t* guard_hits_dev) {
    if (!c || !o3_dev || !d3_dev || n < 0) return fail(c, PT_ERR_ARG, "pt_trace_rays_noalloc: bad argument");
    if (n > PT_TRACE_NOALLOC_MAX_RAYS)
        return fail(c, PT_ERR_ARG, "pt_trace_rays_noalloc: " + std::to_string(n) + " rays, at most " + std::to_string(PT_TRACE_NOALLOC_MAX_RAYS) + " per call (split the batch, or use pt_trace_rays)");
    if (((uintptr_t)counts4_dev & 15) != 0) return fail(c, PT_ERR_ARG, "pt_trace_rays_noalloc: counts4_dev must be 16-byte aligned");
    if (!c->scene_built) return fail(c, PT_ERR_STATE, "scene has not been built yet");
    HIPOK(c, hipSetDevice(c->device));
    StepHeader* hdr = (StepHeader*)c->d_scratch;
    int32_t* spill = (int32_t*)((unsigned char*)c->d_scratch + PT_NOALLOC_HDR_BYTES);
    HIPOK(c, hipMemsetAsync(hdr, 0, sizeof(StepHeader), c->stream));    // the array keeps the last call's contents: the guard counter starts at 0
    SceneView sv = scene_view(c);
    for (int64_t base = 0; base < n; base += PT_NOALLOC_CHUNK) {       // chunks run one after the other on the stream and share the spill
        int64_t m = std::min<int64_t>(PT_NOALLOC_CHUNK, n - base);
        ptk_extend_direct(sv, o3_dev + 3 * base, d3_dev + 3 * base, 3, (int)m, hit_obj_dev ? hit_obj_dev + base : nullptr,
                          hit_tri_dev ? hit_tri_dev + base : nullptr, hit_dist_dev ? hit_dist_dev + base : nullptr,
                          counts4_dev ? (int4*)counts4_dev + base : nullptr, spill, hdr, c->stream);
    }
    HIPOK(c, hipGetLastError());
    if (guard_hits_dev) HIPOK(c, hipMemcpyAsync(guard_hits_dev, &hdr->overflow, sizeof(uint64_t), hipMemcpyDeviceToDevice, c->stream));
    return PT_OK;
}

int pt_trace_last_guard_hits(pt_ctx* c, int64_t* out) {
    if (!c || !out) return fail(c, PT_ERR_ARG, "pt_trace_last_guard_hits: null argument");
    *out = c->last_guard_hits;
    return PT_OK;
}

// ----------------------------------------------------------------------------------------- render
int pt_render_init(pt_ctx* c, const pt_render_params* p) {
    if (!c || !p) return fail(c, PT_ERR_ARG, "pt_render_init: null argument");
    if (!c->scene_built) return fail(c, PT_ERR_STATE, "scene has not been built yet");   // scene.js:149 via path-tracer.js:209
    if (p->width < 1 || p->height < 1 || (int64_t)p->width * p->height > (1ll << 29) || p->spp < 1)
        return fail(c, PT_ERR_ARG, "pt_render_init: bad image size or spp");
    int64_t np = p->num_paths > 0 ? p->num_paths : 1000000;   // path-tracer.js:4
    if (np > (1ll << 28)) return fail(c, PT_ERR_ARG, "pt_render_init: num_paths too large");
    for (pt_ctx* b : c->borrowers)    // a pool context that adds into this context's own image would be left with a dangling pointer
        if (c->render_ready && c->own_image && b->render_ready && b->d_image == c->d_image)
            return fail(c, PT_ERR_STATE, "pt_render_init: a context made by pt_create_sharing_scene still accumulates into this context's image; destroy it (or give it another image) first");
    HIPOK(c, hipSetDevice(c->device));
    HIPOK(c, hipStreamSynchronize(c->stream));
    free_render(c);
    RenderConst& rc = c->rc;
    rc.width = p->width; rc.height = p->height; rc.num_paths = (int32_t)np; rc.spp = p->spp; rc.fov_deg = p->fov_deg;
    for (int k = 0; k < 3; k++) { rc.cam_pos[k] = p->cam_pos[k]; rc.cam_look_at[k] = p->cam_look_at[k]; }
    int64_t total = ((int64_t)p->width * p->height * p->spp + PT_SAMPLES_PER_TICKET - 1) / PT_SAMPLES_PER_TICKET;   // logic.js:177
    rc.slot_base = p->slot_base;
    rc.ticket_limit = p->ticket_end >= 0 ? p->ticket_end : total;
    rc.max_nb = 20; rc.seed_mix = 0u;            // logic.js:111, :77 -- the reference's rules until pt_render_set_extensions says otherwise
    c->n_ext_mats = 0;
    c->ticket_begin = p->ticket_end >= 0 ? p->ticket_begin : 0;
    int r; size_t n = (size_t)np;
    PathState& ps = c->ps;   // zero-filled like a new GPUBuffer (path-tracer.js:271-275)
    if ((r = ralloc(c, &ps.pixel_index, n)) || (r = ralloc(c, &ps.state, n)) || (r = ralloc(c, &ps.random_seed, n)) ||
        (r = ralloc(c, &ps.path_throughput, n)) || (r = ralloc(c, &ps.material_throughput_pdf, n)) || (r = ralloc(c, &ps.path_o, n)) ||
        (r = ralloc(c, &ps.path_d, n)) || (r = ralloc(c, &ps.hit, n)) || (r = ralloc(c, &ps.ticket_acc, n)))
        return r;
    c->nblocks = ptk_step_blocks((int)np);
    if ((r = ralloc(c, &c->d_code, (size_t)c->nblocks * PT_STEP_BLOCK)) || (r = ralloc(c, &c->d_blk_counts, (size_t)c->nblocks)) ||
        (r = ralloc(c, &c->d_cam_q, n)) || (r = ralloc(c, &c->d_mat_q, n)) || (r = ralloc(c, &c->d_spill, ptk_spill_ints(np))) ||
        (r = ralloc(c, &c->d_hdr, 1)))
        return r;
    size_t img_floats = (size_t)p->width * p->height * 4;
    if (p->image_dev) { c->d_image = (float*)p->image_dev; c->own_image = false; HIPOK(c, hipMemsetAsync(c->d_image, 0, img_floats * 4, c->stream)); }
    else { if ((r = ralloc(c, &c->d_image, img_floats))) return r; c->own_image = true; }
    HIPOK(c, hipHostMalloc((void**)&c->h_hdr, 2 * sizeof(StepHeader), hipHostMallocDefault));
    {   // the pixel-independent part of the camera: evaluated by the device (same arithmetic as every other kernel), kept in rc
        CameraFrame* d_cf = nullptr; Tmp tmp;
        if ((r = tmp.alloc(c, &d_cf, 1, true))) return r;
        ptk_camera_frame(rc, d_cf, c->stream);
        HIPOK(c, hipGetLastError());
        if ((r = read_back(c, &rc.frame, d_cf, sizeof(CameraFrame)))) return r;
    }
    c->render_ready = true;
    return write_header(c);
}

int pt_render_set_extensions(pt_ctx* c, const pt_render_ext* e) {
    if (!c || !e) return fail(c, PT_ERR_ARG, "pt_render_set_extensions: null argument");
    if (!c->render_ready) return fail(c, PT_ERR_STATE, "pt_render_set_extensions: pt_render_init has not been called");
    if (c->step != 0) return fail(c, PT_ERR_STATE, "pt_render_set_extensions: the render has started; call pt_render_reset first");
    if (e->max_nb < 0 || e->max_nb > PT_MAX_NB_LIMIT || e->n_materials < 0 || e->n_materials > PT_MAX_EXT_MATERIALS) return fail(c, PT_ERR_ARG, "pt_render_set_extensions: bad max_nb or n_materials");
    for (int k = 0; k < e->n_materials; k++)
        if (e->materials[k].type < 0 || e->materials[k].type > PT_MAT_GLOSSY) return fail(c, PT_ERR_ARG, "pt_render_set_extensions: unknown material type");
    HIPOK(c, hipSetDevice(c->device));
    c->rc.max_nb = e->max_nb > 0 ? e->max_nb : 20;
    c->rc.seed_mix = e->seed * 0x9E3779B1u;
    if (e->n_materials > 0) {
        int r;
        if (!c->d_ext_mats && (r = dalloc(c, &c->d_ext_mats, (size_t)2 * PT_MAX_EXT_MATERIALS, true))) return r;   // survives pt_render_reset
        HIPOK(c, hipMemcpyAsync(c->d_ext_mats, e->materials, sizeof(pt_material) * e->n_materials, hipMemcpyHostToDevice, c->stream));
        HIPOK(c, hipStreamSynchronize(c->stream));
    }
    c->n_ext_mats = e->n_materials;
    return PT_OK;
}

int pt_render_reset(pt_ctx* c) {
    if (!c) return PT_ERR_ARG;
    if (!c->render_ready) return fail(c, PT_ERR_STATE, "pt_render_reset: pt_render_init has not been called");
    HIPOK(c, hipSetDevice(c->device));
    for (auto& a : c->render_allocs) HIPOK(c, hipMemsetAsync(a.first, 0, a.second, c->stream));
    if (!c->own_image) HIPOK(c, hipMemsetAsync(c->d_image, 0, (size_t)c->rc.width * c->rc.height * 16, c->stream));
    return write_header(c);
}

int pt_render_step(pt_ctx* c, pt_step_stats* stats) {
    if (!c) return PT_ERR_ARG;
    if (!c->render_ready) return fail(c, PT_ERR_STATE, "pt_render_step: pt_render_init has not been called");
    HIPOK(c, hipSetDevice(c->device));
    enqueue_step(c, 0, nullptr, nullptr);
    HIPOK(c, hipGetLastError());
    if (stats) {
        int r = read_back(c, c->h_hdr, c->d_hdr, sizeof(StepHeader));
        if (r) return r;
        stats->n_camera = c->h_hdr->n_camera; stats->n_material = c->h_hdr->n_material; stats->n_rays = c->h_hdr->n_rays;
        note_header(c, *c->h_hdr);
        stats->step_index = c->step - 1;
    }
    return PT_OK;
}

// pt_render_run over n contexts at once, from one host thread.  Batches of POLL steps are issued one ahead of the host's
// look at the queue sizes: batch b+1 is on a context's stream before the host waits for the header copy that followed its
// batch b, so the GPU never idles for the read-back (a render ends with up to 2*POLL - 1 empty steps instead).  With
// several contexts (pt_create_sharing_scene) the batches are issued in turn on the contexts' own streams, and one context's
// count / compaction launches run in the tail of another's extend launch.
// PT_RUN_TIME_EXTEND: one extend launch in every second batch (one in sixteen) is bracketed by events, at a position that
// rotates through the batch.  (Events around every launch were measured to cost 5.5 % of the render, around one in eight
// still 2 % with three pools: each record is a barrier packet between two kernels of a step.)
// PT_RUN_TIME_EXTEND_ALL: every extend launch is bracketed (for a pass that measures the kernel, not the render: bench.py's
// exclusive roofline pass, where the mean over all launches is what a kernel trace of the same command shows).
namespace {
const int POLL = 8;   // steps between looks at the queue sizes (the reference never looks: main.js:61-72)
const int EV_PER_BATCH = 2 * POLL + 1;   // events of one batch parity: POLL {begin, end} pairs, then "header copy landed"
struct RunState { pt_ctx* c; double extend_ms = 0.0; int64_t launches = 0, issued = 0, last = 0; unsigned had_timed[2] = {0u, 0u}; bool more = false, done = false; };
int run_issue(RunState& rs, int64_t batch, int64_t max_steps, int timed /* 0, PT_RUN_TIME_EXTEND or PT_RUN_TIME_EXTEND_ALL */, int count) {
    pt_ctx* c = rs.c;
    HIPOK(c, hipSetDevice(c->device));        // the contexts may sit on different GPUs (one host thread drives them all)
    int p = (int)(batch & 1); rs.had_timed[p] = 0u;
    hipEvent_t* ev = c->events.data() + EV_PER_BATCH * p;
    for (int k = 0; k < POLL && rs.issued < max_steps; k++, rs.issued++) {
        bool t = timed == PT_RUN_TIME_EXTEND_ALL || (timed && (batch & 1) == 0 && k == (int)((batch >> 1) % POLL));   // one launch in 2 * POLL
        if (t) rs.had_timed[p] |= 1u << k;
        enqueue_step(c, count, t ? ev[2 * k] : nullptr, t ? ev[2 * k + 1] : nullptr);
    }
    HIPOK(c, hipGetLastError());
    HIPOK(c, hipMemcpyAsync(c->h_hdr + p, c->d_hdr, sizeof(StepHeader), hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipEventRecord(ev[2 * POLL], c->stream));
    return PT_OK;
}
int run_collect(RunState& rs, int64_t batch) {
    pt_ctx* c = rs.c;
    int p = (int)(batch & 1);
    hipEvent_t* ev = c->events.data() + EV_PER_BATCH * p;
    HIPOK(c, hipSetDevice(c->device));
    HIPOK(c, hipEventSynchronize(ev[2 * POLL]));
    for (int k = 0; k < POLL; k++)
        if (rs.had_timed[p] >> k & 1u) { float ms = 0; HIPOK(c, hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1])); rs.extend_ms += ms; rs.launches++; }
    return PT_OK;
}
}  // namespace

int pt_render_run_many(pt_ctx** ctxs, int n, int64_t max_steps, int flags, pt_run_stats* out) {
    if (!ctxs || n < 1) return PT_ERR_ARG;
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return PT_ERR_ARG;
        if (!ctxs[i]->render_ready) return fail(ctxs[i], PT_ERR_STATE, "pt_render_run: pt_render_init has not been called");
        for (int k = 0; k < i; k++) if (ctxs[k] == ctxs[i]) return fail(ctxs[i], PT_ERR_ARG, "pt_render_run_many: the same context twice");
    }
    const int timed = (flags & PT_RUN_TIME_EXTEND_ALL) ? PT_RUN_TIME_EXTEND_ALL : (flags & PT_RUN_TIME_EXTEND); int count = (flags & PT_RUN_COUNT_BYTES) ? 1 : 0;
    if (max_steps <= 0) max_steps = INT64_MAX;
    std::vector<RunState> rs((size_t)n);
    int r;
    for (int i = 0; i < n; i++) {
        pt_ctx* c = rs[i].c = ctxs[i];
        c->overlapped = false;
        for (int k = 0; k < n; k++) if (k != i && ctxs[k]->device == c->device) c->overlapped = true;   // another pool on this GPU
        HIPOK(c, hipSetDevice(c->device));
        if (c->events.empty()) {   // per batch parity: {begin, end} per step of the batch for timed extend launches, then "header copy landed"
            c->events.resize(2 * EV_PER_BATCH);
            for (hipEvent_t& e : c->events) HIPOK(c, hipEventCreate(&e));
        }
        if ((r = run_issue(rs[i], 0, max_steps, timed, count))) return r;
    }
    for (int64_t b = 0;; b++) {
        for (RunState& s : rs) {
            if (s.done) continue;
            s.more = s.issued < max_steps;
            if (s.more) { if ((r = run_issue(s, b + 1, max_steps, timed, count))) return r; s.last = b + 1; }
        }
        bool all_done = true;
        for (RunState& s : rs) {
            if (s.done) continue;
            pt_ctx* c = s.c;
            if ((r = run_collect(s, b))) return r;
            note_header(c, c->h_hdr[b & 1]);
            if (c->h_hdr[b & 1].first_empty_step >= 0 || !s.more) {
                if (s.last > b && (r = run_collect(s, s.last))) return r;    // drain the batch issued ahead
                s.done = true;
            } else all_done = false;
        }
        if (all_done) break;
    }
    for (int i = 0; i < n; i++) {
        pt_ctx* c = ctxs[i];
        if (rs[i].last & 1) c->h_hdr[0] = c->h_hdr[1];             // the final header lives in h_hdr[0] for the code below
        if (c->h_hdr->first_empty_step < 0) {   // stopped by max_steps: bring the image up to date for the caller
            HIPOK(c, hipSetDevice(c->device));
            ptk_flush_tickets(c->ps, c->rc.num_paths, c->d_image, c->stream);
            HIPOK(c, hipStreamSynchronize(c->stream));
        }
        if (out) {
            const StepHeader& h = *c->h_hdr; pt_run_stats* o = out + i;
            o->steps = h.first_empty_step >= 0 ? h.first_empty_step + 1 : h.step_index;
            o->total_rays = h.total_rays; o->total_samples = h.total_samples; o->stack_overflows = (int64_t)h.overflow;
            o->extend_ms = rs[i].extend_ms; o->extend_launches = rs[i].launches;
            unsigned long long v[4] = {0, 0, 0, 0};
            for (int sl = 0; sl < 64; sl++) for (int k = 0; k < 4; k++) v[k] += h.visits[sl][k];
            o->traversal_bytes = count ? (int64_t)(44 * h.total_rays + 64 * v[0] + 48 * v[1] + 128 * (v[2] + v[3])) : 0;
            o->wasted_rays = h.total_retired;
            o->checked_rays = (int64_t)h.chk_rays; o->mismatched_rays = (int64_t)h.chk_mismatch; o->flipped_rays = (int64_t)h.chk_flips;
            o->beyond_tolerance_rays = (int64_t)h.chk_beyond; o->lost_hit_rays = (int64_t)h.chk_lost;
            { float f; memcpy(&f, &h.chk_max_rel_bits, 4); o->max_rel_dist_diff = f; }
        }
    }
    return PT_OK;
}

int pt_render_run(pt_ctx* c, int64_t max_steps, int flags, pt_run_stats* out) {
    if (!c) return PT_ERR_ARG;
    return pt_render_run_many(&c, 1, max_steps, flags, out);
}

int pt_image_read(pt_ctx* c, float* rgba, int64_t n_floats) {
    if (!c || !rgba) return fail(c, PT_ERR_ARG, "pt_image_read: null argument");
    if (!c->render_ready) return fail(c, PT_ERR_STATE, "pt_image_read: pt_render_init has not been called");
    if (n_floats != (int64_t)c->rc.width * c->rc.height * 4) return fail(c, PT_ERR_ARG, "pt_image_read: size mismatch");
    HIPOK(c, hipSetDevice(c->device));
    ptk_flush_tickets(c->ps, c->rc.num_paths, c->d_image, c->stream);
    return read_back(c, rgba, c->d_image, (size_t)n_floats * 4);
}

int pt_image_flush(pt_ctx* c) {
    if (!c) return PT_ERR_ARG;
    if (!c->render_ready) return fail(c, PT_ERR_STATE, "pt_image_flush: pt_render_init has not been called");
    HIPOK(c, hipSetDevice(c->device));
    ptk_flush_tickets(c->ps, c->rc.num_paths, c->d_image, c->stream);
    HIPOK(c, hipGetLastError());
    return PT_OK;
}

int pt_image_dev_ptr(pt_ctx* c, void** dev_ptr) {
    if (!c || !dev_ptr) return fail(c, PT_ERR_ARG, "pt_image_dev_ptr: null argument");
    if (!c->render_ready) return fail(c, PT_ERR_STATE, "pt_image_dev_ptr: pt_render_init has not been called");
    *dev_ptr = c->d_image;
    return PT_OK;
}

int pt_image_size(pt_ctx* c, int32_t* width, int32_t* height) {
    if (!c || !width || !height) return fail(c, PT_ERR_ARG, "pt_image_size: null argument");
    if (!c->render_ready) return fail(c, PT_ERR_STATE, "pt_image_size: pt_render_init has not been called");
    *width = c->rc.width; *height = c->rc.height;
    return PT_OK;
}

int pt_path_state_read(pt_ctx* c, const char* field, void* out, size_t bytes) {
    if (!c || !field || !out) return fail(c, PT_ERR_ARG, "pt_path_state_read: null argument");
    if (!c->render_ready) return fail(c, PT_ERR_STATE, "pt_path_state_read: pt_render_init has not been called");
    HIPOK(c, hipSetDevice(c->device));
    const PathState& ps = c->ps; size_t n = (size_t)c->rc.num_paths; const void* src = nullptr; size_t need = 0;
    {   // members the device keeps packed (pt_internal.h PathState): unpacked into the reference's i32 / u32 arrays
        const bool nb = !strcmp(field, "num_bounces"), fl = !strcmp(field, "flags"), sip = !strcmp(field, "samples_in_pixel");
        const bool ho = !strcmp(field, "hit_obj"), ht = !strcmp(field, "hit_tri");
        if (nb || fl || sip || ho || ht) {
            if (bytes != n * 4) return fail(c, PT_ERR_ARG, "pt_path_state_read: size mismatch");
            std::vector<uint32_t> raw(n * ((ho || ht) ? 2 : 1));
            int r = read_back(c, raw.data(), (ho || ht) ? (const void*)ps.hit : (const void*)ps.state, raw.size() * 4);
            if (r) return r;
            uint32_t* o = (uint32_t*)out;
            for (size_t i = 0; i < n; i++)
                o[i] = ho ? raw[2 * i] : ht ? raw[2 * i + 1] : nb ? PT_ST_NB(raw[i]) : sip ? PT_ST_SIP(raw[i]) : (raw[i] & (PT_ST_RETIRED | 3u));
            return PT_OK;
        }
    }
    struct { const char* name; const void* p; size_t elem; } tab[] = {   // elem 12: a vec3f member, stored packed, returned at 16-byte stride
        {"pixel_index", ps.pixel_index, 4}, {"random_seed", ps.random_seed, 4},
        {"path_throughput", ps.path_throughput, 12}, {"material_throughput_pdf", ps.material_throughput_pdf, 16},
        {"path_o", ps.path_o, 12}, {"path_d", ps.path_d, 12}};
    for (auto& t : tab) if (!strcmp(field, t.name)) {
        if (t.elem == 12) {
            if (bytes != n * 16) return fail(c, PT_ERR_ARG, "pt_path_state_read: size mismatch");
            std::vector<float> packed(n * 3);
            int r = read_back(c, packed.data(), t.p, n * 12);
            if (r) return r;
            float* o4 = (float*)out;
            for (size_t i = 0; i < n; i++) { o4[4 * i] = packed[3 * i]; o4[4 * i + 1] = packed[3 * i + 1]; o4[4 * i + 2] = packed[3 * i + 2]; o4[4 * i + 3] = 0.0f; }
            return PT_OK;
        }
        src = t.p; need = n * t.elem;
    }
    if (!strcmp(field, "pixel_counter")) { src = &c->d_hdr->pixel_counter; need = 8; }
    if (!src) return fail(c, PT_ERR_ARG, std::string("pt_path_state_read: unknown field ") + field);
    if (bytes != need) return fail(c, PT_ERR_ARG, "pt_path_state_read: size mismatch");
    return read_back(c, out, src, need);
}

int pt_queue_read(pt_ctx* c, int which, int32_t* out, int64_t* n_io) {
    if (!c || !out || !n_io || which < 0 || which > 1) return fail(c, PT_ERR_ARG, "pt_queue_read: bad argument");
    if (!c->render_ready) return fail(c, PT_ERR_STATE, "pt_queue_read: pt_render_init has not been called");
    HIPOK(c, hipSetDevice(c->device));
    int r = read_back(c, c->h_hdr, c->d_hdr, sizeof(StepHeader));
    if (r) return r;
    int64_t cnt = which == 0 ? c->h_hdr->n_camera : c->h_hdr->n_material;
    if (cnt > *n_io) return fail(c, PT_ERR_ARG, "pt_queue_read: buffer too small");
    *n_io = cnt;
    return cnt ? read_back(c, out, which == 0 ? c->d_cam_q : c->d_mat_q, (size_t)cnt * 4) : PT_OK;
}

// ------------------------------------------------------------------------------- measurement hooks
int pt_extend_variants(pt_ctx* c, int variant, char* name_out, size_t name_cap) {
    if (name_out && name_cap) { strncpy(name_out, ptk_extend_variant_name(variant), name_cap - 1); name_out[name_cap - 1] = 0; }
    return ptk_extend_num_variants();
}

int pt_set_extend_variant(pt_ctx* c, int variant) {
    if (!c || variant < -1 || variant >= ptk_extend_num_variants()) return fail(c, PT_ERR_ARG, "pt_set_extend_variant: no such variant");
    c->extend_variant = variant;
    if (variant == 7) { HIPOK(c, hipSetDevice(c->device)); return ensure_soa(c); }
    return PT_OK;
}

int pt_extend_replay(pt_ctx* c, int variant, int reps, double* avg_us) {
    if (!c || !avg_us || reps < 1 || variant < 0 || variant >= ptk_extend_num_variants()) return fail(c, PT_ERR_ARG, "pt_extend_replay: bad argument");
    if (!c->render_ready || c->step == 0) return fail(c, PT_ERR_STATE, "pt_extend_replay: take at least one step first");
    HIPOK(c, hipSetDevice(c->device));
    if (variant == 7) { int r7 = ensure_soa(c); if (r7) return r7; }
    hipEvent_t e0, e1; HIPOK(c, hipEventCreate(&e0)); HIPOK(c, hipEventCreate(&e1));
    SceneView sv = scene_view(c);
    double total_ms = 0;
    for (int i = 0; i < reps; i++) {   // the kernel only reads path_o/path_d and rewrites the same hit_obj/hit_tri: idempotent
        HIPOK(c, hipEventRecord(e0, c->stream));
        ptk_extend_queue(c->ps, c->rc, sv, c->d_hdr, c->d_cam_q, c->d_mat_q, c->d_spill, 0, variant, nullptr, nullptr, nullptr, nullptr, c->rc.num_paths, c->stream);
        HIPOK(c, hipEventRecord(e1, c->stream));
        HIPOK(c, hipEventSynchronize(e1));
        float ms = 0; HIPOK(c, hipEventElapsedTime(&ms, e0, e1)); total_ms += ms;
    }
    HIPOK(c, hipGetLastError());
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = total_ms * 1e3 / reps;
    return PT_OK;
}

int pt_extend_pass_counts(pt_ctx* c, int64_t* out8 /* 12 entries */) {
    if (!c || !out8) return fail(c, PT_ERR_ARG, "pt_extend_pass_counts: null argument");
    if (!c->render_ready || !c->h_hdr) return fail(c, PT_ERR_STATE, "pt_extend_pass_counts: no render");
    const StepHeader& h = *c->h_hdr;
    for (int k = 0; k < 4; k++) { out8[k] = (int64_t)h.passes[k]; out8[4 + k] = 0; for (int sl = 0; sl < 64; sl++) out8[4 + k] += (int64_t)h.visits[sl][k]; }
    for (int k = 0; k < 4; k++) out8[8 + k] = (int64_t)h.lane_hist[k];
    return PT_OK;
}

int pt_extend_wave_log(pt_ctx* c, int variant, uint64_t* out, int64_t cap_waves, int64_t* n_waves) {
    if (!c || !out || !n_waves || variant < 0 || variant >= ptk_extend_num_variants()) return fail(c, PT_ERR_ARG, "pt_extend_wave_log: bad argument");
    if (!c->render_ready || c->step == 0) return fail(c, PT_ERR_STATE, "pt_extend_wave_log: take at least one step first");
    HIPOK(c, hipSetDevice(c->device));
    int64_t n = ptk_extend_waves(c->rc.num_paths, variant);
    if (n > cap_waves) return fail(c, PT_ERR_ARG, "pt_extend_wave_log: buffer too small");
    unsigned long long* d_log = nullptr; int r; Tmp tmp;
    if ((r = tmp.alloc(c, &d_log, (size_t)n * 4, true))) return r;
    SceneView sv = scene_view(c);
    ptk_extend_queue(c->ps, c->rc, sv, c->d_hdr, c->d_cam_q, c->d_mat_q, c->d_spill, 0, variant, d_log, nullptr, nullptr, nullptr, c->rc.num_paths, c->stream);
    HIPOK(c, hipGetLastError());
    r = read_back(c, out, d_log, (size_t)n * 32);
    *n_waves = n;
    return r;
}

// ---------------------------------------------------------------------------------------- display
int pt_tonemap_rgba8(pt_ctx* c, const void* image_dev, int32_t width, int32_t height, uint8_t* rgba8) {
    if (!c || !rgba8) return fail(c, PT_ERR_ARG, "pt_tonemap_rgba8: null argument");
    HIPOK(c, hipSetDevice(c->device));
    if (!image_dev) {
        if (!c->render_ready) return fail(c, PT_ERR_STATE, "pt_tonemap_rgba8: no image");
        image_dev = c->d_image; width = c->rc.width; height = c->rc.height;
    }
    if (c->render_ready && image_dev == c->d_image)      // a draw in the middle of a render shows every finished sample
        ptk_flush_tickets(c->ps, c->rc.num_paths, c->d_image, c->stream);
    if (width < 1 || height < 1) return fail(c, PT_ERR_ARG, "pt_tonemap_rgba8: bad size");
    int64_t n = (int64_t)width * height; uint32_t* d_out = nullptr; int r; Tmp tmp;
    if ((r = tmp.alloc(c, &d_out, (size_t)n, false))) return r;
    ptk_tonemap((const float4*)image_dev, n, d_out, c->stream);
    HIPOK(c, hipGetLastError());
    return read_back(c, rgba8, d_out, (size_t)n * 4);
}

}  // extern "C"
