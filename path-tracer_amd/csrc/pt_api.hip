// pt_api.hip -- host side of libpt_amd.so, first half: context, device memory, scene build, ray-tracing hooks of the C ABI.
// (Render, image, measurement hooks and display: pt_api_render.hip.)  Declarations and the reference interface each entry point
// replaces: include/pt_amd.h.
#include "pt_ctx.h"

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
}  // namespace

int pt_fail(pt_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; else g_err = msg; return code; }

// Most stack entries a ray can hold at once: every entry is the far child of an inner node on the path from the root to
// the node being visited (TLAS part below switch_pt, BLAS part above it), so the bound is the number of inner nodes on the
// longest root-to-leaf path of the TLAS plus that of the deepest BLAS.  The TLAS is walked on the host (up to 2^20 objects), a mesh's
// longest path comes from the device (k_dfs_index, read back when the mesh is built).  A bound within the LDS share lets the staged
// variant drop its spill path and its guards; a bound within PT_STACK_SIZE lets the default variant drop the guards (Stack GUARDED).
// A tree the walk cannot finish on (the degenerate one-primitive tree can point at itself) leaves the bound unknown.
static int tree_path_inner_nodes(const std::vector<pt_bvh_node>& nodes, int64_t first, int64_t count, int64_t root) {
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
static int scene_stack_bound(pt_ctx* c) {
    c->stack_bound = INT32_MAX;
    if (c->n_objects > (1 << 20)) return PT_OK;
    std::vector<pt_bvh_node> tlas((size_t)c->n_objects);
    int r;
    if ((r = read_back(c, tlas.data(), c->d_tlas_nodes, tlas.size() * sizeof(pt_bvh_node)))) return r;
    int64_t bound = tree_path_inner_nodes(tlas, 0, c->n_objects, 0), deepest = 0;
    for (const Mesh& m : c->meshes)
        if (m.used) deepest = std::max<int64_t>(deepest, m.n >= 2 ? m.max_inner_path : INT32_MAX);     // a one-triangle mesh is the degenerate tree
    c->stack_bound = (int32_t)std::min<int64_t>(bound + deepest, INT32_MAX);
    return PT_OK;
}

// 8 waves/SIMD, class of the longest wave first, a tiny scene staged in LDS, kind-scheduled walk: +7...13 % over variant 3 on scenes
// with 2 to 101 objects and, with the inner body's weight at 3 (scene_view), +2.5 % on a one-object scene (profiles/r02_sched_weight_sweep.txt).
// The kind-scheduled bodies address records through 32-bit byte offsets (pt_kernels_extend.hip rec32): scenes with 2^26 or more records in
// one array (4 GB of nodes) take the per-lane loop of variant 2, which addresses with 64 bits.
bool fits_32bit_offsets(const pt_ctx* c) { return c->n_tris + c->n_objects + c->top_nodes < (1ll << 26) && c->n_wide < (1ll << 26); }
// The library's choice (extend_variant < 0): the kind-scheduled walk; with more than one object the one whose triangle passes also serve
// the lanes at inner nodes (variant 9: +4 % on configs[2], +5 % on 100 instances, profiles/r03_extend_ab_fused_pass.txt; the one-object
// grid loses 1 % with it and keeps variant 4).  A scene small enough for the LDS stage takes the staged kernel either way.
int extend_variant_of(const pt_ctx* c) {
    const int v = c->extend_variant >= 0 ? c->extend_variant : (c->n_objects > 1 ? 9 : 4);
    return fits_32bit_offsets(c) ? v : std::min(v, 2);
}

// PT_TRAVERSAL_AUTO: the wide walk pays where a ray's walk inside an object is long -- it halves the dependent fetches of that walk and
// costs a decode per visit plus a fourth body in the scheduled loop.  Measured (profiles/r03_mode_crossover.txt): the one-mesh displaced
// grid gains at every size, +15 % at 500 k triangles (31 MB of binary nodes, cache-resident) to +31 % at 10 M; scenes of many objects
// lose -- 100 instances of 500 k triangles -8 %, the 82 k-triangle mesh over the floor -2 %, the Cornell box (whose reference walk runs
// out of LDS) -26 %.  The rule that separates them: at least 2^18 packed triangles per object.
bool wide_wanted(const pt_ctx* c) {
    return c->traversal_mode == PT_TRAVERSAL_WIDE || (c->traversal_mode == PT_TRAVERSAL_AUTO && c->n_tris >= ((int64_t)std::max(c->n_objects, 1) << 18));
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
    if (extend_variant_of(c) == 9 && !wide_in_use(c) && 12 * (int64_t)c->n_objects + 7 * c->n_tris > PT_SCENE_LDS_QUADS) {
        // the fused pass makes a triangle pass cheap in trips to memory: it is worth running for fewer waiting lanes; a scene of many
        // objects wants its switches sooner as well (profiles/r03_extend_ab_fused_pass.txt)
        if (c->n_objects > 8) { v.sched_wi = 4; v.sched_wl = 12; v.sched_ws = 8; }
        else                  { v.sched_wi = 6; v.sched_wl = 30; v.sched_ws = 5; }
    }
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
    if ((rc = read_back(c, &m.max_inner_path, m.d_dfs + m.n + PT_TOP_SLOTS, 4))) return rc;     // k_dfs_index's maximum (the stream is synchronised by the copy)
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
int read_back(pt_ctx* c, void* dst, const void* src, size_t bytes) {
    HIPOK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPOK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
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

// for pt_comm.hip (which does not include pt_ctx.h)
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
    c->extend_variant = -1;  // automatic: the kind-scheduled walk, variant 9 or 4 (extend_variant_of)
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

}  // extern "C"
