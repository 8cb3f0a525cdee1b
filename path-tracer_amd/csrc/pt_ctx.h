// pt_ctx.h -- the context behind the C ABI (struct pt_ctx is opaque in include/pt_amd.h) and what the two halves of the host code share:
// pt_api.hip (context, scene, ray-tracing hooks) and pt_api_render.hip (render, image, measurement hooks, display).
#pragma once
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

struct Mesh {
    int64_t n = 0; float bounds[6] = {0, 0, 0, 0, 0, 0};
    float4* d_tris_in = nullptr;      // upload order (parse-obj.js order)
    float4* d_nodes = nullptr;        // per-mesh BLAS, child pointers relative to the mesh
    float4* d_tris_sorted = nullptr;  // Morton order
    int32_t* d_perm = nullptr;        // sorted position -> upload position
    int32_t* d_dfs = nullptr;         // inner node -> its number in depth-first preorder (traversal-order copy of the BLAS);
                                      // behind these n entries: PT_TOP_SLOTS + 1 ints, the nodes of the first levels by heap slot (k_dfs_index)
    uint8_t* d_depth = nullptr;       // inner node -> links between it and the root (the even depths become the wide nodes of the tolerance mode)
    int32_t max_inner_path = 0;       // inner nodes on the longest root-to-leaf path (k_dfs_index), 0 for a one-triangle mesh
    bool built = false, used = false; int64_t tri_offset = 0;
};

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

int pt_fail(pt_ctx* c, int code, const std::string& msg);    // records the message (on the context, or for pt_last_error(NULL)) and returns code
static inline int fail(pt_ctx* c, int code, const std::string& msg) { return pt_fail(c, code, msg); }

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
static inline void dfree(void* p) { if (p) (void)hipFree(p); }

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

// ---- shared between pt_api.hip and pt_api_render.hip ----
int read_back(pt_ctx* c, void* dst, const void* src, size_t bytes);      // copy to the host on the context's stream, waited for
SceneView scene_view(const pt_ctx* c);
int extend_variant_of(const pt_ctx* c);
bool wide_in_use(const pt_ctx* c);
bool fits_32bit_offsets(const pt_ctx* c);
int ensure_soa(pt_ctx* c);
void free_render(pt_ctx* c);
