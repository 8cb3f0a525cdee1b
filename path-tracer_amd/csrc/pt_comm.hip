// pt_comm.hip -- multi-GPU part of libpt_amd.so: the per-GPU accumulation buffers are added up by one RCCL
// sum-reduce over xGMI.  The path shards with no other exchange: every GPU holds the whole scene, works on its own pixel-ticket
// range with its own RNG streams (pt_render_params), and the image is sums of radiance + sample counts (logic.js:146-149), so
// adding the buffers gives exactly the single-GPU image's form.
// librccl is loaded on first use (dlopen), so that a one-GPU host pays nothing for it and the library loads without it.
#include "pt_amd.h"
#include "pt_internal.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

// accessors of pt_ctx that pt_api.hip provides (the struct is private to that file)
int pt_ctx_device(const pt_ctx* c);
hipStream_t pt_ctx_stream(const pt_ctx* c);
int pt_ctx_fail(pt_ctx* c, int code, const std::string& msg);

namespace {
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

bool load_rccl(std::string& why) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.so) return true;
    if (!g_rccl.error.empty()) { why = g_rccl.error; return false; }
    void* so = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((so = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!so) { g_rccl.error = why = std::string("librccl not found: ") + dlerror(); return false; }
    bool ok = true;
#define PT_SYM(field, sym) ok = ok && ((*(void**)&g_rccl.field = dlsym(so, sym)) != nullptr)
    PT_SYM(GetUniqueId, "ncclGetUniqueId"); PT_SYM(CommInitRank, "ncclCommInitRank"); PT_SYM(CommInitAll, "ncclCommInitAll");
    PT_SYM(CommDestroy, "ncclCommDestroy"); PT_SYM(Reduce, "ncclReduce"); PT_SYM(GroupStart, "ncclGroupStart");
    PT_SYM(GroupEnd, "ncclGroupEnd"); PT_SYM(GetErrorString, "ncclGetErrorString");
#undef PT_SYM
    if (!ok) { g_rccl.error = why = "librccl lacks an expected symbol"; dlclose(so); return false; }
    g_rccl.so = so;
    return true;
}
}  // namespace

struct pt_comm {
    std::vector<pt_ctx*> ctxs;         // the local contexts, one per GPU (one entry in the process-per-GPU form)
    std::vector<int> devs;             // their GPUs (kept for pt_comm_destroy, which may run after the contexts are gone)
    std::vector<ncclComm_t> comms;     // their communicators
    int world = 0, first_rank = 0;     // ranks of the local contexts: first_rank .. first_rank + ctxs.size() - 1
};

#define NCCLOK(c, expr)                                                                                              \
    do {                                                                                                             \
        ncclResult_t r_ = (expr);                                                                                    \
        if (r_ != ncclSuccess) return pt_ctx_fail(c, PT_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

extern "C" {

int pt_comm_unique_id(void* id128) {
    if (!id128) return PT_ERR_ARG;
    std::string why;
    if (!load_rccl(why)) return pt_ctx_fail(nullptr, PT_ERR_STATE, why);
    static_assert(sizeof(ncclUniqueId) == PT_COMM_ID_BYTES, "pt_amd.h PT_COMM_ID_BYTES");
    ncclUniqueId id;
    NCCLOK(nullptr, g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return PT_OK;
}

int pt_comm_create_rank(pt_ctx* ctx, const void* id128, int rank, int world, pt_comm** out) {
    if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) return pt_ctx_fail(ctx, PT_ERR_ARG, "pt_comm_create_rank: bad argument");
    *out = nullptr;
    std::string why;
    if (!load_rccl(why)) return pt_ctx_fail(ctx, PT_ERR_STATE, why);
    if (hipSetDevice(pt_ctx_device(ctx)) != hipSuccess) return pt_ctx_fail(ctx, PT_ERR_HIP, "pt_comm_create_rank: hipSetDevice failed");
    ncclUniqueId id; memcpy(&id, id128, sizeof id);
    ncclComm_t comm;
    NCCLOK(ctx, g_rccl.CommInitRank(&comm, world, id, rank));
    pt_comm* c = new pt_comm();
    c->ctxs = {ctx}; c->devs = {pt_ctx_device(ctx)}; c->comms = {comm}; c->world = world; c->first_rank = rank;
    *out = c;
    return PT_OK;
}

int pt_comm_create(pt_ctx** ctxs, int n, pt_comm** out) {
    if (!ctxs || n < 1 || !out) return pt_ctx_fail(nullptr, PT_ERR_ARG, "pt_comm_create: bad argument");
    *out = nullptr;
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return pt_ctx_fail(nullptr, PT_ERR_ARG, "pt_comm_create: null context");
        devs[(size_t)i] = pt_ctx_device(ctxs[i]);
        for (int k = 0; k < i; k++)
            if (devs[(size_t)k] == devs[(size_t)i]) return pt_ctx_fail(ctxs[0], PT_ERR_ARG, "pt_comm_create: two contexts on one GPU (one communicator rank per GPU)");
    }
    std::string why;
    if (!load_rccl(why)) return pt_ctx_fail(ctxs[0], PT_ERR_STATE, why);
    std::vector<ncclComm_t> comms((size_t)n);
    NCCLOK(ctxs[0], g_rccl.CommInitAll(comms.data(), n, devs.data()));
    pt_comm* c = new pt_comm();
    c->ctxs.assign(ctxs, ctxs + n); c->devs = devs; c->comms = comms; c->world = n; c->first_rank = 0;
    *out = c;
    return PT_OK;
}

// Adds every rank's accumulation buffer onto rank `root`'s, in place, on the contexts' own streams (so it is ordered behind
// the render that filled them), and waits for the local part.  Every local context must hold a render of the same size.
int pt_comm_reduce_image(pt_comm* comm, int root) {
    if (!comm || root < 0 || root >= comm->world) return pt_ctx_fail(comm && !comm->ctxs.empty() ? comm->ctxs[0] : nullptr, PT_ERR_ARG, "pt_comm_reduce_image: bad argument");
    const size_t n = comm->ctxs.size();
    std::vector<void*> img(n); int32_t w0 = 0, h0 = 0;
    for (size_t i = 0; i < n; i++) {
        int32_t w, h; int r;
        if ((r = pt_image_size(comm->ctxs[i], &w, &h)) || (r = pt_image_dev_ptr(comm->ctxs[i], &img[i]))) return r;
        if (i == 0) { w0 = w; h0 = h; }
        else if (w != w0 || h != h0) return pt_ctx_fail(comm->ctxs[i], PT_ERR_ARG, "pt_comm_reduce_image: the contexts' images differ in size");
        // unfinished tickets' partial sums belong to the image (a render that ran to its end has none)
        if ((r = pt_image_flush(comm->ctxs[i]))) return r;
    }
    const size_t count = (size_t)w0 * (size_t)h0 * 4;
    pt_ctx* c0 = comm->ctxs[0];
    NCCLOK(c0, g_rccl.GroupStart());
    for (size_t i = 0; i < n; i++) {
        if (hipSetDevice(pt_ctx_device(comm->ctxs[i])) != hipSuccess) return pt_ctx_fail(comm->ctxs[i], PT_ERR_HIP, "pt_comm_reduce_image: hipSetDevice failed");
        NCCLOK(comm->ctxs[i], g_rccl.Reduce(img[i], img[i], count, ncclFloat32, ncclSum, root, comm->comms[i], pt_ctx_stream(comm->ctxs[i])));
    }
    NCCLOK(c0, g_rccl.GroupEnd());
    for (size_t i = 0; i < n; i++) { int r = pt_sync(comm->ctxs[i]); if (r) return r; }
    return PT_OK;
}

void pt_comm_destroy(pt_comm* comm) {
    if (!comm) return;
    for (size_t i = 0; i < comm->comms.size(); i++) {
        (void)hipSetDevice(comm->devs[i]);
        (void)g_rccl.CommDestroy(comm->comms[i]);
    }
    delete comm;
}

}  // extern "C"
