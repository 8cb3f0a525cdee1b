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

// accessors of pt_ctx that pt_api.hip provides (this file stays clear of pt_ctx.h)
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

// ---- loopback backend (PT_COMM_LOOPBACK=1 in the environment when the first communicator is made) ----------------------------------
// The same eight entry points for "ranks" that are contexts of ONE process, summed by a kernel of this file instead of RCCL.  It exists
// so that the semantics of pt_comm_reduce_image at world size > 1 -- what ends up where, what a second reduce adds -- can be tested on
// a box with one GPU (RCCL refuses two ranks on one device).  It is not a fallback: nothing selects it but that variable.
struct LoopOp { const float* send; float* recv; size_t count; int root, rank; hipStream_t stream; };
struct LoopWorld { int world = 0; std::vector<LoopOp> pending; };
struct LoopComm { LoopWorld* w; int rank; int device; };
thread_local bool g_loop_grouped = false;
thread_local std::vector<LoopComm*> g_loop_touched;

__global__ void k_loop_sum(float* __restrict__ recv, const float* const* __restrict__ sends, int n, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float acc = sends[0][i];
    for (int k = 1; k < n; k++) acc += sends[k][i];               // rank order: a fixed order, like a ring's
    recv[i] = acc;
}
ncclResult_t loop_flush(LoopWorld* w) {
    if ((int)w->pending.size() != w->world) return ncclInvalidUsage;     // every rank of the communicator takes part in a reduce
    std::vector<const float*> sends((size_t)w->world); const LoopOp* root = nullptr;
    for (const LoopOp& op : w->pending) { sends[(size_t)op.rank] = op.send; if (op.rank == op.root) root = &op; if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError; }
    if (!root) return ncclInvalidArgument;
    const float** d_ptrs = nullptr;
    if (hipMalloc((void**)&d_ptrs, sends.size() * sizeof(float*)) != hipSuccess) return ncclUnhandledCudaError;
    (void)hipMemcpy(d_ptrs, sends.data(), sends.size() * sizeof(float*), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_loop_sum, dim3((unsigned)((root->count + 255) / 256)), dim3(256), 0, root->stream, root->recv, d_ptrs, w->world, root->count);
    hipError_t e = hipStreamSynchronize(root->stream);
    (void)hipFree(d_ptrs);
    w->pending.clear();
    return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
ncclResult_t loop_GetUniqueId(ncclUniqueId* id) { memset(id, 0, sizeof *id); return ncclSuccess; }
ncclResult_t loop_CommInitRank(ncclComm_t* out, int world, ncclUniqueId, int rank) {
    if (world != 1) return ncclInvalidUsage;                              // ranks in other processes need the real thing
    LoopWorld* w = new LoopWorld(); w->world = 1;
    *out = (ncclComm_t) new LoopComm{w, rank, 0};
    return ncclSuccess;
}
ncclResult_t loop_CommInitAll(ncclComm_t* out, int n, const int* devs) {
    LoopWorld* w = new LoopWorld(); w->world = n;
    for (int i = 0; i < n; i++) out[i] = (ncclComm_t) new LoopComm{w, i, devs[i]};
    return ncclSuccess;
}
ncclResult_t loop_CommDestroy(ncclComm_t c) { LoopComm* lc = (LoopComm*)c; if (lc->rank == 0) delete lc->w; delete lc; return ncclSuccess; }
ncclResult_t loop_GroupStart() { g_loop_grouped = true; g_loop_touched.clear(); return ncclSuccess; }
ncclResult_t loop_Reduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, int root, ncclComm_t c, hipStream_t stream) {
    if (t != ncclFloat32 || op != ncclSum) return ncclInvalidArgument;
    LoopComm* lc = (LoopComm*)c;
    lc->w->pending.push_back({(const float*)send, (float*)recv, count, root, lc->rank, stream});
    if (g_loop_grouped) { g_loop_touched.push_back(lc); return ncclSuccess; }
    return loop_flush(lc->w);
}
ncclResult_t loop_GroupEnd() {
    g_loop_grouped = false;
    ncclResult_t r = ncclSuccess;
    if (!g_loop_touched.empty()) r = loop_flush(g_loop_touched[0]->w);
    g_loop_touched.clear();
    return r;
}
const char* loop_GetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "loopback backend: invalid use (a reduce needs every rank of its communicator)"; }
bool g_loopback = false;

bool load_rccl(std::string& why) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.so) return true;
    if (!g_rccl.error.empty()) { why = g_rccl.error; return false; }
    if (const char* e = getenv("PT_COMM_LOOPBACK"); e && e[0] == '1') {       // test backend, see above
        g_rccl.GetUniqueId = loop_GetUniqueId; g_rccl.CommInitRank = loop_CommInitRank; g_rccl.CommInitAll = loop_CommInitAll; g_rccl.CommDestroy = loop_CommDestroy;
        g_rccl.Reduce = loop_Reduce; g_rccl.GroupStart = loop_GroupStart; g_rccl.GroupEnd = loop_GroupEnd; g_rccl.GetErrorString = loop_GetErrorString;
        g_rccl.so = (void*)&g_rccl; g_loopback = true;
        return true;
    }
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
    }
    std::string why;
    if (!load_rccl(why)) return pt_ctx_fail(ctxs[0], PT_ERR_STATE, why);
    for (int i = 0; i < n && !g_loopback; i++)
        for (int k = 0; k < i; k++)
            if (devs[(size_t)k] == devs[(size_t)i]) return pt_ctx_fail(ctxs[0], PT_ERR_ARG, "pt_comm_create: two contexts on one GPU (one communicator rank per GPU)");
    std::vector<ncclComm_t> comms((size_t)n);
    NCCLOK(ctxs[0], g_rccl.CommInitAll(comms.data(), n, devs.data()));
    pt_comm* c = new pt_comm();
    c->ctxs.assign(ctxs, ctxs + n); c->devs = devs; c->comms = comms; c->world = n; c->first_rank = 0;
    *out = c;
    return PT_OK;
}

// MOVES every rank's accumulation buffer onto rank `root`'s: the root's image becomes the sum of all ranks' images, every other rank's
// image is zeroed (its samples now live in the root's), so a later reduce -- a progressive refresh in the middle of a render -- adds only
// what the ranks have accumulated since and counts nothing twice.  Runs on the contexts' own streams (ordered behind the work already
// on them) and waits for the local part.  Every local context must hold a render of the same size.  The caller has to have finished
// (pt_render_run / pt_render_run_many return finished; after pt_render_step: pt_sync) every OTHER context that adds into one of these
// images -- the pools of pt_create_sharing_scene -- and, in the middle of a render, to have flushed their open pixel tickets
// (pt_image_flush + pt_sync): this call only knows the communicator's own contexts.
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
    // a failure between GroupStart and GroupEnd must still close the group, or this thread's next collective would queue into it
    int failed = PT_OK; size_t failed_at = 0; std::string failed_msg;
    for (size_t i = 0; i < n && failed == PT_OK; i++) {
        if (hipSetDevice(pt_ctx_device(comm->ctxs[i])) != hipSuccess) { failed = PT_ERR_HIP; failed_at = i; failed_msg = "pt_comm_reduce_image: hipSetDevice failed"; break; }
        ncclResult_t r = g_rccl.Reduce(img[i], img[i], count, ncclFloat32, ncclSum, root, comm->comms[i], pt_ctx_stream(comm->ctxs[i]));
        if (r != ncclSuccess) { failed = PT_ERR_HIP; failed_at = i; failed_msg = std::string("ncclReduce: ") + g_rccl.GetErrorString(r); }
    }
    ncclResult_t ge = g_rccl.GroupEnd();
    if (failed != PT_OK) return pt_ctx_fail(comm->ctxs[failed_at], failed, failed_msg);
    if (ge != ncclSuccess) return pt_ctx_fail(c0, PT_ERR_HIP, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(ge));
    for (size_t i = 0; i < n; i++) {
        pt_ctx* c = comm->ctxs[i];
        if (comm->first_rank + (int)i != root) {             // this rank's samples have moved to the root
            if (hipSetDevice(pt_ctx_device(c)) != hipSuccess || hipMemsetAsync(img[i], 0, count * 4, pt_ctx_stream(c)) != hipSuccess)
                return pt_ctx_fail(c, PT_ERR_HIP, "pt_comm_reduce_image: clearing a non-root image failed");
        }
        int r = pt_sync(c); if (r) return r;
    }
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
