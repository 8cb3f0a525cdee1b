// pt_kernels_render.hip -- the wavefront integrator on gfx950 (extend is in pt_kernels_extend.hip).
//
// One step() = logic -> compact+generate+shade -> extend: three launches on one stream, no host round trip in
// between (the reference awaits the queue after each of its four dispatches, path-tracer.js:86-107).  The thread
// that compacts a slot also produces the slot's next ray: the camera ray as soon as the pixel ticket is known, or
// the bounce ray (the camera and material kernels only touch their slot's own state).
// Queue appends are order-preserving: wave64 ballots give each lane its rank, the per-workgroup counts in front of a
// workgroup give it its base (no scan launch), so every queue is in ascending path-slot order and pixel tickets are handed out in slot order (the reference's canonical serial order; its own order is a race,
// logic.js:158,168,205,213).  Because every camera-queue and every material-queue entry produces exactly one
// ray, the ray-trace queue (QueuesStage3.ray_trace_queue) is the concatenation camera ++ material and is
// never stored.
#include "pt_internal.h"
#include "pt_math.cuh"

#define ALBEDO0_POW 0.217637640824031f   // pow(0.5, 2.2): material.js:107 through :72
#define ALBEDO1_POW 0.0289911870530680f  // pow(0.2, 2.2): material.js:110 through :72

static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int rank_below(unsigned long long mask) {   // set bits below my lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}
__device__ __forceinline__ int4 add4(int4 a, int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// ============================================================================================
// generate (camera.js:60-118)
// ============================================================================================
__device__ __forceinline__ void generate_one(const PathState& ps, const RenderConst& rc, int path) {
    int px = ps.pixel_index[path];
    float seed = ps.random_seed[path];
    float2 r2 = rand2(seed);
    ps.random_seed[path] = seed + 2.0f;
    float cx = (float)(px % rc.width) + r2.x, cy = (float)(px / rc.width) + r2.y;
    float sx = cx / (float)rc.width, sy = cy / (float)rc.height;
    sx = sx * 2.0f - 1.0f; sy = sy * 2.0f - 1.0f; sy *= -1.0f;
    float aspect = (float)rc.width / (float)rc.height;
    float sn, cs; pt_sincos(rc.fov_deg * 3.14159265358979323846f / 180.0f, sn, cs);
    f3 local = F3(aspect * sx * sn, 1.0f, sy * sn);
    f3 pos = F3(rc.cam_pos[0], rc.cam_pos[1], rc.cam_pos[2]);
    f3 fwd = normalize3(F3(rc.cam_look_at[0], rc.cam_look_at[1], rc.cam_look_at[2]) - pos);
    f3 right = normalize3(F3(fwd.y, -fwd.x, 0.0f));
    f3 up = cross3(right, fwd);
    f3 w = normalize3(local);
    f3 d = (right * w.x + fwd * w.y) + up * w.z;
    ps.path_o[path] = make_float4(pos.x, pos.y, pos.z, 0.0f);
    ps.path_d[path] = make_float4(d.x, d.y, d.z, 0.0f);
}

// ============================================================================================
// shade (material.js:75-120; scene.js:204-258 get_triangle_hit_info)
// ============================================================================================
__device__ __forceinline__ void shade_one(const PathState& ps, const SceneView& sc, int path) {
    f3 o = F3(ps.path_o[path]), d = F3(ps.path_d[path]);
    int hit_obj = ps.hit_obj[path], hit_tri = ps.hit_tri[path];
    const float4* ob = sc.objects + 8 * (int64_t)hit_obj;
    float4 l0 = ob[0], l1 = ob[1], l2 = ob[2], inf = ob[3], w0 = ob[4], w1 = ob[5], w2 = ob[6];
    f3 lo = xform(w0, w1, w2, o, 1.0f), ld = xform(w0, w1, w2, d, 0.0f);
    const float4* tr = sc.tris + 3 * (int64_t)hit_tri;
    f3 v0 = F3(tr[0]), v0v1 = F3(tr[1]) - v0, v0v2 = F3(tr[2]) - v0;
    f3 pvec = cross3(ld, v0v2);
    float det = dot3(v0v1, pvec), i_det = 1.0f / det;
    f3 tvec = lo - v0;
    f3 qvec = cross3(tvec, v0v1);
    f3 ln = normalize3(cross3(v0v1, v0v2));
    float ldist = dot3(v0v2, qvec) * i_det;
    f3 lhit = lo + ld * ldist;
    f3 whit = xform(l0, l1, l2, lhit, 1.0f);
    f3 n = normalize3(xform(l0, l1, l2, ln, 0.0f));
    float dist = length3(whit - o);
    f3 hit_pos = o + d * dist;
    f3 ov = __builtin_fabsf(n.x) > __builtin_fabsf(n.y) ? F3(-n.y, n.x, 0.0f) : F3(0.0f, -n.z, n.y);   // material.js:166-171
    f3 o1 = normalize3(ov), o2 = normalize3(cross3(o1, n));
    float seed = ps.random_seed[path];
    float4 bp = make_float4(0.0f, 0.0f, 0.0f, 0.0f); f3 wi = F3(0.0f, 0.0f, 0.0f);
    int mat = __float_as_int(inf.y);
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
    f3 no = hit_pos + n * 0.0001f;
    ps.material_throughput_pdf[path] = bp;
    ps.flags[path] |= 1u;
    ps.random_seed[path] = seed;
    ps.path_o[path] = make_float4(no.x, no.y, no.z, 0.0f);
    ps.path_d[path] = make_float4(nd.x, nd.y, nd.z, 0.0f);
}

// ============================================================================================
// logic (logic.js:62-152): per-slot state machine, image accumulation, decisions
// ============================================================================================
// code bits: 1 = camera queue, 2 = material queue, 4 = wants a pixel ticket.
// blk_counts[b] = {#camera, #material, #tickets, #finished samples} of workgroup b.
// Image accumulation: a slot owns its pixel for the 8 samples of a ticket (logic.js:6,129), so the samples are
// summed in the slot's ticket accumulator and reach the image with ONE group of float atomics per ticket instead
// of one per sample (scattered float atomics are the slowest memory operation on this chip).  The reference's
// own add order is a race (logic.js:222-233); sample counts stay exact.
__global__ __launch_bounds__(PT_STEP_BLOCK) void k_logic(PathState ps, RenderConst rc, int first_sample, float* __restrict__ image,
                                                         uint8_t* __restrict__ code_out, int4* __restrict__ blk_counts) {
    __shared__ int4 s_wave[PT_STEP_BLOCK / 64];
    int i = blockIdx.x * PT_STEP_BLOCK + threadIdx.x;
    unsigned code = 0; bool finished = false;
    if (i < rc.num_paths && (ps.flags[i] & (1u << 31)) == 0u) {
        if (first_sample) {
            uint32_t gi = (uint32_t)(rc.slot_base + i);
            ps.random_seed[i] = (float)base_hash(gi, gi + 1u) * 2.3283064365386962890625e-10f + 0.008f;   // logic.js:77
            ps.path_throughput[i] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
            code = 1u | 4u;
        } else {
            f3  tp = F3(ps.path_throughput[i]);
            int nb = ps.num_bounces[i] + 1;
            bool b_hit = ps.hit_obj[i] >= 0;
            uint32_t flags = ps.flags[i];
            f3 contrib = F3(0.0f, 0.0f, 0.0f);
            if (flags & 1u) {                                                   // logic.js:94-109
                tp = tp * F3(ps.material_throughput_pdf[i]);
                if (nb > 3) {
                    float seed = ps.random_seed[i];
                    float2 r2 = rand2(seed);
                    ps.random_seed[i] = seed + 2.0f;
                    float q = fmin_(fmax_(0.1f, 1.0f - tp.y), 0.7f);
                    if (r2.x < q) tp = F3(0.0f, 0.0f, 0.0f); else tp = tp / (1.0f - q);
                }
            }
            if (nb > 20) tp = F3(0.0f, 0.0f, 0.0f);                              // logic.js:111-113
            if (!b_hit) { contrib = F3(0.0f + 2.0f * tp.x, 0.0f + 2.0f * tp.y, 0.0f + 2.0f * tp.z); tp = F3(0.0f, 0.0f, 0.0f); }
            if (tp.x == 0.0f && tp.y == 0.0f && tp.z == 0.0f) {                  // logic.js:121-133
                finished = true; code |= 1u; nb = 0; tp = F3(1.0f, 1.0f, 1.0f);
                int sip = ps.samples_in_pixel[i];
                if (sip >= PT_SAMPLES_PER_TICKET - 1) code |= 4u; else ps.samples_in_pixel[i] = sip + 1;
            } else { code |= 2u; nb += 1; }
            ps.path_throughput[i] = make_float4(tp.x, tp.y, tp.z, 0.0f);
            ps.num_bounces[i] = nb;
            ps.flags[i] = 0u;
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
        }
    }
    if (i < rc.num_paths) code_out[i] = (uint8_t)code;
    unsigned long long b_cam = __ballot(code & 1u), b_mat = __ballot(code & 2u), b_tix = __ballot(code & 4u), b_fin = __ballot(finished);
    if (lane_id() == 0)
        s_wave[threadIdx.x >> 6] = make_int4(__popcll(b_cam), __popcll(b_mat), __popcll(b_tix), __popcll(b_fin));
    __syncthreads();
    if (threadIdx.x == 0) {
        int4 t = s_wave[0];
        for (int w = 1; w < PT_STEP_BLOCK / 64; w++) t = add4(t, s_wave[w]);
        blk_counts[blockIdx.x] = t;
    }
}

// ============================================================================================
// compact + generate + shade: stable scatter into the camera / material queues and pixel tickets (logic.js:155-219),
// then the slot's next ray by the same thread -- the camera ray of a slot that starts a sample (camera.js:60-118) or
// the bounce ray of a slot that continues (material.js:75-120); both only touch that slot's own state, so neither
// waits for a queue.  Workgroup 0 also closes the step's books in the StepHeader.
// No scan launch: logic leaves one count record per 1024 slots (~1000 per step); a workgroup here (256 slots) adds up
// the records in front of its 1024-slot group (<= 4 16-B loads per thread, L2 resident) and re-counts the codes of the
// <= 3 workgroups in front of it inside the group.  Measured alternatives: a one-workgroup scan launch (9 us of
// pure latency per step), "last logic workgroup to arrive scans" (+100 us of contended arrivals), per-64-workgroup
// totals kept with integer atomics in logic (+25 us: 4 atomics per workgroup, profiles/r01_step_fusion_*).
// ============================================================================================
__device__ __forceinline__ int4 wave_sum4(int4 v) {
    for (int off = 32; off > 0; off >>= 1)
        v = add4(v, make_int4(__shfl_xor(v.x, off), __shfl_xor(v.y, off), __shfl_xor(v.z, off), __shfl_xor(v.w, off)));
    return v;
}
__global__ __launch_bounds__(PT_BLOCK) void k_compact_generate_shade(PathState ps, RenderConst rc, SceneView sc, const uint8_t* __restrict__ code_in,
                                                                     const int4* __restrict__ blk_counts, int nrecords, int parity, StepHeader* hdr,
                                                                     int32_t* __restrict__ cam_q, int32_t* __restrict__ mat_q) {
    constexpr int NW = PT_BLOCK / 64, PER = PT_STEP_BLOCK / PT_BLOCK;
    __shared__ int4 s_wave[NW], s_front[NW], s_all[NW];
    const int b = blockIdx.x, group = b / PER, sub = b % PER, wave = threadIdx.x >> 6, lane = lane_id();
    int i = b * PT_BLOCK + threadIdx.x;
    unsigned code = i < rc.num_paths ? code_in[i] : 0u;
    unsigned long long b_cam = __ballot(code & 1u), b_mat = __ballot(code & 2u), b_tix = __ballot(code & 4u);
    if (lane == 0) s_wave[wave] = make_int4(__popcll(b_cam), __popcll(b_mat), __popcll(b_tix), 0);
    const int64_t ticket_base = hdr->ticket_next[parity];
    {   // counts in front of this workgroup; workgroup 0 adds up all records for the header
        const int upto = b == 0 ? nrecords : group;
        int4 acc = make_int4(0, 0, 0, 0);
        for (int j = threadIdx.x; j < upto; j += PT_BLOCK) acc = add4(acc, blk_counts[j]);
        for (int k = 0; k < sub; k++) {
            unsigned c = code_in[(group * PER + k) * PT_BLOCK + threadIdx.x];
            acc.x += c & 1u; acc.y += (c >> 1) & 1u; acc.z += (c >> 2) & 1u;
        }
        acc = wave_sum4(acc);
        if (lane == 0) { if (b == 0) s_all[wave] = acc; s_front[wave] = b == 0 ? make_int4(0, 0, 0, 0) : acc; }
    }
    __syncthreads();
    if (b == 0 && wave == NW - 1) {                             // the step's totals -> StepHeader
        int4 t = make_int4(0, 0, 0, 0);
        for (int w = 0; w < NW; w++) t = add4(t, s_all[w]);
        unsigned* cell = &hdr->cls_max[lane >> 1][lane & 1];    // dispatch order of this step's extend launch (StepHeader::cls_max)
        unsigned v = *cell;
        *cell = 0;
        for (int off = 2; off < 64; off <<= 1) v = max(v, (unsigned)__shfl_xor((int)v, off));
        unsigned cam = __shfl((int)v, 0), mat = __shfl((int)v, 1);
        if (lane == 0) {
            hdr->extend_order = mat > cam ? 1 : 0;
            hdr->n_camera = t.x; hdr->n_material = t.y; hdr->n_rays = t.x + t.y;
            hdr->ticket_base = ticket_base;
            hdr->pixel_counter = ticket_base + t.z;
            hdr->ticket_next[parity ^ 1] = ticket_base + t.z;
            hdr->total_rays += (int64_t)t.x + t.y;
            hdr->total_samples += t.w;
            if (t.x + t.y == 0 && hdr->first_empty_step < 0) hdr->first_empty_step = hdr->step_index;
            hdr->step_index += 1;
            hdr->work_counter = 0;
        }
    }
    int4 off = make_int4(0, 0, 0, 0);
    for (int w = 0; w < NW; w++) { off = add4(off, s_front[w]); if (w < wave) off = add4(off, s_wave[w]); }
    if (code & 1u) cam_q[off.x + rank_below(b_cam)] = i;
    if (code & 2u) mat_q[off.y + rank_below(b_mat)] = i;
    if (code & 4u) {
        int64_t ticket = ticket_base + off.z + rank_below(b_tix);
        if (ticket >= rc.ticket_limit) ps.flags[i] |= (1u << 31);                  // logic.js:177-179
        else {                                                                      // logic.js:181-182
            ps.pixel_index[i] = (int32_t)(ticket % ((int64_t)rc.width * rc.height));
            ps.samples_in_pixel[i] = 0;
        }
    }
    if (code & 1u) generate_one(ps, rc, i);
    if (code & 2u) shade_one(ps, sc, i);
}

// Adds the unfinished tickets' partial sums to the image (and clears them) so that a host looking at the image in
// the middle of a render sees every finished sample, as the reference's per-sample adds would show.
__global__ __launch_bounds__(PT_BLOCK) void k_flush_tickets(PathState ps, int num_paths, float* __restrict__ image) {
    int i = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= num_paths) return;
    float4 acc = ps.ticket_acc[i];
    if (acc.w == 0.0f) return;
    float* px = image + 4 * (int64_t)ps.pixel_index[i];
    unsafeAtomicAdd(px + 0, acc.x); unsafeAtomicAdd(px + 1, acc.y); unsafeAtomicAdd(px + 2, acc.z); unsafeAtomicAdd(px + 3, acc.w);
    ps.ticket_acc[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ============================================================================================
// tonemap (display.js:98-113) + unorm8 store of the canvas
// ============================================================================================
__global__ __launch_bounds__(PT_BLOCK) void k_tonemap(const float4* __restrict__ image, int64_t n, uint32_t* __restrict__ rgba8) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    float4 p = image[i];
    float c[3] = { p.x / p.w, p.y / p.w, p.z / p.w };
    float lum = (c[0] * 0.2126f + c[1] * 0.7152f) + c[2] * 0.0722f;
    uint32_t out = 0xff000000u;
    for (int k = 0; k < 3; k++) {
        float v = powf(c[k] / (1.0f + lum), 1.0f / 2.2f);
        if (!(v > 0.0f)) v = 0.0f;
        if (v > 1.0f) v = 1.0f;
        out |= (uint32_t)(v * 255.0f + 0.5f) << (8 * k);
    }
    rgba8[i] = out;
}

// ---- launchers -------------------------------------------------------------------------------
int ptk_step_blocks(int num_paths) { return (num_paths + PT_STEP_BLOCK - 1) / PT_STEP_BLOCK; }
void ptk_logic(const PathState& ps, const RenderConst& rc, int first_sample, float* image, uint8_t* code, int4* blk_counts, hipStream_t s) {
    hipLaunchKernelGGL(k_logic, dim3(ptk_step_blocks(rc.num_paths)), dim3(PT_STEP_BLOCK), 0, s, ps, rc, first_sample, image, code, blk_counts);
}
void ptk_compact_generate_shade(const PathState& ps, const RenderConst& rc, const SceneView& sc, const uint8_t* code, const int4* blk_counts,
                                int parity, StepHeader* hdr, int32_t* cam_q, int32_t* mat_q, hipStream_t s) {
    hipLaunchKernelGGL(k_compact_generate_shade, grid_for(rc.num_paths), dim3(PT_BLOCK), 0, s, ps, rc, sc, code, blk_counts,
                       ptk_step_blocks(rc.num_paths), parity, hdr, cam_q, mat_q);
}
void ptk_flush_tickets(const PathState& ps, int num_paths, float* image, hipStream_t s) {
    hipLaunchKernelGGL(k_flush_tickets, grid_for(num_paths), dim3(PT_BLOCK), 0, s, ps, num_paths, image);
}
void ptk_tonemap(const float4* image, int64_t n_pixels, uint32_t* rgba8, hipStream_t s) {
    hipLaunchKernelGGL(k_tonemap, grid_for(n_pixels), dim3(PT_BLOCK), 0, s, image, n_pixels, rgba8);
}
