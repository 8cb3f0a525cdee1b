// pt_kernels_render.hip -- the wavefront integrator on gfx950 (extend is in pt_kernels_extend.hip).
//
// One step() = compact+shade -> generate+extend(+logic, + the next step's count records): two launches on one stream, no host round trip in
// between (the reference awaits the queue after each of its four dispatches, path-tracer.js:86-107).  The launch
// that compacts the slots also samples the materials of the slots that continue (their bounce rays); the extend launch
// makes the camera ray of every camera-queue entry and traces it at once, and the thread that traced a slot's ray runs
// the slot's logic for the next step (pt_kernels_extend.hip).  The camera, material and logic kernels only touch their
// slot's own state, which is what makes the regrouping legal.
// Queue appends are order-preserving: wave64 ballots give each lane its rank, the count records in front of a
// workgroup give it its base (no scan launch), so every queue is in ascending path-slot order and pixel tickets are
// handed out in slot order (the reference's canonical serial order; its own order is a race,
// logic.js:158,168,205,213).  Because every camera-queue and every material-queue entry produces exactly one
// ray, the ray-trace queue (QueuesStage3.ray_trace_queue) is the concatenation camera ++ material and is
// never stored.
#include "pt_shading.h"

static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int rank_below(unsigned long long mask) {   // set bits below my lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}
__device__ __forceinline__ int4 add4(int4 a, int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// ============================================================================================
// first logic (logic.js:72-80): the step-0 branch -- seed the slot's random stream, ask for a camera ray and a ticket.
// Every later logic pass runs in the tail of the extend kernel (pt_kernels_extend.hip).
// ============================================================================================
__global__ __launch_bounds__(PT_BLOCK) void k_first_logic(PathState ps, RenderConst rc, uint8_t* __restrict__ code_out, unsigned long long* __restrict__ rec0) {
    int i = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= rc.num_paths) return;
    if (i % PT_STEP_BLOCK == 0) {                                          // step 0's count record: every slot wants a camera ray and a ticket
        unsigned long long n = (unsigned long long)min(PT_STEP_BLOCK, rc.num_paths - i);
        rec0[2 * (i / PT_STEP_BLOCK)] = n | (n << 32);
    }
    uint32_t gi = (uint32_t)(rc.slot_base + i) + rc.seed_mix;              // seed_mix: 0 unless the seed extension is set
    ps.random_seed[i] = (float)base_hash(gi, gi + 1u) * 2.3283064365386962890625e-10f + 0.008f;   // logic.js:77
    ps.path_throughput[i] = mk3(1.0f, 1.0f, 1.0f);
    code_out[i] = (uint8_t)(PT_CODE_CAMERA | PT_CODE_TICKET);
}

// ============================================================================================
// compact + shade (the kernel keeps the name it had when it also made the camera rays): stable scatter into the camera /
// material queues and pixel tickets (logic.js:155-219), then the bounce ray of every slot that continues
// (material.js:75-120) by the same workgroup.  Workgroup 0 also closes the step's books in the StepHeader.
// No scan and no counting launch: the extend kernel's logic tail leaves one count record per 1024 slots (64-bit atomics,
// two buffers alternating between steps; this launch clears the one the coming extend launch fills); a workgroup here adds up the records in
// front of its 1024-slot group (<= 4 16-B loads per thread, L2 resident) and, when it is one of four per group,
// re-counts the codes of the <= 3 workgroups in front of it inside the group.  Measured alternatives: a one-workgroup scan launch (9 us of
// pure latency per step), "last logic workgroup to arrive scans" (+100 us of contended arrivals), per-64-workgroup
// totals kept with integer atomics in logic (+25 us: 4 atomics per workgroup, profiles/r01_step_fusion_*).
// Also measured: shading here instead of in the extend tail, with logic as its own launch (20.8 + 28.9 us per step
// against 3 + ~12 us now, profiles/r01_cornell_1080p_64spp_kernel_stats_v3.txt); packing a workgroup's camera and
// material slots through LDS so that whole waves run only generate or only shade measured no gain when first tried
// (47.3 -> 47.5 us, the kernel then waited on the state traffic); with the 88-byte state it is the default (50 -> 44.5 us
// with the pool full, 48 -> 37 us while it drains).
// ============================================================================================
__device__ __forceinline__ int4 wave_sum4(int4 v) {
    for (int off = 32; off > 0; off >>= 1)
        v = add4(v, make_int4(__shfl_xor(v.x, off), __shfl_xor(v.y, off), __shfl_xor(v.z, off), __shfl_xor(v.w, off)));
    return v;
}
template <int PL, int BLK = PT_BLOCK>
__global__ __launch_bounds__(BLK) void k_compact_generate_shade(PathState ps, RenderConst rc, SceneView sc, const uint8_t* __restrict__ code_in,
                                                                     unsigned long long* __restrict__ rec, int nrecords, int parity, StepHeader* hdr,
                                                                     int32_t* __restrict__ cam_q, int32_t* __restrict__ mat_q) {
    // A workgroup takes PL planes of 256 consecutive slots (plane-major = slot order).  PL = 1: four workgroups per count
    // record, the shortest launch while the pool is full (6 waves per SIMD, one body per thread); PL = 4: one workgroup per
    // record, a quarter of the workgroups and of the prefix work -- half the cost of a step that has little else to do
    // (8 instead of 15 us for an empty pool, 24 instead of 34 us at 12 % live slots), slower with the pool full (48
    // against 44.5 us: 5 waves per SIMD, four bodies in a row per thread).  The host picks (pt_api_render.hip enqueue_step).
    constexpr int NW = BLK / 64, PER = PT_STEP_BLOCK / (PL * BLK);     // BLK: threads per workgroup (64 = one wave: A/B of a workgroup that fits wherever one extend wave retires)
    static_assert(PER * PL * BLK == PT_STEP_BLOCK, "a count record is a whole number of workgroups");
    __shared__ int4 s_cnt[PL][NW], s_front[NW], s_all[NW];
    __shared__ unsigned short s_todo[PL * BLK];
    const int b = blockIdx.x, base = b * PL * BLK, group = b / PER, sub = b % PER, wave = threadIdx.x >> 6, lane = lane_id();
    unsigned code[PL]; unsigned long long b_cam[PL], b_mat[PL], b_tix[PL];
#pragma unroll
    for (int p = 0; p < PL; p++) {
        int i = base + p * BLK + (int)threadIdx.x;
        code[p] = i < rc.num_paths ? code_in[i] : 0u;
    }
#pragma unroll
    for (int p = 0; p < PL; p++) {
        b_cam[p] = __ballot(code[p] & 1u); b_mat[p] = __ballot(code[p] & 2u); b_tix[p] = __ballot(code[p] & 4u);
        if (lane == 0) s_cnt[p][wave] = make_int4(__popcll(b_cam[p]), __popcll(b_mat[p]), __popcll(b_tix[p]), 0);
    }
    const int64_t ticket_base = hdr->ticket_next[parity];
    {   // counts in front of this workgroup; workgroup 0 adds up all records for the header
        const int upto = b == 0 ? nrecords : group;
        int4 acc = make_int4(0, 0, 0, 0);
        for (int j = threadIdx.x; j < upto; j += BLK) {
            unsigned long long v = rec[2 * j + parity];
            acc = add4(acc, make_int4((int)(v & 0xffffu), (int)((v >> 16) & 0xffffu), (int)((v >> 32) & 0xffffu), (int)(v >> 48)));
        }
        if (sub == 0 && threadIdx.x == 0) rec[2 * group + (parity ^ 1)] = 0ull;       // filled by the extend launch that follows
        for (int k = 0; k < sub * PL; k++) {                    // the planes of this record in front of this workgroup
            unsigned c = code_in[group * PT_STEP_BLOCK + k * BLK + threadIdx.x];
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
            hdr->prev_n_rays = hdr->n_rays;
            hdr->n_camera = t.x; hdr->n_material = t.y; hdr->n_rays = t.x + t.y;
            hdr->ticket_base = ticket_base;
            hdr->pixel_counter = ticket_base + t.z;
            hdr->ticket_next[parity ^ 1] = ticket_base + t.z;
            hdr->total_rays += (int64_t)t.x + t.y;
            hdr->total_samples += t.w;
            {   // tickets at or past the limit retire their slot (below); the slot's camera ray of this step is traced all the same
                const int64_t first_dead = ticket_base > rc.ticket_limit ? ticket_base : rc.ticket_limit;
                const int64_t dead = ticket_base + t.z - first_dead;
                if (dead > 0) hdr->total_retired += dead;
            }
            if (t.x + t.y == 0 && hdr->first_empty_step < 0) hdr->first_empty_step = hdr->step_index;
            hdr->step_index += 1;
        }
    }
    int4 off = make_int4(0, 0, 0, 0), tot = make_int4(0, 0, 0, 0), loc[PL];
    for (int w = 0; w < NW; w++) off = add4(off, s_front[w]);
#pragma unroll
    for (int p = 0; p < PL; p++) {
        loc[p] = tot;
        for (int w = 0; w < NW; w++) { int4 c = s_cnt[p][w]; if (w < wave) loc[p] = add4(loc[p], c); tot = add4(tot, c); }
    }
    // The material slots of the workgroup's planes are handed to its threads in turn through LDS: whole waves run the
    // material sample, and when the pool drains at the end of a render (a few live slots per record,
    // tools/step_profile.py) one wave does the work.  Camera slots only get their queue entry and, where due, their
    // pixel ticket here: the camera ray itself is made by the extend kernel, which traces it at once.
#pragma unroll
    for (int p = 0; p < PL; p++) {
        const int i = base + p * BLK + (int)threadIdx.x, j = p * BLK + (int)threadIdx.x;
        const int r_cam = rank_below(b_cam[p]), r_mat = rank_below(b_mat[p]);
        if (code[p] & 1u) cam_q[off.x + loc[p].x + r_cam] = i;
        if (code[p] & 2u) { mat_q[off.y + loc[p].y + r_mat] = i; s_todo[loc[p].y + r_mat] = (unsigned short)j; }
        if (code[p] & 4u) {
            int64_t ticket = ticket_base + off.z + loc[p].z + rank_below(b_tix[p]);
            if (ticket >= rc.ticket_limit) ps.state[i] |= PT_ST_RETIRED;                // logic.js:177-179
            else {                                                                      // logic.js:181-182
                ps.pixel_index[i] = (int32_t)(ticket % ((int64_t)rc.width * rc.height));
                ps.state[i] &= ~(7u << 2);                                              // samples_in_pixel = 0
            }
        }
    }
    __syncthreads();
    auto body = [&](int t) {
        const int slot = base + s_todo[t];
        int2 h = ps.hit[slot]; shade_one(ps, sc, slot, h.x, h.y);
    };
    if (PL == 1) { if ((int)threadIdx.x < tot.y) body((int)threadIdx.x); }                  // no loop: fewer registers, 6 waves per SIMD
    else for (int t = threadIdx.x; t < tot.y; t += BLK) body(t);
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
__global__ void k_camera_frame(RenderConst rc, CameraFrame* out) { *out = camera_frame(rc); }
void ptk_camera_frame(const RenderConst& rc, CameraFrame* out_dev, hipStream_t s) { hipLaunchKernelGGL(k_camera_frame, dim3(1), dim3(1), 0, s, rc, out_dev); }
int ptk_step_blocks(int num_paths) { return (num_paths + PT_STEP_BLOCK - 1) / PT_STEP_BLOCK; }
void ptk_first_logic(const PathState& ps, const RenderConst& rc, uint8_t* code, int4* blk_counts, hipStream_t s) {
    hipLaunchKernelGGL(k_first_logic, grid_for(rc.num_paths), dim3(PT_BLOCK), 0, s, ps, rc, code, (unsigned long long*)blk_counts);
}
void ptk_count_compact_generate_shade(const PathState& ps, const RenderConst& rc, const SceneView& sc, const uint8_t* code, int4* blk_counts, int parity, StepHeader* hdr,
                                int32_t* cam_q, int32_t* mat_q, int fat, hipStream_t s) {
    int nrecords = ptk_step_blocks(rc.num_paths);
#define PT_COMPACT(PLN, BLKN) hipLaunchKernelGGL((k_compact_generate_shade<PLN, BLKN>), dim3((PT_STEP_BLOCK / ((PLN) * (BLKN))) * nrecords), dim3(BLKN), 0, s, ps, rc, sc, code, (unsigned long long*)blk_counts, nrecords, parity, hdr, cam_q, mat_q)
    if (fat == 3) PT_COMPACT(4, 64);          // A/B shapes (PT_COMPACT_SHAPE): one-wave workgroups, four per record
    else if (fat == 4) PT_COMPACT(16, 64);    // one wave per record
    else if (fat == 5) PT_COMPACT(8, 128);    // two waves per record
    else if (fat == 6) PT_COMPACT(2, 128);    // two-wave workgroups, four per record
    else if (fat == 7) PT_COMPACT(8, 64);     // one-wave workgroups, two per record
    else if (fat == 8) PT_COMPACT(4, 128);    // two-wave workgroups, two per record
#undef PT_COMPACT
    else if (fat) hipLaunchKernelGGL(k_compact_generate_shade<PT_STEP_BLOCK / PT_BLOCK>, dim3(nrecords), dim3(PT_BLOCK), 0, s, ps, rc, sc, code, (unsigned long long*)blk_counts, nrecords, parity, hdr, cam_q, mat_q);
    else hipLaunchKernelGGL(k_compact_generate_shade<1>, grid_for(rc.num_paths), dim3(PT_BLOCK), 0, s, ps, rc, sc, code, (unsigned long long*)blk_counts, nrecords, parity, hdr, cam_q, mat_q);
}
void ptk_flush_tickets(const PathState& ps, int num_paths, float* image, hipStream_t s) {
    hipLaunchKernelGGL(k_flush_tickets, grid_for(num_paths), dim3(PT_BLOCK), 0, s, ps, num_paths, image);
}
void ptk_tonemap(const float4* image, int64_t n_pixels, uint32_t* rgba8, hipStream_t s) {
    hipLaunchKernelGGL(k_tonemap, grid_for(n_pixels), dim3(PT_BLOCK), 0, s, image, n_pixels, rgba8);
}
