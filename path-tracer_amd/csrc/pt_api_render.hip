// pt_api_render.hip -- host side of libpt_amd.so, second half: render (init / step / run), image, path-state read-backs, measurement
// hooks, display.  Declarations and the reference interface each entry point replaces: include/pt_amd.h.
#include "pt_ctx.h"

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

namespace {
void enqueue_step(pt_ctx* c, int count_visits, hipEvent_t ev0, hipEvent_t ev1) {
    SceneView sv = scene_view(c);
    int parity = (int)(c->step & 1);
    if (c->step == 0) ptk_first_logic(c->ps, c->rc, c->d_code, c->d_blk_counts, c->stream);
    // One compaction workgroup per count record while the step has little to do -- the first steps (camera rays only) and
    // the ramp-down, as far as the host knows it (pt_render_run sees headers up to two batches late) -- else four.  When
    // several pools run together (pt_render_run_many) always one: the other pools' launches hide this one's latencies,
    // so the shape that does less work wins (+2 %).
    int fat = c->overlapped || c->step < 6 || (c->known_rays >= 0 && 2 * c->known_rays < c->rc.num_paths);
    // ... and in 128-thread workgroups, unless the extend kernel is the LDS-staged one with its 256-thread workgroups: a compaction workgroup
    // then fits where two extend waves of a CU have retired instead of waiting for four (configs[2] +2.3 %, 10 M grid +1.4 %, instanced +0.6 %;
    // the Cornell box, staged, -2 %: profiles/r03_bench_ab_compaction_shape.txt)
    const bool staged = extend_variant_of(c) >= 3 && 12 * (int64_t)c->n_objects + 7 * c->n_tris <= PT_SCENE_LDS_QUADS;
    if (c->overlapped && !staged) fat = 5;
    static const int shape = getenv("PT_COMPACT_SHAPE") ? atoi(getenv("PT_COMPACT_SHAPE")) : 0;   // measurement: 1 = always four per record, 2 = always one (256 threads), 3 .. 7 = the shapes of ptk_count_compact_generate_shade
    if (shape) fat = shape >= 3 ? shape : shape == 2;
    ptk_count_compact_generate_shade(c->ps, c->rc, sv, c->d_code, c->d_blk_counts, parity, c->d_hdr, c->d_cam_q, c->d_mat_q, fat, c->stream);
    if (ev0) (void)hipEventRecord(ev0, c->stream);
    ptk_extend_queue(c->ps, c->rc, sv, c->d_hdr, c->d_cam_q, c->d_mat_q, c->d_spill, count_visits, extend_variant_of(c), nullptr, c->d_image, c->d_code,
                     (unsigned long long*)c->d_blk_counts + (parity ^ 1), (int)std::min<int64_t>(c->rays_cap, c->rc.num_paths), c->stream);
    if (ev1) (void)hipEventRecord(ev1, c->stream);
    c->step++;
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
        static const int xcd_chunk = getenv("PT_XCD_CHUNK") ? std::max(0, atoi(getenv("PT_XCD_CHUNK"))) : 0;     // measurement override
        rc.xcd_chunk = xcd_chunk; rc.pad0 = 0;
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

extern "C" {

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

int pt_extend_pass_counts(pt_ctx* c, int64_t* out12) {
    if (!c || !out12) return fail(c, PT_ERR_ARG, "pt_extend_pass_counts: null argument");
    if (!c->render_ready || !c->h_hdr) return fail(c, PT_ERR_STATE, "pt_extend_pass_counts: no render");
    const StepHeader& h = *c->h_hdr;
    for (int k = 0; k < 4; k++) { out12[k] = (int64_t)h.passes[k]; out12[4 + k] = 0; for (int sl = 0; sl < 64; sl++) out12[4 + k] += (int64_t)h.visits[sl][k]; }
    for (int k = 0; k < 4; k++) out12[8 + k] = (int64_t)h.lane_hist[k];
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
