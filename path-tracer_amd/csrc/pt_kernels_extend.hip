// pt_kernels_extend.hip -- the extend stage on gfx950: closest hit over the two-level LBVH
// (ray-trace.js:53-69 driving intersect_bvh, scene.js:313-515).
//
// One ray per lane.  Every variant below performs the reference's traversal in the reference's order
// (near child first, ties to the right child, single stack shared by TLAS and BLAS, hit_dist carried
// through the object-space round trips), so hit indices, hit distances and per-ray visit counts do not
// depend on the variant; variants differ only in how rays are mapped to lanes and where the stack lives.
//
// Stack: the first LDS_N entries of a lane's stack live in LDS as [entry][thread] (bank = thread & 31
// whatever the entry, lanes l / l+32 are served in different LDS cycles: divergent depths never
// conflict); deeper entries go to an HBM spill laid out [entry][slot] so that equal depths coalesce.
#include "pt_shading.h"
#include <algorithm>

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// record `idx` of an array of QUADS-quad records through a 32-bit byte offset from the (wave-uniform) base: one shift instead of 64-bit
// address arithmetic (global_load ... v_offset, s[base]).  Only the kind-scheduled bodies use it; the host sends scenes with 2^26 or
// more records in an array to the per-lane loop, which addresses with 64 bits (pt_api.hip extend_variant_of).
template <int QUADS> __device__ __forceinline__ const float4* rec32(const float4* base, int idx) {
    return (const float4*)((const char*)base + (uint32_t)idx * (uint32_t)(16 * QUADS));
}
__device__ __forceinline__ int rank_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// SHALLOW: the host has checked that no ray's stack can exceed LDS_N entries in this scene (TLAS depth + deepest BLAS
// depth, pt_api.hip scene_stack_bound), so neither the spill nor the overflow guard is compiled in.
// GUARDED = false: the host knows a bound of at most PT_STACK_SIZE entries for this scene and that its trees are proper (no one-primitive
// tree, whose right child is the node itself), so a walk ends by itself and never fills the stack: neither the iteration guard nor the
// full-stack guard is compiled in (the spill stays).  SHALLOW implies it.
template <int LDS_N, int BLOCK, bool SHALLOW = false, bool GUARDED = true>
struct Stack {
    static constexpr bool shallow = SHALLOW;
    static constexpr bool guarded = GUARDED && !SHALLOW;
    int* lds;            // &s_stack[0][thread]
    int32_t* spill_base; // &spill[0][0] (wave-uniform: the lane's address is only formed on the rare path that needs it)
    int slot;            // this lane's slot in a spill entry
    int64_t stride;      // slots per spill entry
    __device__ __forceinline__ int32_t* spill_at(int sp) const { return spill_base + ((int64_t)(sp - LDS_N) * stride + slot); }
    // Whether any lane of the wave is past the LDS part is ONE wave-uniform branch; the per-lane choice between LDS and
    // spill (exec-mask bookkeeping on the scalar unit, which a CU's 32 waves share) is only made behind it.
    __device__ __forceinline__ void put(int sp, int v) const {
        if (SHALLOW || !__any(sp >= LDS_N)) lds[sp * BLOCK] = v;
        else if (sp < LDS_N) lds[sp * BLOCK] = v; else *spill_at(sp) = v;
    }
    __device__ __forceinline__ int get(int sp) const {               // sp = -1 ends the loop; the value read then is not used
        if (SHALLOW || !__any(sp >= LDS_N)) return lds[(sp < 0 ? 0 : sp) * BLOCK];
        return sp < 0 ? 0 : (sp < LDS_N ? lds[sp * BLOCK] : *spill_at(sp));
    }
    static constexpr int lds_n = LDS_N;
    static constexpr int block = BLOCK;
    __device__ __forceinline__ int& park(int j) const { return lds[(LDS_N + j) * BLOCK]; }     // five words per lane behind the stack (tolerance-mode kernels only)
};

#define PT_ITER_OVERFLOW (1 << 30)
struct Ray {
    f3 o, d, id;                 // origin, direction, 1/direction in the current space
    float hit_dist; int hit_obj, hit_tri;
    int sp, node, obj, switch_pt, iter;
    int n_inner, n_leaf, n_enter, n_leave;
    // "stopped by a guard" is bit 30 of iter (one register less in every walk; iter itself stays below 2^21)
    __device__ __forceinline__ void set_overflow(bool c) { iter |= c ? PT_ITER_OVERFLOW : 0; }
    __device__ __forceinline__ int overflow() const { return (iter >> 30) & 1; }
    __device__ __forceinline__ int trips() const { return iter & (PT_ITER_OVERFLOW - 1); }
};

__device__ __forceinline__ void ray_init(Ray& r, f3 o, f3 d, int root /* SceneView::tlas_root */) {
    r.o = o; r.d = d; r.id = rcp3(d);
    r.hit_dist = 1e6f; r.hit_obj = -1; r.hit_tri = -1;                       // scene.js:317-319
    r.sp = 0; r.node = root; r.obj = -1; r.switch_pt = -1; r.iter = 0;        // scene.js:321-324
    r.n_inner = r.n_leaf = r.n_enter = r.n_leave = 0;
}

// scene.js:508-515 with 1/d hoisted (it only changes when the ray changes space)
__device__ __forceinline__ float slab(f3 low, f3 high, f3 o, f3 id) {
    f3 f = (high - o) * id, n = (low - o) * id;
    f3 tmax = max3(f, n), tmin = min3(f, n);
    float t0 = fmax_(tmin.x, fmax_(tmin.y, tmin.z));
    float t1 = fmin_(tmax.x, fmin_(tmax.y, tmax.z));
    return (t1 >= t0) ? ((t1 < 0.0f) ? -1e30f : t0) : -1e30f;
}
// scene.js:478-506
__device__ __forceinline__ float tri_hit(f3 o, f3 d, float4 a, float4 b, float4 c) {
    f3 v0 = F3(a), v0v1 = F3(b), v0v2 = F3(c);                  // SceneView::tris holds the edges
    f3 pvec = cross3(d, v0v2);
    float det = dot3(v0v1, pvec);
    if (__builtin_fabsf(det) < 1e-10f) return -1.0f;
    float i_det = 1.0f / det;
    f3 tvec = o - v0;
    float u = dot3(tvec, pvec) * i_det;
    if (u < 0.0f || u > 1.0f) return -1.0f;
    f3 qvec = cross3(tvec, v0v1);
    float v = dot3(d, qvec) * i_det;
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    return dot3(v0v2, qvec) * i_det;
}
// scene.js:333-350 / :364-379: map o and the point o + d*hit_dist, renormalise d, recompute hit_dist
__device__ __forceinline__ void switch_space(float4 r0, float4 r1, float4 r2, Ray& r) {
    f3 hit_pos = r.o + r.d * r.hit_dist;
    f3 new_o = xform(r0, r1, r2, r.o, 1.0f), new_h = xform(r0, r1, r2, hit_pos, 1.0f);
    f3 nd = new_h - new_o;
    r.hit_dist = length3(nd);
    r.d = nd / r.hit_dist;
    r.o = new_o;
    r.id = rcp3(r.d);
}

// One trip through the reference's while loop (scene.js:328-440).  Returns false when the loop ends.
// tolerance mode: order two (distance, child) pairs, nearer first
__device__ __forceinline__ void near_first(float& ka, int& ra, float& kb, int& rb) {
    bool sw = kb < ka;
    float k = sw ? kb : ka; int rr = sw ? rb : ra;
    kb = sw ? ka : kb; rb = sw ? ra : rb; ka = k; ra = rr;
}

// WIDE (tolerance mode): inside an object the nodes are the compressed 4-wide collapse (SceneView::wide, layout in
// pt_internal.h): one 64-B fetch tests up to four (outward-rounded) boxes, the nearest valid child is taken and the others
// are pushed farthest first.  Same closest hit up to ties;
// visit order, visit counts and the low bits of hit_dist differ from the reference's (hit_dist is carried through the
// object-space round trips, which happen at different moments).  The TLAS stays binary.
template <bool COUNT, class STK, bool WIDE = false>
__device__ __forceinline__ bool trav_step(const SceneView& sc, Ray& r, const STK& stk) {
    r.iter++;
    if (STK::guarded && r.iter > PT_MAX_ITERS) { r.set_overflow(true); return false; }   // a scene with a stack bound has proper trees: the walk ends
    if (r.sp < r.switch_pt) {                                   // :330-355 leaving the object
        const float4* ob = sc.objects + 8 * (int64_t)r.obj;
        switch_space(ob[0], ob[1], ob[2], r);
        r.switch_pt = -1; r.obj = -1;
        if (COUNT) r.n_leave++;
    }
    if (r.node < 0) {
        if (r.obj < 0) {                                        // :357-379 entering an object
            r.obj = -(r.node + 1);
            const float4* ob = (WIDE ? sc.objects_wide : sc.objects) + 8 * (int64_t)r.obj;
            r.node = __float_as_int(ob[3].x); r.switch_pt = r.sp;
            switch_space(ob[4], ob[5], ob[6], r);
            if (COUNT) r.n_enter++;
        } else {                                                // :380-391 triangle
            int t = -(r.node + 1);
            const float4* tr = sc.tris + 3 * (int64_t)t;
            float dist = tri_hit(r.o, r.d, tr[0], tr[1], tr[2]);
            if (dist > 0.0f && dist < r.hit_dist) { r.hit_dist = dist; r.hit_tri = t; r.hit_obj = r.obj; }
            if (COUNT) r.n_leaf++;
            r.sp -= 1; r.node = stk.get(r.sp);
        }
    } else if (WIDE && r.obj >= 0) {                            // tolerance mode: 4-wide BLAS node
        const float4* nd = sc.wide + PT_WIDE_QUADS * (int64_t)r.node;
        float4 h = nd[0], rf = nd[1], qa = nd[2], qb = nd[3];
        if (COUNT) r.n_inner++;                                 // one 64-B record
        f3 org = F3(h), st = F3(h.w, qb.z, qb.w);
        unsigned lx = __float_as_uint(qa.x), ly = __float_as_uint(qa.y), lz = __float_as_uint(qa.z), hx = __float_as_uint(qa.w),
                 hy = __float_as_uint(qb.x), hz = __float_as_uint(qb.y);
        int r0 = __float_as_int(rf.x), r1 = __float_as_int(rf.y), r2 = __float_as_int(rf.z), r3 = __float_as_int(rf.w);
        const float INF = __builtin_inff();
        auto child_t = [&](int c, int ref) -> float {
            int sh = 8 * c;
            f3 lo = F3(__builtin_fmaf((float)((lx >> sh) & 0xffu), st.x, org.x), __builtin_fmaf((float)((ly >> sh) & 0xffu), st.y, org.y),
                       __builtin_fmaf((float)((lz >> sh) & 0xffu), st.z, org.z));
            f3 hi = F3(__builtin_fmaf((float)((hx >> sh) & 0xffu), st.x, org.x), __builtin_fmaf((float)((hy >> sh) & 0xffu), st.y, org.y),
                       __builtin_fmaf((float)((hz >> sh) & 0xffu), st.z, org.z));
            float t = slab(lo, hi, r.o, r.id);
            return (ref != PT_WIDE_EMPTY && t != -1e30f && t < r.hit_dist) ? t : INF;
        };
        float k0 = child_t(0, r0), k1 = child_t(1, r1), k2 = child_t(2, r2), k3 = child_t(3, r3);
        near_first(k0, r0, k1, r1); near_first(k2, r2, k3, r3); near_first(k0, r0, k2, r2); near_first(k1, r1, k3, r3); near_first(k1, r1, k2, r2);
        int nv = (k0 != INF) + (k1 != INF) + (k2 != INF) + (k3 != INF);
        if (nv == 0) { r.sp -= 1; r.node = stk.get(r.sp); }
        else {
            if (r.sp + nv - 1 > PT_STACK_SIZE) { r.set_overflow(true); return false; }
            if (nv > 3) { stk.put(r.sp, r3); r.sp += 1; }
            if (nv > 2) { stk.put(r.sp, r2); r.sp += 1; }
            if (nv > 1) { stk.put(r.sp, r1); r.sp += 1; }
            r.node = r0;
        }
    } else {                                                    // :392-439 inner node
        const float4* nd = sc.nodes + 4 * (int64_t)r.node;     // TLAS and BLAS ids share one array (SceneView::nodes)
        float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
        if (COUNT) r.n_inner++;
        float l_dist = slab(F3(q0), F3(q1), r.o, r.id), r_dist = slab(F3(q2), F3(q3), r.o, r.id);
        bool l_valid = l_dist != -1e30f && l_dist < r.hit_dist;
        bool r_valid = r_dist != -1e30f && r_dist < r.hit_dist;
        int l_child = __float_as_int(q0.w), r_child = __float_as_int(q2.w);
        // one select for the child to visit, one guarded push, one guarded pop (fewer exec-mask manipulations than the
        // reference's four-way if / else chain; same outcome in every case)
        const bool both = l_valid && r_valid;
        const bool take_l = l_valid && (!r_valid || l_dist < r_dist);
        if (both) {
            if (STK::guarded && __any(r.sp >= STK::lds_n) && r.sp >= PT_STACK_SIZE) { r.set_overflow(true); return false; }
            stk.put(r.sp, take_l ? r_child : l_child);
            r.sp += 1;
        }
        r.node = take_l ? l_child : r_child;
        if (!(l_valid || r_valid)) { r.sp -= 1; r.node = stk.get(r.sp); }
    }
    return r.sp >= 0;
}
// --------------------------------------------------------------------------------------------
// Kind-scheduled traversal (extend variant 4).  The reference's loop does one of four things per trip and lane -- test an inner
// node's two boxes, test a triangle, enter an object, and, in front of any of them, leave one (scene.js:330-440).  With one ray per
// lane a wave pays for every kind some lane needs in every trip: all four bodies, one after the other, each under a partial exec
// mask (measured on the 82 k-triangle mesh: 43 % of the lane slots of the issued vector instructions do work).  Here every lane
// keeps the kind of its NEXT operation, and each pass through the loop the wave runs ONE body: the one that advances the most lanes
// per instruction it costs.  A lane's own sequence of operations -- visit order, arithmetic, stack, counts -- is exactly the
// reference's; only when it runs changes.
// Measured (profiles/r02_extend_ab_sched_*): each body grows by the bookkeeping (next kind, guard: inner node 95 -> 116 vector
// instructions, triangle 88 -> 112, one switch body of 182 for the two of 131 + 134), so on identical queues the launch issues MORE
// vector instructions (+14 % on the 82 k-triangle mesh, +56 % on the one-object 10 M-triangle grid, whose walk is inner nodes and
// triangles only) -- a model fed with the oracle's per-ray kind sequences had promised 1.4x fewer, by pricing the bodies too low.
// What it does buy is fewer waits (SQ_WAIT_ANY -14 %): in the full render it was +3...8 % on scenes with several objects and -12 %
// on the one-object grid with the first weights; with the inner body's weight per scene (SceneView::sched_wi: 3 for one object, 6
// otherwise) it wins there too: the kind-scheduled walk is the library's default for every scene (pt_api.hip extend_variant_of: variant 4,
// or variant 9 -- the same with the fused inner + triangle pass below -- for scenes of more than one object).
// --------------------------------------------------------------------------------------------
// (The tolerance-mode kernels need 65-72 registers, 7 waves per SIMD.  Capped at 64 for 8 waves they spill two registers inside the
// outer loop and lose 11 %: profiles/r03_wide_occupancy_ab.txt.)
enum { PT_K_INNER = 0, PT_K_LEAF = 1, PT_K_SWITCH = 2, PT_K_DONE = 3, PT_K_WIDE = 4 };
// the weights of the three bodies in the choice of the next pass are per scene: SceneView::sched_wi / wl / ws
// WIDE (tolerance mode): a node id >= 0 inside an object names a 4-wide node (SceneView::wide); in the TLAS it is a binary node as ever
template <bool WIDE = false>
__device__ __forceinline__ int kind_after(const Ray& r) {
    // three selects, no branches (the exec-mask bookkeeping of nested ifs costs the shared scalar unit more than the selects cost a SIMD)
    int k = r.node >= 0 ? ((WIDE && r.obj >= 0) ? PT_K_WIDE : PT_K_INNER) : (r.obj < 0 ? PT_K_SWITCH : PT_K_LEAF);   // :357 enter / :380 triangle
    k = r.sp < r.switch_pt ? PT_K_SWITCH : k;                  // :330 the next trip begins by leaving the object
    return r.sp < 0 ? PT_K_DONE : k;                           // the loop condition (scene.js:328)
}
// a trip = one INNER, LEAF or entering SWITCH; a leaving SWITCH belongs to the trip that follows it (r.iter counts trips, as in trav_step)
template <class STK>
__device__ __forceinline__ bool trip_guard(Ray& r) {
    r.iter++;
    if (STK::guarded && r.iter > PT_MAX_ITERS) { r.set_overflow(true); return false; }
    return true;
}
// TOP: the TLAS and the tree-top blocks of the meshes, node ids >= sc.n_tris, are in LDS at `lds` (staged by the kernel).
// The other BLAS nodes come from the depth-first copy in memory as ever.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) v4f* LdsQuads;   // typed as LDS, and volatile, so that the two fetch paths below stay two
                                                                          // (the compiler otherwise merges them into flat loads through a select of addresses)
// The visit of the inner node whose record is in q0..q3 (:392-439): both slab tests, the nearer valid child next, the other one pushed,
// a pop when neither is valid.  Returns whether a push found the stack full (guarded stacks only).
template <class STK>
__device__ __forceinline__ bool inner_visit(Ray& r, const STK& stk, float4 q0, float4 q1, float4 q2, float4 q3) {
    float l_dist = slab(F3(q0), F3(q1), r.o, r.id), r_dist = slab(F3(q2), F3(q3), r.o, r.id);
    bool l_valid = l_dist != -1e30f && l_dist < r.hit_dist;
    bool r_valid = r_dist != -1e30f && r_dist < r.hit_dist;
    int l_child = __float_as_int(q0.w), r_child = __float_as_int(q2.w);
    const bool both = l_valid && r_valid;
    const bool take_l = l_valid && (!r_valid || l_dist < r_dist);
    const bool full = STK::guarded && both && r.sp >= PT_STACK_SIZE;       // a push onto a full stack ends the ray
    if (both && !full) {
        stk.put(r.sp, take_l ? r_child : l_child);
        r.sp += 1;
    }
    r.node = take_l ? l_child : r_child;
    // (push and pop without branches -- every lane writes its first free slot and reads its top entry whenever all of them are in
    // LDS -- was measured: -1 %, profiles/r02_instruction_cuts.txt)
    if (!(l_valid || r_valid)) { r.sp -= 1; r.node = stk.get(r.sp); }
    // (Prefetching the pushed child's record here -- one dword by a load that writes LDS directly, no register -- was built as an A/B
    // variant at commit 701babf and measured 8-11 % slower: loads return in order, so the near child's record waits for the prefetched
    // line.  profiles/r03_extend_ab_prefetch_pushed_child.txt)
    return full;
}
// TOP == 2 (extend variant 7, an A/B of north_star's "structure-of-arrays node buffers"): the four quads of a node come from four arrays
// (SceneView::soa: quad k of node i at soa[k * soa_stride + i]) instead of one 64-B record.
// KIND = false (traverse_masked): the caller classifies the lanes itself; the body only reports a lane stopped by a guard (1) or not (0).
template <bool COUNT, class STK, int TOP = 0, bool WIDE = false, bool KIND = true>
__device__ __forceinline__ int op_inner(const SceneView& sc, Ray& r, const STK& stk, LdsQuads lds = nullptr) {      // :392-439
    // One exit: a lane stopped by a guard runs the visit like the others (it changes no hit; its stack slot and node are its own and
    // never read again) and is retired at the end.  Early returns here cost every lane of every pass a dozen register copies.
    r.iter++;
    const bool stop = STK::guarded && r.iter > PT_MAX_ITERS;
    float4 q0, q1, q2, q3;
    if (TOP == 1 && r.node >= sc.n_tris) {
        LdsQuads nd = lds + 4 * (r.node - sc.n_tris);
        const v4f a = nd[0], b = nd[1], c = nd[2], d = nd[3];
        q0 = make_float4(a.x, a.y, a.z, a.w); q1 = make_float4(b.x, b.y, b.z, b.w); q2 = make_float4(c.x, c.y, c.z, c.w); q3 = make_float4(d.x, d.y, d.z, d.w);
    } else if (TOP == 2) {
        const float4* nd = rec32<1>(sc.soa, r.node);
        q0 = nd[0]; q1 = nd[sc.soa_stride]; q2 = nd[2 * (int64_t)sc.soa_stride]; q3 = nd[3 * (int64_t)sc.soa_stride];
    } else {
        const float4* nd = rec32<4>(sc.nodes, r.node);
        q0 = nd[0]; q1 = nd[1]; q2 = nd[2]; q3 = nd[3];
    }
    if (COUNT) r.n_inner += stop ? 0 : 1;
#ifdef PT_EXTRA_VALU    // sensitivity experiment (tools/valu_sensitivity.sh): PT_EXTRA_VALU dependent adds per inner-node visit, results unchanged
    { float acc = q1.w;
#pragma unroll
      for (int i = 0; i < PT_EXTRA_VALU; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(q3.w)); }
#endif
    const bool full = inner_visit(r, stk, q0, q1, q2, q3);
    if (STK::guarded) r.set_overflow(stop || full);
    if (!KIND) return (stop || full) ? 1 : 0;
    return (stop || full) ? PT_K_DONE : kind_after<WIDE>(r);
}
// Tolerance mode: one 64-B record holds up to four quantised child boxes (pt_internal.h).  The nearest valid child is taken, the others
// are pushed farthest first.  One exit, like op_inner.  Written for instruction count (the body is VALU work between two dependent
// fetches): the ray is brought into the node's frame once -- t = (org + q * step - o) * id = q * (step * id) + (org - o) * id, a plane
// costs one byte-to-float convert and one fma -- the near and far plane words of an axis are chosen once per node by the sign of the
// direction, and the four validity tests are selects, not branches.  (The boxes are rounded outwards with a pad of >= 2e-6 of the
// coordinate and 0.2 % of the frame, which covers what this rounding moves; a plane that comes out NaN -- a direction component of
// exactly 0 -- is skipped by min3 / max3, which can only widen the box.)
__device__ __forceinline__ float ubyte_f32(unsigned w, int c) { return (float)((w >> (8 * c)) & 0xffu); }     // v_cvt_f32_ubyte<c>
__device__ __forceinline__ float fmax3_(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float fmin3_(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// (The fused pass of op_inner_leaf -- this body also serving the lanes at a triangle -- was measured here too: 10 M grid -2.8 %, 100
// instances +0.2 %, profiles/r03_extend_ab_fused_pass.txt; not kept.)
template <bool COUNT, class STK>
__device__ __forceinline__ int op_wide(const SceneView& sc, Ray& r, const STK& stk) {
    r.iter++;
    const bool stop = STK::guarded && r.iter > PT_MAX_ITERS;
    const float4* nd = rec32<PT_WIDE_QUADS>(sc.wide, r.node);
    const float4 h = nd[0], rf = nd[1], qa = nd[2], qb = nd[3];
    if (COUNT) r.n_inner += stop ? 0 : 1;                       // one 64-B record
    const f3 a = F3(h.w, qb.z, qb.w) * r.id, b = (F3(h) - r.o) * r.id;
    const unsigned lx = __float_as_uint(qa.x), ly = __float_as_uint(qa.y), lz = __float_as_uint(qa.z), hx = __float_as_uint(qa.w),
                   hy = __float_as_uint(qb.x), hz = __float_as_uint(qb.y);
    const bool bx = r.id.x < 0.0f, by = r.id.y < 0.0f, bz = r.id.z < 0.0f;
    const unsigned nx = bx ? hx : lx, fx = bx ? lx : hx, ny = by ? hy : ly, fy = by ? ly : hy, nz = bz ? hz : lz, fz = bz ? lz : hz;
    int r0 = __float_as_int(rf.x), r1 = __float_as_int(rf.y), r2 = __float_as_int(rf.z), r3 = __float_as_int(rf.w);
    const float INF = __builtin_inff(), lim = r.hit_dist;
#define PT_CHILD_T(c, ref)                                                                                                             \
    [&]() -> float {                                                                                                                   \
        const float tn = fmax3_(__builtin_fmaf(ubyte_f32(nx, c), a.x, b.x), __builtin_fmaf(ubyte_f32(ny, c), a.y, b.y), __builtin_fmaf(ubyte_f32(nz, c), a.z, b.z)); \
        const float tf = fmin3_(__builtin_fmaf(ubyte_f32(fx, c), a.x, b.x), __builtin_fmaf(ubyte_f32(fy, c), a.y, b.y), __builtin_fmaf(ubyte_f32(fz, c), a.z, b.z)); \
        const bool ok = (fmax_(tn, 0.0f) <= tf) & (tn < lim) & ((ref) != PT_WIDE_EMPTY);                                                \
        return ok ? tn : INF;                                                                                                          \
    }()
    float k0 = PT_CHILD_T(0, r0), k1 = PT_CHILD_T(1, r1), k2 = PT_CHILD_T(2, r2), k3 = PT_CHILD_T(3, r3);
#undef PT_CHILD_T
    near_first(k0, r0, k1, r1); near_first(k2, r2, k3, r3); near_first(k0, r0, k2, r2); near_first(k1, r1, k3, r3); near_first(k1, r1, k2, r2);
    // sorted: the valid children come first, so child j is valid iff its key is finite; children 1..3 are pushed farthest first, i.e.
    // child j lands j entries below the new top of the stack
    const bool p1 = k1 != INF, p2 = k2 != INF, p3 = k3 != INF;
    const int top = r.sp + (int)p1 + (int)p2 + (int)p3;
    const bool full = STK::guarded && top > PT_STACK_SIZE;                 // pushes onto a full stack end the ray
    if (STK::shallow || !__any(top > STK::lds_n)) {                         // ONE wave-uniform test for the three pushes: all of them stay in LDS
        int* t = stk.lds + top * STK::block;
        if (p1) t[-1 * STK::block] = r1;
        if (p2) t[-2 * STK::block] = r2;
        if (p3) t[-3 * STK::block] = r3;
    } else if (!full) {
        if (p1) stk.put(top - 1, r1);
        if (p2) stk.put(top - 2, r2);
        if (p3) stk.put(top - 3, r3);
    }
    r.sp = full ? r.sp : top;
    r.node = r0;
    if (k0 == INF) { r.sp -= 1; r.node = stk.get(r.sp); }
    if (STK::guarded) r.set_overflow(stop || full);
    return (stop || full) ? PT_K_DONE : kind_after<true>(r);
}
// WIDE: the kind-scheduled wide walk keeps what only the triangle and switch bodies touch -- the direction, hit_obj, hit_tri -- in five
// LDS words per lane behind the stack ("parked", Stack::park): the 4-wide node body is the one that needs the registers.
template <bool COUNT, class STK, bool WIDE = false, bool KIND = true>
__device__ __forceinline__ int op_leaf(const SceneView& sc, Ray& r, const STK& stk) {       // :380-391
    r.iter++;
    const bool stop = STK::guarded && r.iter > PT_MAX_ITERS;                // as in op_inner: one exit; a stopped lane's test changes nothing
    int t = -(r.node + 1);
    const float4* tr = rec32<3>(sc.tris, t);
    const f3 d = WIDE ? F3(__int_as_float(stk.park(0)), __int_as_float(stk.park(1)), __int_as_float(stk.park(2))) : r.d;
    float dist = tri_hit(r.o, d, tr[0], tr[1], tr[2]);
    if (!stop && dist > 0.0f && dist < r.hit_dist) {
        r.hit_dist = dist;
        if (WIDE) { stk.park(3) = r.obj; stk.park(4) = t; } else { r.hit_tri = t; r.hit_obj = r.obj; }
    }
    if (COUNT) r.n_leaf += stop ? 0 : 1;
    r.sp -= 1; r.node = stk.get(r.sp);
    if (STK::guarded) r.set_overflow(stop);
    if (!KIND) return stop ? 1 : 0;
    return stop ? PT_K_DONE : kind_after<WIDE>(r);
}
// leaving (:330-355, rows 0-2 of the object = localToWorld) and entering (:357-379, rows 4-6 = worldToLocal, row 3 = root and
// material) run the same arithmetic on different rows: one body serves both groups of lanes
template <bool COUNT, class STK, bool WIDE = false, bool KIND = true>
__device__ __forceinline__ int op_switch(const SceneView& sc, Ray& r, const STK& stk) {
    const bool entering = r.obj < 0;
    if (entering && !trip_guard<STK>(r)) return KIND ? PT_K_DONE : 1;
    const int oi = entering ? -(r.node + 1) : r.obj;
    const float4* ob = (WIDE ? sc.objects_wide : sc.objects) + 8 * (int64_t)oi + (entering ? 4 : 0);
    float4 r0 = ob[0], r1 = ob[1], r2 = ob[2];
    if (entering) { r.node = __float_as_int(ob[-1].x); r.switch_pt = r.sp; r.obj = oi; if (COUNT) r.n_enter++; }
    else { r.switch_pt = -1; r.obj = -1; if (COUNT) r.n_leave++; }
    if (WIDE) r.d = F3(__int_as_float(stk.park(0)), __int_as_float(stk.park(1)), __int_as_float(stk.park(2)));
    switch_space(r0, r1, r2, r);
    if (WIDE) { stk.park(0) = __float_as_int(r.d.x); stk.park(1) = __float_as_int(r.d.y); stk.park(2) = __float_as_int(r.d.z); }
    if (!KIND) return 0;
    return kind_after<WIDE>(r);
}
// FUSED (extend variant 9): one pass for the lanes at an inner node AND the lanes at a triangle.  Both records are requested by the same
// four load instructions (a lane's address is its node's or its triangle's; the fourth quad is only loaded for a node), then the two
// groups run their arithmetic one after the other: the vector instructions of an inner pass plus a triangle pass, ONE trip to memory.
// Run where the schedule chooses the triangle body while lanes wait at inner nodes: their visit would be the next pass anyway.
// (Serving the lanes at an object switch too, in the passes that choose the switch body, was measured: it needs 66 registers -- forced
// to 64 it spills outside the loop only -- and gives +0.7 % on 100 instances, -3 % on configs[2]: profiles/r03_extend_ab_fused_pass.txt.)
template <bool COUNT, class STK>
__device__ __forceinline__ int op_inner_leaf(const SceneView& sc, Ray& r, const STK& stk) {
    r.iter++;
    const bool stop = STK::guarded && r.iter > PT_MAX_ITERS;
    const bool leaf = r.node < 0;
    const int t = -(r.node + 1);
    const char* rec_b = leaf ? (const char*)sc.tris + (uint32_t)t * 48u : (const char*)sc.nodes + (uint32_t)r.node * 64u;
    const float4* rec = (const float4*)rec_b;
    float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (!leaf) q3 = rec[3];
    bool full = false;
    if (leaf) {                                                                 // :380-391, as op_leaf
        float dist = tri_hit(r.o, r.d, q0, q1, q2);
        if (!stop && dist > 0.0f && dist < r.hit_dist) { r.hit_dist = dist; r.hit_tri = t; r.hit_obj = r.obj; }
        if (COUNT) r.n_leaf += stop ? 0 : 1;
        r.sp -= 1; r.node = stk.get(r.sp);
    } else {                                                                    // :392-439, as op_inner
        if (COUNT) r.n_inner += stop ? 0 : 1;
        full = inner_visit(r, stk, q0, q1, q2, q3);
    }
    if (STK::guarded) r.set_overflow(stop || full);
    return (stop || full) ? PT_K_DONE : kind_after<false>(r);
}
// Runs the walk of every lane of the wave to its end.  `kind`: PT_K_INNER for a lane with a ray, PT_K_DONE for one without.
template <bool COUNT, class STK, int TOP = 0>
__device__ __forceinline__ void traverse_scheduled(const SceneView& sc, Ray& r, const STK& stk, int kind, LdsQuads lds = nullptr, StepHeader* hdr = nullptr) {
    int np_i = 0, np_l = 0, np_s = 0;      // COUNT: passes by body (measurement: lane use = lane visits / (64 x passes))
    int lh[4] = {0, 0, 0, 0};              // COUNT: node-body passes by active lanes
    // Same sequence of passes as "count, choose, run one body" in one loop; written as a tight loop per body inside the outer loop so
    // that a run of passes of one kind keeps the values that body does not touch where they are (no copies at the joins).
    int ci, cl, cs;
    auto count = [&]() {
        ci = __popcll(__ballot(kind == PT_K_INNER)) * sc.sched_wi; cl = __popcll(__ballot(kind == PT_K_LEAF)) * sc.sched_wl;
        cs = __popcll(__ballot(kind == PT_K_SWITCH)) * sc.sched_ws;
    };
    // The body that serves most lanes, weighted.  Weights by cost alone (lanes advanced per instruction issued: the bodies cost
    // about 1 : 1.4 : 2.4, i.e. 12 : 9 : 5) were the first choice; the sweep (profiles/r02_sched_weight_sweep.txt) prefers the inner
    // body at half that weight -- lanes parked at a triangle or an object switch then rejoin the others sooner: +3.7 % on the
    // 82 k-triangle mesh, +6 % on the instanced scene, Cornell unchanged; below 4 it falls off again -- except on a one-object scene,
    // whose walk has no switches to wait for: there 3 is best (sc.sched_wi, set by the host; PT_SCHED_W overrides).
    count();
    while ((ci | cl | cs) != 0) {
        if (ci >= cl && ci >= cs) {
            do { if (COUNT) lh[(__popcll(__ballot(kind == PT_K_INNER)) - 1) >> 4]++;
                 if (kind == PT_K_INNER) kind = op_inner<COUNT, STK, TOP>(sc, r, stk, lds);
                 count(); if (COUNT) np_i++; } while (ci != 0 && ci >= cl && ci >= cs);
        } else if (cl >= cs) {
            do { if (TOP == 4) { if (kind == PT_K_LEAF || kind == PT_K_INNER) kind = op_inner_leaf<COUNT, STK>(sc, r, stk); }
                 else if (kind == PT_K_LEAF) kind = op_leaf<COUNT, STK>(sc, r, stk);
                 count(); if (COUNT) np_l++; } while (cl != 0 && ci < cl && cl >= cs);
        } else {
            if (kind == PT_K_SWITCH) kind = op_switch<COUNT, STK>(sc, r, stk);
            count(); if (COUNT) np_s++;
        }
    }
    if (COUNT && hdr && lane_id() == 0) {
        atomicAdd(&hdr->passes[0], (unsigned long long)np_i); atomicAdd(&hdr->passes[2], (unsigned long long)np_l); atomicAdd(&hdr->passes[3], (unsigned long long)np_s);
        for (int k = 0; k < 4; k++) atomicAdd(&hdr->lane_hist[k], (unsigned long long)lh[k]);
    }
}

// The same schedule with the lanes' kinds kept as three wave-wide masks in scalar registers instead of a kind per lane (MASKS, extend
// variant 8).  A pass of traverse_scheduled pays for its bookkeeping in vector instructions: kind_after (four compares, three selects),
// count() (three compares of the kind) and the body's own `kind == K` test -- eleven of the inner body's ~75.  Here a body runs under its
// mask directly (inverse ballot: the mask IS the exec mask), and after it three compares under the full exec mask -- node >= 0,
// sp < switch_pt, sp < 0 -- give the ballots from which scalar and/or/andn2 re-sort the lanes that ran; "inside an object" is a fourth
// mask that only the switch body changes.  Same passes in the same order, same arithmetic per lane: bit-exact with variant 4.
// Measured slower (profiles/r03_extend_ab_kind_masks.txt: -2...-4 % on identical queues, -1.6 % in the configs[2] render): 8 vector
// instructions fewer per pass, 19 scalar ones more, in a dependent chain on the unit a CU's four SIMDs share.  Kept as an A/B variant.
template <bool COUNT, class STK>
__device__ __forceinline__ void traverse_masked(const SceneView& sc, Ray& r, const STK& stk, bool has_ray, StepHeader* hdr = nullptr) {
    typedef unsigned long long u64;
    u64 mi = __ballot(has_ray), ml = 0, ms = 0, in_obj = 0;
    int np_i = 0, np_l = 0, np_s = 0, lh[4] = {0, 0, 0, 0};
    // the lanes of `ran` are re-sorted by where the body left them (kind_after's rules); `dead`: lanes a guard has stopped
    auto resort = [&](u64 ran, u64 dead) {
        const u64 b_node = __ballot(r.node >= 0), b_leave = __ballot(r.sp < r.switch_pt), b_done = __ballot(r.sp < 0) | dead;
        const u64 live = ran & ~b_done;
        const u64 sw = live & (b_leave | (~b_node & ~in_obj));          // :330 leave first; :357 a TLAS leaf enters an object
        const u64 inner = live & ~sw & b_node;
        mi = (mi & ~ran) | inner; ms = (ms & ~ran) | sw; ml = (ml & ~ran) | (live & ~sw & ~b_node);
    };
    for (;;) {
        int ci = __popcll(mi) * sc.sched_wi, cl = __popcll(ml) * sc.sched_wl, cs = __popcll(ms) * sc.sched_ws;
        if ((ci | cl | cs) == 0) break;
        if (ci >= cl && ci >= cs) {
            do {
                if (COUNT) { lh[(__popcll(mi) - 1) >> 4]++; np_i++; }
                const u64 ran = mi; int dead = 0;
                if (__builtin_amdgcn_inverse_ballot_w64(ran)) dead = op_inner<COUNT, STK, 0, false, false>(sc, r, stk);
                resort(ran, STK::guarded ? __ballot(dead != 0) : 0ull);
                ci = __popcll(mi) * sc.sched_wi; cl = __popcll(ml) * sc.sched_wl; cs = __popcll(ms) * sc.sched_ws;
            } while (ci != 0 && ci >= cl && ci >= cs);
        } else if (cl >= cs) {
            do {
                if (COUNT) np_l++;
                const u64 ran = ml; int dead = 0;
                if (__builtin_amdgcn_inverse_ballot_w64(ran)) dead = op_leaf<COUNT, STK, false, false>(sc, r, stk);
                resort(ran, STK::guarded ? __ballot(dead != 0) : 0ull);
                ci = __popcll(mi) * sc.sched_wi; cl = __popcll(ml) * sc.sched_wl; cs = __popcll(ms) * sc.sched_ws;
            } while (cl != 0 && ci < cl && cl >= cs);
        } else {
            if (COUNT) np_s++;
            const u64 ran = ms; int dead = 0;
            if (__builtin_amdgcn_inverse_ballot_w64(ran)) dead = op_switch<COUNT, STK, false, false>(sc, r, stk);
            const u64 d = STK::guarded ? __ballot(dead != 0) : 0ull;
            in_obj = (in_obj & ~ran) | (ran & ~in_obj & ~d);          // the lanes that entered are inside now, the ones that left are not (a stopped lane is never looked at again)
            resort(ran, d);
        }
    }
    if (COUNT && hdr && lane_id() == 0) {
        atomicAdd(&hdr->passes[0], (unsigned long long)np_i); atomicAdd(&hdr->passes[2], (unsigned long long)np_l); atomicAdd(&hdr->passes[3], (unsigned long long)np_s);
        for (int k = 0; k < 4; k++) atomicAdd(&hdr->lane_hist[k], (unsigned long long)lh[k]);
    }
}

// The same for the tolerance mode: four bodies (binary TLAS node, 4-wide BLAS node, triangle, object switch).
template <bool COUNT, class STK>
__device__ __forceinline__ void traverse_scheduled_wide(const SceneView& sc, Ray& r, const STK& stk, int kind, StepHeader* hdr = nullptr) {
    int ci, cw, cl, cs, np_i = 0, np_w = 0, np_l = 0, np_s = 0, lh[4] = {0, 0, 0, 0};
    stk.park(0) = __float_as_int(r.d.x); stk.park(1) = __float_as_int(r.d.y); stk.park(2) = __float_as_int(r.d.z); stk.park(3) = r.hit_obj; stk.park(4) = r.hit_tri;
    auto count = [&]() {
        ci = __popcll(__ballot(kind == PT_K_INNER)) * sc.sched_wi; cw = __popcll(__ballot(kind == PT_K_WIDE)) * sc.sched_ww;
        cl = __popcll(__ballot(kind == PT_K_LEAF)) * sc.sched_wl; cs = __popcll(__ballot(kind == PT_K_SWITCH)) * sc.sched_ws;
    };
    count();
    while ((ci | cw | cl | cs) != 0) {
        if (cw >= ci && cw >= cl && cw >= cs) {
            do { if (COUNT) lh[(__popcll(__ballot(kind == PT_K_WIDE)) - 1) >> 4]++;
                 if (kind == PT_K_WIDE) kind = op_wide<COUNT, STK>(sc, r, stk); count(); if (COUNT) np_w++; } while (cw != 0 && cw >= ci && cw >= cl && cw >= cs);
        } else if (cl >= ci && cl >= cs) {
            do { if (kind == PT_K_LEAF) kind = op_leaf<COUNT, STK, true>(sc, r, stk); count(); if (COUNT) np_l++; } while (cl != 0 && cw < cl && cl >= ci && cl >= cs);
        } else if (ci >= cs) {
            if (kind == PT_K_INNER) kind = op_inner<COUNT, STK, false, true>(sc, r, stk);
            count(); if (COUNT) np_i++;
        } else {
            if (kind == PT_K_SWITCH) kind = op_switch<COUNT, STK, true>(sc, r, stk);
            count(); if (COUNT) np_s++;
        }
    }
    r.d = F3(__int_as_float(stk.park(0)), __int_as_float(stk.park(1)), __int_as_float(stk.park(2))); r.hit_obj = stk.park(3); r.hit_tri = stk.park(4);
    if (COUNT && hdr && lane_id() == 0) {
        atomicAdd(&hdr->passes[0], (unsigned long long)np_i); atomicAdd(&hdr->passes[1], (unsigned long long)np_w);
        atomicAdd(&hdr->passes[2], (unsigned long long)np_l); atomicAdd(&hdr->passes[3], (unsigned long long)np_s);
        for (int k = 0; k < 4; k++) atomicAdd(&hdr->lane_hist[k], (unsigned long long)lh[k]);
    }
}

// scene.js:442-466: a ray that ends inside an object is mapped back to world space
template <bool COUNT>
__device__ __forceinline__ void trav_finish(const SceneView& sc, Ray& r) {
    if (r.obj >= 0) {
        const float4* ob = sc.objects + 8 * (int64_t)r.obj;
        switch_space(ob[0], ob[1], ob[2], r);
        if (COUNT) r.n_leave++;
    }
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}
// overflow count (always) and visit counts (counting variant) -> StepHeader, one atomic per wave and counter
template <bool COUNT>
__device__ __forceinline__ void publish(unsigned long long ov, unsigned long long a, unsigned long long b, unsigned long long c,
                                        unsigned long long e, StepHeader* hdr) {
    if (__any(ov != 0)) { ov = wave_sum(ov); if (lane_id() == 0) atomicAdd(&hdr->overflow, ov); }
    if (COUNT) {
        a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); e = wave_sum(e);
        if (lane_id() == 0) {
            unsigned long long* v = hdr->visits[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 63];
            atomicAdd(v + 0, a); atomicAdd(v + 1, b); atomicAdd(v + 2, c); atomicAdd(v + 3, e);
        }
    }
}

// --------------------------------------------------------------------------------------------
// The extend kernel: one launch thread per ray-queue entry (the reference's mapping, ray-trace.js:55-60).
// --------------------------------------------------------------------------------------------
// LOG (measurement only, tools/wave_timeline.py): every wave records {start, end} in 100 MHz ticks, where it ran and how
// many trips it took -- wave_log[4 * wave] = {t0, t1, XCC_ID << 32 | HW_ID, block << 32 | class << 24 | trips}.
// LOGIC: the thread that traced a slot's ray goes straight on with that slot's logic (logic.js:62-152): it touches only
// the slot's own state, the hit is still in registers, and its memory traffic hides under the traversal of the other
// waves instead of costing a launch.  (The material sample too was tried here: its ~90 registers take the traversal from
// 8 to 5 waves per SIMD, +30 us per launch; capped at 64 registers it spills, +43 us.  It runs in the compaction launch.)
// SCENE_QUADS > 0: a scene whose four buffers together fit in SCENE_QUADS quads (12 per object + 7 per triangle) is copied
// into LDS by every workgroup and traversed from there: the same records, an LDS read instead of an L1 round trip in
// every link of the trip's dependent chain.
// TOP (with SCHED; SCENE_QUADS is then the LDS budget, not a whole-scene limit): the TLAS, the objects and the tree-top blocks of the
// meshes (SceneView::top_*) are staged instead of the whole scene, and the walk starts from objects_top (set by the launcher).
// WIDE (tolerance mode): the walk inside an object uses the 4-wide collapse.  CHECK (with WIDE): the rays of one queue block in
// rc.check_period -- which ones rotates with the step -- are walked a second time with the reference walk, and how the two walks differ
// is added up in the StepHeader (chk_*); the wide walk's hit is what the slot gets either way.  The second walk is not done by the wave
// that did the first: a wave that walks twice lasts twice as long, and a launch of one pool is only one and a half generations of
// waves deep, so such waves end the launch with a tail of their own wherever they are dispatched (measured: -10 % of the render for
// 1.6 % of the rays, -24 % when spread over the launch).  Instead the first walk leaves the ray and its hit in a side buffer
// (rc.chk_buf, two halves by step parity), and the NEXT step's launch starts with rc.chk_cap extra workgroups that take those rays,
// walk them exactly and compare: they are dispatched first, last as long as an ordinary wave, and share nothing with the other
// workgroups but the counters.  (pt_render_run always issues at least one step after the last one that traced a ray.)
template <int LDS_N, int BLOCK, bool COUNT, int ORDER, bool LOG, bool LOGIC, bool WIDE = false, int SCENE_QUADS = 0, bool SHALLOW = false, bool SCHED = false, int TOP = 0, bool CHECK = false, bool GUARDED = true>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_num_sgpr(96))) void k_extend_queue(PathState ps, SceneView sc, StepHeader* hdr, const int32_t* __restrict__ cam_q,
                                                        const int32_t* __restrict__ mat_q, int32_t* spill, int64_t spill_stride,
                                                        unsigned long long* wave_log, float* __restrict__ image, uint8_t* __restrict__ code_out,
                                                        unsigned long long* __restrict__ rec_next, RenderConst rc) {
    unsigned long long t_start = LOG ? wall_clock64() : 0;
    __shared__ int s_stack[(LDS_N + ((WIDE && SCHED) ? 5 : 0)) * BLOCK];     // + the wide walk's parked words (op_leaf)
    int n_cam = hdr->n_camera, n_rays = n_cam + hdr->n_material;
    // ORDER 2: a launch lasts as long as its longest wave plus the time that wave waited to start
    // (profiles/r01_wave_timeline_*), so the class of rays that held the longest wave in the previous step is dispatched
    // first: queue order when that was a camera wave, reversed when it was a bounce wave (workgroup 0 of the compaction launch decides).
    typedef Stack<LDS_N, BLOCK, SHALLOW, GUARDED> STK;
    if (CHECK && (int)blockIdx.x < rc.chk_cap) {
        // ---- one of the workgroups in front of the launch: the reference walk of the rays the previous step set aside ----
        const int P = rc.check_period, s = hdr->step_index;      // the compaction launch has already counted this step: s >= 1
        if (P <= 0) return;
        const int n_prev = hdr->prev_n_rays, nblk_prev = (n_prev + BLOCK - 1) / BLOCK, pp = (s - 1) % P, x = (int)blockIdx.x;
        if (pp >= nblk_prev || x >= (nblk_prev - pp + P - 1) / P) return;
        const bool valid = (x * P + pp) * BLOCK + (int)threadIdx.x < n_prev;
        const float* S = rc.chk_buf + ((int64_t)(((s - 1) & 1) * rc.chk_cap + x) * 9) * BLOCK + threadIdx.x;
        Ray e; ray_init(e, F3(0, 0, 1), F3(0, 0, 1), sc.tlas_root);
        if (valid) ray_init(e, F3(S[0], S[BLOCK], S[2 * BLOCK]), F3(S[3 * BLOCK], S[4 * BLOCK], S[5 * BLOCK]), sc.tlas_root);
        STK stk{s_stack + threadIdx.x, rc.chk_spill, x * BLOCK + (int)threadIdx.x, (int64_t)rc.chk_cap * BLOCK};
        traverse_scheduled<false, STK, false>(sc, e, stk, valid ? PT_K_INNER : PT_K_DONE);
        bool differ = false, flip = false, beyond = false, lost = false; float rel = 0.0f;
        if (valid) {
            trav_finish<false>(sc, e);
            const int w_obj = __float_as_int(S[6 * BLOCK]), w_tri = __float_as_int(S[7 * BLOCK]); const float w_dist = S[8 * BLOCK];
            differ = w_obj != e.hit_obj || w_tri != e.hit_tri;
            flip = (w_obj < 0) != (e.hit_obj < 0);
            if (w_obj >= 0 && e.hit_obj >= 0) rel = __builtin_fabsf(w_dist - e.hit_dist) / e.hit_dist;
            beyond = flip || rel > 1e-6f;                        // north_star's allowance: "grazing/edge rays within 1e-6"
            lost = beyond && (w_obj < 0 || (e.hit_obj >= 0 && w_dist > e.hit_dist));
        }
        const unsigned long long m_all = __ballot(valid), m_diff = __ballot(differ), m_flip = __ballot(flip), m_bey = __ballot(beyond), m_lost = __ballot(lost);
        for (int off = 32; off > 0; off >>= 1) rel = fmax_(rel, __shfl_xor(rel, off));
        if (lane_id() == 0) {
            atomicAdd(&hdr->chk_rays, (unsigned long long)__popcll(m_all));
            if (m_diff) atomicAdd(&hdr->chk_mismatch, (unsigned long long)__popcll(m_diff));
            if (m_flip) atomicAdd(&hdr->chk_flips, (unsigned long long)__popcll(m_flip));
            if (m_bey) atomicAdd(&hdr->chk_beyond, (unsigned long long)__popcll(m_bey));
            if (m_lost) atomicAdd(&hdr->chk_lost, (unsigned long long)__popcll(m_lost));
            if (__float_as_uint(rel) > hdr->chk_max_rel_bits) atomicMax(&hdr->chk_max_rel_bits, __float_as_uint(rel));
        }
        publish<false>(e.overflow(), 0, 0, 0, 0, hdr);
        return;
    }
    int bx = (int)blockIdx.x - (CHECK ? rc.chk_cap : 0);
    if (rc.xcd_chunk > 0) {
        // The hardware hands workgroup i to XCD i % 8, each with an L2 of its own: with block i on workgroup i eight neighbouring blocks -- rays
        // through neighbouring pixels, the same nodes -- run on eight XCDs and every L2 fetches those nodes for itself.  Here every group of
        // 8 C workgroups gives each XCD C consecutive blocks instead (the launcher rounds the grid up to whole groups).
        const int C = rc.xcd_chunk, G = 8 * C, g = bx / G, r = bx - g * G;
        bx = g * G + (r & 7) * C + (r >> 3);
    }
    int nblk = (n_rays + BLOCK - 1) / BLOCK;
    int blk = (ORDER == 2 && hdr->extend_order) ? nblk - 1 - bx : bx;
    if (blk < 0 || blk >= nblk) return;
    if (LOG && lane_id() == 0) wave_log[4 * (int64_t)(blockIdx.x * (BLOCK >> 6) + (threadIdx.x >> 6)) + 1] = 0;
    // dynamic LDS, sized by the launcher to the scene's own quads (not to the SCENE_QUADS limit): what a workgroup does not
    // take lets another one start in the wave slots that the finished waves of resident workgroups have already left.
    // (Sizing the stack part to the scene's stack bound as well was measured: the run-time LDS base costs more than it returns.)
    extern __shared__ float4 s_scene[];
    if (TOP == 1) {                                              // [nodes from id n_tris on: tlas, top blocks | objects (with top roots)]
        const int t4 = 4 * (sc.n_objects + sc.top_nodes), total = t4 + 8 * sc.n_objects;
        const float4* high = sc.nodes + 4 * (int64_t)sc.n_tris;
        for (int i = threadIdx.x; i < total; i += BLOCK) s_scene[i] = i < t4 ? high[i] : sc.objects[i - t4];
        __syncthreads();
        sc.objects = s_scene + t4;
    } else if (SCENE_QUADS > 0) {                                // [nodes: blas, tlas | objects | tris]
        const int n4 = 4 * (sc.n_tris + sc.n_objects), m8 = 8 * sc.n_objects, total = n4 + m8 + 3 * sc.n_tris;
        for (int i = threadIdx.x; i < total; i += BLOCK)
            s_scene[i] = i < n4 ? sc.nodes[i] : i < n4 + m8 ? sc.objects[i - n4] : sc.tris[i - n4 - m8];
        __syncthreads();
        sc.nodes = s_scene; sc.objects = s_scene + n4; sc.tris = s_scene + n4 + m8;
    }
    int q = blk * BLOCK + threadIdx.x;
    Ray r; ray_init(r, F3(0, 0, 1), F3(0, 0, 1), sc.tlas_root);
    unsigned code = 0u; int grp = -1;
    STK stk{s_stack + threadIdx.x, spill, q, spill_stride};
    int path = -1;
    if (q < n_rays) {
        path = q < n_cam ? cam_q[q] : mat_q[q - n_cam];
        f3 ro, rd;
        if (LOGIC && q < n_cam) generate_one(ps, rc, path, ro, rd);          // camera.js:60-118; replays (no LOGIC) re-read the stored ray
        else { ro = F3(ps.path_o[path]); rd = F3(ps.path_d[path]); }
        ray_init(r, ro, rd, sc.tlas_root);
        if (!SCHED) while (trav_step<COUNT, STK, WIDE>(sc, r, stk)) {}
    }
    if (SCHED && WIDE) traverse_scheduled_wide<COUNT, STK>(sc, r, stk, q < n_rays ? PT_K_INNER : PT_K_DONE, hdr);
    else if (SCHED && TOP == 3) traverse_masked<COUNT, STK>(sc, r, stk, q < n_rays, hdr);                     // the kinds as wave-wide masks
    else if (SCHED) traverse_scheduled<COUNT, STK, TOP>(sc, r, stk, q < n_rays ? PT_K_INNER : PT_K_DONE, (LdsQuads)s_scene, hdr);     // the whole wave walks together
    if (q < n_rays) {
        trav_finish<COUNT>(sc, r);
        ps.hit[path] = make_int2(r.hit_obj, r.hit_tri);
    }
    if (CHECK && rc.check_period > 0 && blk % rc.check_period == hdr->step_index % rc.check_period && q < n_rays) {
        // set aside for the reference walk at the front of the next step's launch (generate_one has stored the camera ray)
        float* S = rc.chk_buf + ((int64_t)((hdr->step_index & 1) * rc.chk_cap + blk / rc.check_period) * 9) * BLOCK + threadIdx.x;
        const P3 so = ps.path_o[path], sd = ps.path_d[path];
        S[0] = so.x; S[BLOCK] = so.y; S[2 * BLOCK] = so.z; S[3 * BLOCK] = sd.x; S[4 * BLOCK] = sd.y; S[5 * BLOCK] = sd.z;
        S[6 * BLOCK] = __int_as_float(r.hit_obj); S[7 * BLOCK] = __int_as_float(r.hit_tri); S[8 * BLOCK] = r.hit_dist;
    }
    if (q < n_rays) {
        if (LOGIC) {
            code = logic_one(ps, image, path, r.hit_obj >= 0, rc.max_nb);
            code_out[path] = (uint8_t)code;
            grp = path / PT_STEP_BLOCK;
        }
    }
    if (LOGIC) {
        // The next step's count records (one per PT_STEP_BLOCK slots: #camera | #material << 16 | #tickets << 32 | #finished << 48)
        // are added up here, one 64-bit atomic per wave and record: queue entries are in ascending slot order, so a wave's
        // 64 slots fall into one or two records.  (This replaces a counting launch over the codes between extend and compaction.)
        unsigned long long todo = __ballot(grp >= 0);
        while (todo) {
            int g = __shfl(grp, (int)__builtin_ctzll(todo));
            bool in = grp == g;
            unsigned long long m = __ballot(in);
            unsigned long long packed = (unsigned long long)__popcll(__ballot(in && (code & 1u))) | ((unsigned long long)__popcll(__ballot(in && (code & 2u))) << 16) |
                                        ((unsigned long long)__popcll(__ballot(in && (code & 4u))) << 32) | ((unsigned long long)__popcll(__ballot(in && (code & 8u))) << 48);
            if (lane_id() == (int)__builtin_ctzll(m) && packed) atomicAdd(rec_next + 2 * (int64_t)g, packed);
            todo &= ~m;
        }
    }
    publish<COUNT>(r.overflow(), r.n_inner, r.n_leaf, r.n_enter, r.n_leave, hdr);
    if (LOG) {
        int trips = r.trips();
        for (int off = 32; off > 0; off >>= 1) trips = max(trips, __shfl_xor(trips, off));
        if (lane_id() == 0) {
            unsigned long long* w = wave_log + 4 * (int64_t)(blockIdx.x * (BLOCK >> 6) + (threadIdx.x >> 6));
            unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20), hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            w[0] = t_start; w[1] = wall_clock64(); w[2] = ((unsigned long long)xcc << 32) | hw;
            w[3] = ((unsigned long long)blk << 32) | ((q < n_cam ? 0ull : 1ull) << 24) | (unsigned)min(trips, 0xffffff);
        }
    }
    if (ORDER == 2 && blockIdx.x % 5 == 0) {     // one workgroup in five (all XCDs) reports its waves' trip counts
        int trips = r.trips();
        for (int off = 32; off > 0; off >>= 1) trips = max(trips, __shfl_xor(trips, off));
        if (lane_id() == 0) {
            unsigned* cell = &hdr->cls_max[(blockIdx.x / 5 * (BLOCK >> 6) + (threadIdx.x >> 6)) & 31][q < n_cam ? 0 : 1];
            if ((unsigned)trips > *cell) atomicMax(cell, (unsigned)trips);   // rarely taken once the maximum has settled
        }
    }
}

// --------------------------------------------------------------------------------------------
// Test hook: caller-supplied rays, always counting (pt_trace_rays)
// --------------------------------------------------------------------------------------------
// STRIDE: floats per ray in o / d (4: the staged float4 arrays of pt_trace_rays; 3: the caller's packed arrays, pt_trace_rays_noalloc).
template <bool WIDE, int STRIDE>
__global__ __launch_bounds__(PT_BLOCK) void k_extend_direct(SceneView sc, const float* __restrict__ o, const float* __restrict__ d, int n,
                                                            int32_t* __restrict__ hit_obj, int32_t* __restrict__ hit_tri,
                                                            float* __restrict__ hit_dist, int4* __restrict__ counts4, int32_t* spill,
                                                            int64_t spill_stride, StepHeader* hdr) {
    __shared__ int s_stack[PT_LDS_STACK * PT_BLOCK];
    int q = blockIdx.x * PT_BLOCK + threadIdx.x;
    Ray r; ray_init(r, F3(0, 0, 1), F3(0, 0, 1), sc.tlas_root);
    if (q < n) {
        if (STRIDE == 4) ray_init(r, F3(((const float4*)o)[q]), F3(((const float4*)d)[q]), sc.tlas_root);
        else ray_init(r, F3(o[3 * (int64_t)q], o[3 * (int64_t)q + 1], o[3 * (int64_t)q + 2]), F3(d[3 * (int64_t)q], d[3 * (int64_t)q + 1], d[3 * (int64_t)q + 2]), sc.tlas_root);
        Stack<PT_LDS_STACK, PT_BLOCK> stk{s_stack + threadIdx.x, spill, q, spill_stride};
        while (trav_step<true, Stack<PT_LDS_STACK, PT_BLOCK>, WIDE>(sc, r, stk)) {}
        trav_finish<true>(sc, r);
        if (hit_obj) hit_obj[q] = r.hit_obj;
        if (hit_tri) hit_tri[q] = r.hit_tri;
        if (hit_dist) hit_dist[q] = r.hit_dist;
        if (counts4) counts4[q] = make_int4(r.n_inner, r.n_leaf, r.n_enter, r.n_leave);
    }
    publish<false>(r.overflow(), 0, 0, 0, 0, hdr);
}

// ---- launchers -------------------------------------------------------------------------------
static int64_t pad_to(int64_t n, int64_t m) { return (n + m - 1) / m * m; }
// The spill is sized for the smallest LDS share any variant uses (12 entries) and the largest slot count.
size_t ptk_spill_ints(int64_t n_rays) { return (size_t)pad_to(n_rays, 256) * (PT_STACK_SIZE - PT_STAGED_LDS_STACK); }

template <int LDS_N, int BLOCK, int ORDER, bool WIDE = false, int SCENE_QUADS = 0, bool SHALLOW = false, bool SCHED = false, int TOP = 0, bool CHECK = false, bool GUARDED = true>
static void launch_queue(const PathState& ps, int num_paths, int grid_rays, SceneView sc, StepHeader* hdr, const int32_t* cam_q,
                         const int32_t* mat_q, int32_t* spill, int count, unsigned long long* wave_log, float* image, uint8_t* code,
                         unsigned long long* rec_next, const RenderConst& rc, hipStream_t s) {
    int64_t stride = pad_to(num_paths, 256);
    // grid_rays: an upper bound of this step's queue length the host knows (the pool size until the tickets have run out; then
    // the last queue length it has seen: once no ticket is left, a step's rays are a subset of the step before's)
    dim3 grid((unsigned)((std::max(1, std::min(num_paths, grid_rays)) + BLOCK - 1) / BLOCK));
    if (rc.xcd_chunk > 0) grid.x = (grid.x + 8 * rc.xcd_chunk - 1) / (8 * rc.xcd_chunk) * (8 * rc.xcd_chunk);     // whole groups (k_extend_queue's XCD mapping)
    const size_t scene_lds = TOP == 1 ? 16 * (size_t)(12 * sc.n_objects + 4 * sc.top_nodes) : SCENE_QUADS > 0 ? 16 * (size_t)(12 * sc.n_objects + 7 * sc.n_tris) : 0;
    if (TOP == 1) sc.objects = sc.objects_top;
#define PT_LAUNCH(COUNT, LOG, LOGIC) hipLaunchKernelGGL((k_extend_queue<LDS_N, BLOCK, COUNT, ORDER, LOG, LOGIC, WIDE, SCENE_QUADS, SHALLOW, SCHED, TOP, CHECK && LOGIC && !(COUNT), GUARDED>), \
                                                       dim3(grid.x + ((CHECK && LOGIC && !(COUNT)) ? (unsigned)rc.chk_cap : 0u)), dim3(BLOCK), scene_lds, s, ps, sc, hdr, \
                                                       cam_q, mat_q, spill, stride, wave_log, image, code, rec_next, rc)
    if (wave_log)    PT_LAUNCH(false, true, false);    // measurement replays (code == null): traversal only, idempotent
    else if (!code)  PT_LAUNCH(false, false, false);
    else if (count)  PT_LAUNCH(true, false, true);
    else             PT_LAUNCH(false, false, true);
#undef PT_LAUNCH
}

// Variants differ in workgroup size and in how much of the stack is in LDS (occupancy); results are identical.
// Measured and dropped (profiles/r01_extend_ab_*): persistent waves with lane refill (1.2-3x slower: rays at different
// depths in one wave lose coherence, and at full occupancy a wave only ever sees two batches), Morton-sorting the queue
// by ray origin (sort costs more than it saves), one memory wait per trip (all lanes classify, issue their four loads
// together and share the switch code; leaving becomes a trip of its own: 6-9 % slower, profiles/r01_extend_ab_*_v6_*), a branch-light "wave-shaped" trip and packed-f32 slab math (no change:
// the trip is bound by the four divergent 16-B node loads per lane, not by VALU count), one shared object-space switch
// site per trip with enter + root test in the same trip (5-6 % slower: the longer dependent chain per trip costs more than
// the shorter code saves; with the final leave inside the loop 27 % slower, because lanes then finish one by one).
// Dispatch order (variant 2) is the one change that paid: always-reversed order gains 12-22 % on Cornell and the instanced
// scene and loses 10-15 % on the single-mesh scene; choosing by mean trips picks wrong there; raising the priority of
// waves past 12 trips changes nothing (profiles/r01_extend_ab_*_v7_*).
// Sensitivity bound: replacing every correctly rounded divide by a bare v_rcp (results no longer exact) gains 7 % on
// Cornell and 4 % on the 82 k-triangle mesh, so a cheaper exact divide could return at most half of that.
static const char* kVariantNames[] = {"256-thread workgroups, 24 LDS stack entries", "64-thread workgroups, 16 LDS stack entries",
                                      "64-thread workgroups, 16 LDS entries, class of the longest wave first",
                                      "as 2; a scene of <= 8 KB is staged in LDS (256-thread workgroups, 12 LDS stack entries)",
                                      "as 3, kind-scheduled: each pass of the loop the wave runs the one body (inner node, triangle, object switch) that advances most lanes",
                                      "as 4 with 256-thread workgroups and 12 LDS stack entries whatever the scene's size (the workgroup shape an LDS-staged tree top needs)",
                                      "as 5; the TLAS, the objects and the first levels of every mesh's tree are staged in LDS (scenes too large to stage whole)",
                                      "as 4 with the walk's node records split into four arrays of quads (structure of arrays)",
                                      "as 4 with the lanes' kinds kept as wave-wide masks in scalar registers (a body runs under its mask; three compares re-sort the lanes)",
                                      "as 4; a triangle pass also serves the lanes at inner nodes (both records by the same loads: one trip to memory for the two bodies)"};
int ptk_extend_num_variants() { return 10; }
const char* ptk_extend_variant_name(int v) { return (v >= 0 && v < 10) ? kVariantNames[v] : "?"; }
void ptk_extend_queue(const PathState& ps, const RenderConst& rc, const SceneView& sc, StepHeader* hdr, const int32_t* cam_q,
                      const int32_t* mat_q, int32_t* spill, int count, int variant, unsigned long long* wave_log, float* image, uint8_t* code,
                      unsigned long long* rec_next, int grid_rays, hipStream_t s) {
    if (sc.wide && variant == 2)                                                                                                                                         // tolerance mode, per-lane loop (A/B)
        launch_queue<16, 64, 2, true>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (sc.wide && rc.check_period > 0)                                                                                                                             // tolerance mode, kind-scheduled, sampled exact re-walk
        launch_queue<PT_WIDE_LDS_STACK, 64, 2, true, 0, false, true, 0, true>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (sc.wide)      launch_queue<PT_WIDE_LDS_STACK, 64, 2, true, 0, false, true>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (variant == 0) launch_queue<24, 256, 1>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (variant == 1) launch_queue<16, 64, 1>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (variant >= 3 && 12 * (int64_t)sc.n_objects + 7 * (int64_t)sc.n_tris <= PT_SCENE_LDS_QUADS) {
        if (sc.stack_bound <= PT_STAGED_LDS_STACK) {    // no ray's stack can leave LDS: no spill path, no overflow guard
            // (the fused pass of variant 9 in this kernel -- trips to LDS, not to memory -- measured -2 % on the Cornell box: profiles/r03_extend_ab_fused_pass.txt)
            if (variant >= 4) launch_queue<PT_STAGED_LDS_STACK, 256, 2, false, PT_SCENE_LDS_QUADS, true, true>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
            else launch_queue<PT_STAGED_LDS_STACK, 256, 2, false, PT_SCENE_LDS_QUADS, true>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
        } else {
            if (variant >= 4) launch_queue<PT_STAGED_LDS_STACK, 256, 2, false, PT_SCENE_LDS_QUADS, false, true>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
            else launch_queue<PT_STAGED_LDS_STACK, 256, 2, false, PT_SCENE_LDS_QUADS>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
        }
    }
    else if (variant == 6 && sc.top_nodes > 0 && sc.objects_top && 12 * (int64_t)sc.n_objects + 4 * (int64_t)sc.top_nodes <= PT_SCENE_LDS_QUADS)
        launch_queue<PT_STAGED_LDS_STACK, 256, 2, false, PT_SCENE_LDS_QUADS, false, true, 1>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (variant == 9 && sc.stack_bound <= PT_STACK_SIZE)
        launch_queue<16, 64, 2, false, 0, false, true, 4, false, false>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (variant == 9) launch_queue<16, 64, 2, false, 0, false, true, 4>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (variant == 8 && sc.stack_bound <= PT_STACK_SIZE)
        launch_queue<16, 64, 2, false, 0, false, true, 3, false, false>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (variant == 8) launch_queue<16, 64, 2, false, 0, false, true, 3>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (variant == 7 && sc.soa)                                                                      // as 4, the node quads from four arrays
        launch_queue<16, 64, 2, false, 0, false, true, 2>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (variant == 7) launch_queue<16, 64, 2, false, 0, false, true>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (variant >= 5) launch_queue<PT_STAGED_LDS_STACK, 256, 2, false, 0, false, true>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (variant == 4 && sc.stack_bound <= PT_STACK_SIZE)     // proper trees and a bound within the stack: no guards in the walk
        launch_queue<16, 64, 2, false, 0, false, true, 0, false, false>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else if (variant == 4) launch_queue<16, 64, 2, false, 0, false, true>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
    else                   launch_queue<16, 64, 2>(ps, rc.num_paths, grid_rays, sc, hdr, cam_q, mat_q, spill, count, wave_log, image, code, rec_next, rc, s);
}
int64_t ptk_extend_waves(int num_paths, int variant) {
    int block = (variant == 0 || (variant >= 3 && variant != 7)) ? 256 : 64;   // (variant 5 is always 256)   // variants 3 and 4 fall back to 64 when the scene does not fit: an upper bound then
    return (int64_t)((num_paths + block - 1) / block) * (block / 64);
}
void ptk_extend_direct(const SceneView& sc, const float* o, const float* d, int stride, int n, int32_t* hit_obj, int32_t* hit_tri,
                       float* hit_dist, int4* counts4, int32_t* spill, StepHeader* hdr, hipStream_t s) {
    const dim3 grid((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)), block(PT_BLOCK);
    const int64_t spill_stride = pad_to(n, 256);
#define PT_DIRECT(W, S) hipLaunchKernelGGL((k_extend_direct<W, S>), grid, block, 0, s, sc, o, d, n, hit_obj, hit_tri, hit_dist, counts4, spill, spill_stride, hdr)
    if (sc.wide) { if (stride == 4) PT_DIRECT(true, 4); else PT_DIRECT(true, 3); }
    else         { if (stride == 4) PT_DIRECT(false, 4); else PT_DIRECT(false, 3); }
#undef PT_DIRECT
}
