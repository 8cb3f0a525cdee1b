// pt_kernels_build.hip -- LBVH construction on gfx950.
//
// Pipeline per mesh (and once more over object boxes for the TLAS):
//   boxes + 30-bit Morton keys -> stable key/index radix sort (pt_sort.hip) -> Karras radix
//   tree (parent links carry the child's side) -> bottom-up refit -> gather triangles -> pack.
// All of it is HBM-streaming integer/f32 work; no LDS tiling or MFMA applies.  The refit is the
// only inter-workgroup hand-off: device-scope atomics on both sides of one arrival counter.
#include "pt_internal.h"
#include "pt_math.h"

static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + PT_BLOCK - 1) / PT_BLOCK)); }

struct Bounds { float mn[3], mx[3]; };

// aabb-z-idx.js:202-209 split_3 / :197-199 morton_code
__device__ __forceinline__ uint32_t split_3(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
// aabb-z-idx.js:191-194: centroid relative to the bounds, * 1023.99, truncated
__device__ __forceinline__ uint32_t morton_of_box(f3 mn, f3 mx, const Bounds& b) {
    f3 cen = (mx + mn) * 0.5f;
    f3 num = cen - F3(b.mn[0], b.mn[1], b.mn[2]);
    f3 den = F3(b.mx[0], b.mx[1], b.mx[2]) - F3(b.mn[0], b.mn[1], b.mn[2]);
    f3 rel = F3(num.x / den.x, num.y / den.y, num.z / den.z);
    uint32_t ux = f32_to_u32(rel.x * 1023.99f), uy = f32_to_u32(rel.y * 1023.99f), uz = f32_to_u32(rel.z * 1023.99f);
    return split_3(ux) | (split_3(uy) << 1) | (split_3(uz) << 2);
}

// aabb-z-idx.js:176-195
__global__ __launch_bounds__(PT_BLOCK) void k_tri_aabb_morton(const float4* __restrict__ tris, int64_t n, Bounds b,
                                                              float4* __restrict__ aabb, uint32_t* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    f3 v0 = F3(tris[3 * i]), v1 = F3(tris[3 * i + 1]), v2 = F3(tris[3 * i + 2]);
    f3 mn = min3(v0, min3(v1, v2)), mx = max3(v0, max3(v1, v2));
    aabb[2 * i]     = make_float4(mn.x, mn.y, mn.z, 0.0f);
    aabb[2 * i + 1] = make_float4(mx.x, mx.y, mx.z, 0.0f);
    keys[i]         = morton_of_box(mn, mx, b);
}
// aabb2z-idx.js:157-170
__global__ __launch_bounds__(PT_BLOCK) void k_obj_morton(const float4* __restrict__ aabb, int64_t n, Bounds b,
                                                         uint32_t* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    keys[i] = morton_of_box(F3(aabb[2 * i]), F3(aabb[2 * i + 1]), b);
}
// radix-tree.js:160-185: common-prefix length, equal keys fall back to the index bits
__device__ __forceinline__ int del_(const uint32_t* __restrict__ keys, int n, int i1, int i2) {
    if (i2 < 0 || i2 >= n) return -1;
    uint32_t x = keys[i1] ^ keys[i2];
    if (x) return __builtin_clz(x);
    uint32_t y = (uint32_t)i1 ^ (uint32_t)i2;
    return y ? 32 + __builtin_clz(y) : 31;
}
// radix-tree.js:136-157 + :188-238.  One thread per inner node i.  Instead of a bare parent index the
// link stores (parent << 1 | side) so that the refit can place each child in its Karras slot
// (left = lower key range) without depending on arrival order.
// first (optional): the first leaf of node i's key range, for the traversal-order numbering (k_dfs_index).
__global__ __launch_bounds__(PT_BLOCK) void k_radix_tree(const uint32_t* __restrict__ keys, int n, int32_t* __restrict__ parents, int32_t* __restrict__ first) {
    int i = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n - 1) return;
    int a = del_(keys, n, i, i + 1) - del_(keys, n, i, i - 1);
    int d = (a > 0) - (a < 0);
    int del_min = del_(keys, n, i, i - d);
    int l_max = 2;
    // i + l_max*d stays inside int: n <= 2^30 is checked by the host
    while (del_(keys, n, i, i + l_max * d) > del_min) l_max *= 2;
    int l = 0;
    for (int t = l_max / 2; t > 0; t /= 2)
        if (del_(keys, n, i, i + (l + t) * d) > del_min) l += t;
    int j = i + l * d;
    int del_node = del_(keys, n, i, j);
    // largest s in [0, l) with del(i, i + s*d) > del_node: Karras' t = ceil(l/2), ceil(l/4), ..., 1
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (del_(keys, n, i, i + (s + t) * d) > del_node) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    if (first) first[i] = lo;
    int left_slot  = (lo == gamma) ? n + gamma : gamma;           // leaf links live at parents[n + leaf]
    int right_slot = (hi == gamma + 1) ? n + gamma + 1 : gamma + 1;
    parents[left_slot]  = (i << 1);
    parents[right_slot] = (i << 1) | 1;
}

// bvh-up-pass.js:199-280.  One thread per leaf climbs the tree; the second child to arrive at a node merges both boxes and
// continues.  Differences from the reference kernel, none visible in the output of its canonical (serial) execution: the child's
// slot is its Karras side, not its arrival rank, and there is no "is the other half still zero" spin (bvh-up-pass.js:266-268).
// The hand-off between the two arrivers crosses CUs and XCDs, whose L1s and L2s are not coherent with each other.  It is made with
// device-scope atomics on BOTH sides: a child's 32-byte slot is written by four atomic exchanges (3 x 8 B + 4 B; the fourth word of
// the left slot's last pair is the arrival counter f_1, bvh-up-pass.js:230, and is never part of an exchange), the wave waits for
// them (vmcnt), then bumps the counter; the second arriver reads the sibling's slot by atomic adds of a run-time zero.  Atomics are
// performed at the memory side, so no cache can serve an old value, and nothing is read through a cache inside the launch.
// (The first version did it with plain stores between two __threadfence(): an L2 write-back and an L1 invalidate per level and
// thread -- 71.6 ms of the 75 ms the 10 M-triangle build took, profiles/r02_build_trace_grid_before.txt.)
__device__ __forceinline__ unsigned long long pack2(float a, float b) {
    return (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
}
__global__ __launch_bounds__(PT_BLOCK) void k_up_pass(const int32_t* __restrict__ idx, const float4* __restrict__ aabb,
                                                      const int32_t* __restrict__ parents, int n, float4* nodes, unsigned long long zero) {
    int t = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (t >= n) return;
    int src = idx[t];
    f3  mn = F3(aabb[2 * (int64_t)src]), mx = F3(aabb[2 * (int64_t)src + 1]);
    // :209-210 exact zeros are nudged because 0 doubles as "not written yet" in the reference
    mn.x -= (mn.x == 0.0f ? 1.0f : 0.0f) * 1e-8f; mn.y -= (mn.y == 0.0f ? 1.0f : 0.0f) * 1e-8f;
    mn.z -= (mn.z == 0.0f ? 1.0f : 0.0f) * 1e-8f;
    mx.x += (mx.x == 0.0f ? 1.0f : 0.0f) * 1e-8f; mx.y += (mx.y == 0.0f ? 1.0f : 0.0f) * 1e-8f;
    mx.z += (mx.z == 0.0f ? 1.0f : 0.0f) * 1e-8f;
    int w_idx = -(t + 1);
    int link  = parents[(int64_t)n + t];
    for (int level = 0; level < 64; level++) {   // tree depth <= 62 (32 key bits + 30 index bits)
        int p = link >> 1, side = link & 1;
        unsigned long long* nd = (unsigned long long*)(nodes + 4 * (int64_t)p);
        unsigned long long* slot = nd + 4 * side;
        atomicExch(slot + 0, pack2(mn.x, mn.y));
        atomicExch(slot + 1, pack2(mn.z, __int_as_float(w_idx)));
        atomicExch(slot + 2, pack2(mx.x, mx.y));
        atomicExch((unsigned*)(slot + 3), __float_as_uint(mx.z));
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "k_up_pass: the hand-over below waits for its atomics with vmcnt, which counts no-return atomics on gfx9-family targets only (gfx10+ count them in vscnt); build for a gfx9 target (gfx950) or use acq_rel atomics here"
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my slot is in memory before my arrival counts
        int arrived = atomicAdd((int*)nd + 7, 1);          // f_1 doubles as the arrival counter (:230)
        if (arrived == 0) return;                          // first arrival stops (:242)
        const unsigned long long* sib = nd + 4 * (side ^ 1);
        unsigned long long s0 = atomicAdd((unsigned long long*)sib + 0, zero), s1 = atomicAdd((unsigned long long*)sib + 1, zero),
                           s2 = atomicAdd((unsigned long long*)sib + 2, zero);
        unsigned s3 = atomicAdd((unsigned*)(sib + 3), (unsigned)zero);
        f3 smn = F3(__uint_as_float((unsigned)s0), __uint_as_float((unsigned)(s0 >> 32)), __uint_as_float((unsigned)s1));
        f3 smx = F3(__uint_as_float((unsigned)s2), __uint_as_float((unsigned)(s2 >> 32)), __uint_as_float(s3));
        mn = min3(mn, smn); mx = max3(mx, smx);            // :270-271
        if (p == 0) return;                                // merged into the root (:218 w_idx == 0)
        w_idx = p;
        link  = parents[p];
    }
}

// rearrange.js:162-169
__global__ __launch_bounds__(PT_BLOCK) void k_rearrange(const float4* __restrict__ in, const int32_t* __restrict__ idx, int64_t n,
                                                        float4* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    int64_t s = idx[i];
    out[3 * i] = in[3 * s]; out[3 * i + 1] = in[3 * s + 1]; out[3 * i + 2] = in[3 * s + 2];
}

// pack-t-mesh.js:218-245: copy one triangle and one node per thread, rebasing child pointers
__global__ __launch_bounds__(PT_BLOCK) void k_pack_mesh(const float4* __restrict__ src_nodes, const float4* __restrict__ src_tris,
                                                        int64_t n, int32_t offset, float4* __restrict__ dst_nodes,
                                                        float4* __restrict__ dst_tris) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    int64_t o = (int64_t)offset + i;
    dst_tris[3 * o] = src_tris[3 * i]; dst_tris[3 * o + 1] = src_tris[3 * i + 1]; dst_tris[3 * o + 2] = src_tris[3 * i + 2];
    float4 q0 = src_nodes[4 * i], q1 = src_nodes[4 * i + 1], q2 = src_nodes[4 * i + 2], q3 = src_nodes[4 * i + 3];
    int lc = __float_as_int(q0.w), rc = __float_as_int(q2.w);
    lc = lc < 0 ? lc - offset : lc + offset;
    rc = rc < 0 ? rc - offset : rc + offset;
    q0.w = __int_as_float(lc); q2.w = __int_as_float(rc);
    dst_nodes[4 * o] = q0; dst_nodes[4 * o + 1] = q1; dst_nodes[4 * o + 2] = q2; dst_nodes[4 * o + 3] = q3;
}

// builders.js:25-67 packObjects, on the device (the reference reads the TLAS permutation back
// and packs on the host).  instances: pt_instance[n] = {l2w[16], w2l[16], aabb[6], mesh, material}, 40 words.
__global__ __launch_bounds__(PT_BLOCK) void k_pack_objects(const float* __restrict__ inst, const int32_t* __restrict__ perm,
                                                           const int32_t* __restrict__ mesh_tri_offset, int n, float4* __restrict__ objects) {
    int i = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float* in = inst + 40 * (int64_t)perm[i];
    const float* l2w = in; const float* w2l = in + 16;
    int mesh = ((const int*)in)[38], material = ((const int*)in)[39];
    float4* ob = objects + 8 * (int64_t)i;
    for (int r = 0; r < 3; r++) {   // row r of a column-major 4x4: m[r], m[4+r], m[8+r], m[12+r] (builders.js:33-58)
        ob[r]     = make_float4(l2w[r], l2w[4 + r], l2w[8 + r], l2w[12 + r]);
        ob[4 + r] = make_float4(w2l[r], w2l[4 + r], w2l[8 + r], w2l[12 + r]);
    }
    ob[3] = make_float4(__int_as_float(mesh_tri_offset[mesh]), __int_as_float(material), 0.0f, 0.0f);   // builders.js:64-65
    ob[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ---- launchers -------------------------------------------------------------------------------
static Bounds mk_bounds(const float b[6]) { Bounds r; for (int i = 0; i < 3; i++) { r.mn[i] = b[i]; r.mx[i] = b[3 + i]; } return r; }

void ptk_tri_aabb_morton(const float4* tris, int64_t n, const float bounds[6], float4* aabb, uint32_t* keys, hipStream_t s) {
    hipLaunchKernelGGL(k_tri_aabb_morton, grid_for(n), dim3(PT_BLOCK), 0, s, tris, n, mk_bounds(bounds), aabb, keys);
}
void ptk_obj_morton(const float4* aabb, int64_t n, const float bounds[6], uint32_t* keys, hipStream_t s) {
    hipLaunchKernelGGL(k_obj_morton, grid_for(n), dim3(PT_BLOCK), 0, s, aabb, n, mk_bounds(bounds), keys);
}
void ptk_radix_tree(const uint32_t* keys, int64_t n, int32_t* parents, int32_t* first, hipStream_t s) {
    if (n < 2) return;
    hipLaunchKernelGGL(k_radix_tree, grid_for(n - 1), dim3(PT_BLOCK), 0, s, keys, (int)n, parents, first);
}
// ---- traversal-order copy of a BLAS (no reference counterpart; never visible in a result) ----
// The reference numbers an inner node by its position in the sorted keys (radix-tree.js: node i sits at an end of its range), so a
// parent and its children are as far apart in memory as the range is wide.  The extend kernel reads a second copy of the packed
// BLAS in which the inner nodes are numbered in depth-first preorder -- a node's left child is the next record, a subtree is one
// contiguous block -- with the child pointers rewritten to match.  Same records, same walk, same hit indices (leaves keep their
// triangle numbers); pt_get_blas_nodes keeps returning the reference's order.
// preorder(i) = #ancestors + #inner nodes wholly to the left of i's range = first_leaf(i) + #ancestors in whose LEFT subtree i lies
// (the leaves left of the range, first_leaf(i) of them, hang in the left-sibling subtrees of the other ancestors; a subtree of k
// leaves has k - 1 inner nodes).
// top_nodes (PT_TOP_SLOTS ints, preset to -1): the inner nodes of the first PT_TOP_LEVELS levels by heap slot -- root = 0, the
// children of slot s = 2s + 1 (left), 2s + 2 (right) -- for the tree-top block of the traversal copy (k_top_block).
__global__ __launch_bounds__(PT_BLOCK) void k_dfs_index(const int32_t* __restrict__ parents, const int32_t* __restrict__ first, int n, int32_t* __restrict__ dfs,
                                                        int32_t* __restrict__ top_nodes, uint8_t* __restrict__ depth_out) {
    int i = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (i == 0 && top_nodes) top_nodes[0] = 0;              // the root (also of the one-leaf tree)
    if (i >= n - 1) { dfs[i] = i; if (depth_out) depth_out[i] = 0; return; }   // the unused tail record of an n-leaf tree (and the whole one-leaf tree) stays put
    int cnt = 0, depth = 0; unsigned path = 0u;             // path: bit j = the side taken j links above i (1 = right)
    for (int p = i, guard = 0; p != 0 && guard < 64; guard++) {
        int link = parents[p]; cnt += (link & 1) == 0;
        if (depth < 31) path |= (unsigned)(link & 1) << depth;
        depth++; p = link >> 1;
    }
    dfs[i] = first[i] + cnt;
    if (depth_out) depth_out[i] = (uint8_t)depth;            // <= 62 links (32 key bits + 30 index bits)
    if (top_nodes) atomicMax(&top_nodes[PT_TOP_SLOTS], depth + 1);   // the spare cell behind the slots: inner nodes on the longest path from the root (preset to -1)
    if (top_nodes && depth > 0 && depth < PT_TOP_LEVELS) top_nodes[(1 << depth) - 1 + (int)path] = i;
}
// The tree-top block of one packed mesh: the nodes of its first `levels` levels, by heap slot, at node ids top_id .. of `trav`
// (behind the depth-first copy).  Children inside the block are linked by slot, the others by their depth-first ids, so a walk that
// starts at top_id runs through the block and drops into the depth-first copy below it: the same records in the same order.
// An extend launch stages the block in LDS (pt_kernels_extend.hip, TOP).  Unused slots are zero.
__global__ void k_top_block(const float4* __restrict__ packed, const int32_t* __restrict__ dfs, const int32_t* __restrict__ top_nodes, int64_t offset, int n,
                            int levels, int top_id, float4* __restrict__ trav) {
    const int slots = (1 << levels) - 1, s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slots) return;
    float4* dst = trav + 4 * (int64_t)(top_id + s);
    const int i = top_nodes[s];
    if (i < 0) { dst[0] = dst[1] = dst[2] = dst[3] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const float4* src = packed + 4 * (offset + i);
    float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
    int lc = __float_as_int(q0.w), rc = __float_as_int(q2.w);
    auto moved = [&](int child, int slot) -> int {
        if (child < 0) return child;                                         // a triangle
        const int j = child - (int)offset;
        if (slot < slots && top_nodes[slot] == j) return top_id + slot;      // inside the block
        if (j == i) return top_id + s;                                       // the one-leaf tree's self pointer (pack-t-mesh.js keeps r = 0)
        return (int)offset + dfs[j];
    };
    q0.w = __int_as_float(moved(lc, 2 * s + 1)); q2.w = __int_as_float(moved(rc, 2 * s + 2));
    dst[0] = q0; dst[1] = q1; dst[2] = q2; dst[3] = q3;
}
// the TLAS as part of the walk's one node array: pt_bvh_node[n] with the inner links moved up by `delta` (leaf links, < 0, name objects)
__global__ void k_tlas_rebase(const float4* __restrict__ tlas, int n, int delta, float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 q0 = tlas[4 * i], q1 = tlas[4 * i + 1], q2 = tlas[4 * i + 2], q3 = tlas[4 * i + 3];
    int lc = __float_as_int(q0.w), rc = __float_as_int(q2.w);
    if (lc >= 0) q0.w = __int_as_float(lc + delta);
    if (rc >= 0) q2.w = __int_as_float(rc + delta);
    out[4 * i] = q0; out[4 * i + 1] = q1; out[4 * i + 2] = q2; out[4 * i + 3] = q3;
}
// objects for a walk that starts in the tree-top blocks: pt_object[n] with the root node id (word 12) replaced
__global__ void k_objects_top(const float4* __restrict__ objects, int n, const int32_t* __restrict__ mesh_root, const int32_t* __restrict__ top_root, int n_meshes,
                              float4* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 8 * n) return;
    float4 q = objects[i];
    if ((i & 7) == 3) {
        const int root = __float_as_int(q.x);
        for (int k = 0; k < n_meshes; k++) if (mesh_root[k] == root) q.x = __int_as_float(top_root[k]);
    }
    out[i] = q;
}
// one packed mesh: nodes [offset, offset + n) of `packed` (child pointers already rebased by pack-t-mesh.js:218-245) -> `trav`
__global__ __launch_bounds__(PT_BLOCK) void k_reorder_nodes(const float4* __restrict__ packed, const int32_t* __restrict__ dfs, int64_t offset, int n, float4* __restrict__ trav) {
    int i = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4* src = packed + 4 * (offset + i);
    float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
    int lc = __float_as_int(q0.w), rc = __float_as_int(q2.w);
    if (i < n - 1 || n == 1) {                              // a real node: its inner children move
        if (lc >= 0) lc = (int)offset + dfs[lc - (int)offset];
        if (rc >= 0) rc = (int)offset + dfs[rc - (int)offset];
    }
    q0.w = __int_as_float(lc); q2.w = __int_as_float(rc);
    float4* dst = trav + 4 * (offset + dfs[i]);
    dst[0] = q0; dst[1] = q1; dst[2] = q2; dst[3] = q3;
}
// extend variant 7 (A/B): the walk's node array as four arrays of quads
__global__ __launch_bounds__(PT_BLOCK) void k_soa_split(const float4* __restrict__ nodes, int64_t n, float4* __restrict__ soa) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 4; k++) soa[k * n + i] = nodes[4 * i + k];
}
void ptk_soa_split(const float4* nodes, int64_t n, float4* soa, hipStream_t s) { hipLaunchKernelGGL(k_soa_split, grid_for(n), dim3(PT_BLOCK), 0, s, nodes, n, soa); }
void ptk_dfs_index(const int32_t* parents, const int32_t* first, int64_t n, int32_t* dfs, int32_t* top_nodes, uint8_t* depth, hipStream_t s) {
    hipLaunchKernelGGL(k_dfs_index, grid_for(n), dim3(PT_BLOCK), 0, s, parents, first, (int)n, dfs, top_nodes, depth);
}
void ptk_top_block(const float4* packed, const int32_t* dfs, const int32_t* top_nodes, int64_t offset, int64_t n, int levels, int top_id, float4* trav, hipStream_t s) {
    hipLaunchKernelGGL(k_top_block, dim3(1), dim3(PT_TOP_SLOTS + 1), 0, s, packed, dfs, top_nodes, offset, (int)n, levels, top_id, trav);
}
void ptk_tlas_rebase(const float4* tlas, int n, int delta, float4* out, hipStream_t s) {
    hipLaunchKernelGGL(k_tlas_rebase, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tlas, n, delta, out);
}
void ptk_objects_top(const float4* objects, int n, const int32_t* mesh_root, const int32_t* top_root, int n_meshes, float4* out, hipStream_t s) {
    hipLaunchKernelGGL(k_objects_top, dim3((unsigned)((8 * (int64_t)n + 255) / 256)), dim3(256), 0, s, objects, n, mesh_root, top_root, n_meshes, out);
}
void ptk_reorder_nodes(const float4* packed, const int32_t* dfs, int64_t offset, int64_t n, float4* trav, hipStream_t s) {
    hipLaunchKernelGGL(k_reorder_nodes, grid_for(n), dim3(PT_BLOCK), 0, s, packed, dfs, offset, (int)n, trav);
}
void ptk_up_pass(const int32_t* idx, const float4* aabb, const int32_t* parents, int64_t n, float4* nodes, hipStream_t s) {
    hipLaunchKernelGGL(k_up_pass, grid_for(n), dim3(PT_BLOCK), 0, s, idx, aabb, parents, (int)n, nodes, 0ull);   // the zero the kernel adds to read a slot atomically
}
void ptk_rearrange(const float4* in, const int32_t* idx, int64_t n, float4* out, hipStream_t s) {
    hipLaunchKernelGGL(k_rearrange, grid_for(n), dim3(PT_BLOCK), 0, s, in, idx, n, out);
}
// SceneView::tris: edges and unit normal of every packed triangle, by the expressions the extend and material kernels
// used to evaluate per ray (same translation-unit flags: no contraction, correctly rounded divide and sqrt).
__global__ __launch_bounds__(PT_BLOCK) void k_tri_records(const float4* __restrict__ tris, int64_t n, float4* __restrict__ rec) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    f3 v0 = F3(tris[3 * i]), v0v1 = F3(tris[3 * i + 1]) - v0, v0v2 = F3(tris[3 * i + 2]) - v0;
    f3 nrm = normalize3(cross3(v0v1, v0v2));
    rec[3 * i] = make_float4(v0.x, v0.y, v0.z, nrm.x); rec[3 * i + 1] = make_float4(v0v1.x, v0v1.y, v0v1.z, nrm.y);
    rec[3 * i + 2] = make_float4(v0v2.x, v0v2.y, v0v2.z, nrm.z);
}
void ptk_tri_records(const float4* tris, int64_t n, float4* records, hipStream_t s) {
    hipLaunchKernelGGL(k_tri_records, grid_for(n), dim3(PT_BLOCK), 0, s, tris, n, records);
}
void ptk_pack_mesh(const float4* src_nodes, const float4* src_tris, int64_t n, int32_t offset, float4* dst_nodes, float4* dst_tris, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_mesh, grid_for(n), dim3(PT_BLOCK), 0, s, src_nodes, src_tris, n, offset, dst_nodes, dst_tris);
}
// ---- tolerance mode: 4-wide collapse of one packed mesh (binary nodes [offset, offset + n), inner nodes offset .. offset + n - 2) ----
// The binary nodes at even depths become the wide nodes.  They are numbered in depth-first order of the wide tree, which is the
// binary tree's preorder with the odd levels left out: flags[dfs[i]] = 1 for an even-depth node i, rank = exclusive prefix sum.
// A wide subtree is one contiguous block and a node's first inner child is the next record (two 64-B nodes share a 128-B line).
__global__ __launch_bounds__(PT_BLOCK) void k_wide_flags(const int32_t* __restrict__ dfs, const uint8_t* __restrict__ depth, int n, int32_t* __restrict__ flags) {
    int i = blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const bool real = i < n - 1 || n == 1;                  // node n-1 is the unused tail slot of an n-leaf radix tree
    flags[dfs[i]] = (real && (depth[i] & 1) == 0) ? 1 : 0;  // dfs is a permutation of [0, n): every flag is written
}
// exclusive scan of int32, 1024 per tile: tile sums -> their scan by one workgroup -> added back
__global__ __launch_bounds__(256) void k_scan_tiles(int32_t* __restrict__ data, int n, int32_t* __restrict__ tile_sums) {
    __shared__ int s_wave[4];
    const int base = blockIdx.x * 1024 + threadIdx.x * 4, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int v[4], sum = 0;
    for (int k = 0; k < 4; k++) { v[k] = base + k < n ? data[base + k] : 0; sum += v[k]; }
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int front = 0;
    for (int w = 0; w < wave; w++) front += s_wave[w];
    int run = front + incl - sum;
    for (int k = 0; k < 4; k++) { if (base + k < n) data[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255) tile_sums[blockIdx.x] = front + incl;
}
__global__ __launch_bounds__(1024) void k_scan_tops(int32_t* __restrict__ tile_sums, int ntiles, int32_t* __restrict__ total) {
    __shared__ int s_wave[16]; __shared__ int s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int start = 0; start < ntiles; start += 1024) {
        const int i = start + threadIdx.x, v = i < ntiles ? tile_sums[i] : 0;
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int front = s_carry;
        for (int w = 0; w < wave; w++) front += s_wave[w];
        if (i < ntiles) tile_sums[i] = front + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = front + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}
__global__ __launch_bounds__(256) void k_scan_add(int32_t* __restrict__ data, int n, const int32_t* __restrict__ tile_sums) {
    const int base = blockIdx.x * 1024 + threadIdx.x * 4, add = tile_sums[blockIdx.x];
    for (int k = 0; k < 4; k++) if (base + k < n) data[base + k] += add;
}
void ptk_wide_flags(const int32_t* dfs, const uint8_t* depth, int64_t n, int32_t* flags, hipStream_t s) {
    hipLaunchKernelGGL(k_wide_flags, grid_for(n), dim3(PT_BLOCK), 0, s, dfs, depth, (int)n, flags);
}
void ptk_exclusive_scan(int32_t* data, int64_t n, int32_t* tile_sums, int32_t* total, hipStream_t s) {
    const int ntiles = (int)((n + 1023) / 1024);
    hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(256), 0, s, data, (int)n, tile_sums);
    hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(1024), 0, s, tile_sums, ntiles, total);
    hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(256), 0, s, data, (int)n, tile_sums);
}
__global__ __launch_bounds__(PT_BLOCK) void k_collapse_wide(const float4* __restrict__ bin, int64_t offset, int64_t n, const int32_t* __restrict__ dfs,
                                                            const uint8_t* __restrict__ depth, const int32_t* __restrict__ rank, int32_t wide_base,
                                                            float4* __restrict__ wide) {
    int64_t i = (int64_t)blockIdx.x * PT_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (!(i < n - 1 || n == 1) || (depth[i] & 1)) return;
    float lo[4][3], hi[4][3]; int ref[4]; int k = 0;
    for (int s = 0; s < 4; s++) { ref[s] = PT_WIDE_EMPTY; for (int a = 0; a < 3; a++) lo[s][a] = hi[s][a] = 0.0f; }
    // a grandchild that is an inner node sits at an even depth again: its reference is its wide id
    auto slot = [&](int c, float4 mn, float4 mx) {
        ref[k] = c < 0 ? c : wide_base + rank[dfs[c - (int)offset]];
        lo[k][0] = mn.x; lo[k][1] = mn.y; lo[k][2] = mn.z; hi[k][0] = mx.x; hi[k][1] = mx.y; hi[k][2] = mx.z; k++;
    };
    auto child = [&](int c, float4 mn, float4 mx) {
        if (c >= 0) { const float4* cn = bin + 4 * (int64_t)c; float4 c0 = cn[0], c1 = cn[1], c2 = cn[2], c3 = cn[3];
                      slot(__float_as_int(c0.w), c0, c1); slot(__float_as_int(c2.w), c2, c3); }
        else slot(c, mn, mx);
    };
    {
        const float4* nd = bin + 4 * (offset + i);
        float4 a0 = nd[0], a1 = nd[1], a2 = nd[2], a3 = nd[3];
        child(__float_as_int(a0.w), a0, a1);
        if (n > 1) child(__float_as_int(a2.w), a2, a3);   // a one-triangle mesh keeps only its leaf (its r child is the reference's self pointer)
    }
    // quantisation frame: the union of the child boxes, padded so that a box on the frame's border still gets slack
    float fmin[3] = {3e38f, 3e38f, 3e38f}, fmax[3] = {-3e38f, -3e38f, -3e38f};
    for (int c = 0; c < k; c++) for (int a = 0; a < 3; a++) { fmin[a] = fminf(fmin[a], lo[c][a]); fmax[a] = fmaxf(fmax[a], hi[c][a]); }
    float org[3] = {0.0f, 0.0f, 0.0f}, step[3] = {1.0f, 1.0f, 1.0f}; unsigned ebits = 0;
    unsigned qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    for (int a = 0; a < 3 && k > 0; a++) {
        float ext = fmax[a] - fmin[a];
        float pad = fmaxf(fmaxf(ext * 0.002f, fmaxf(fabsf(fmin[a]), fabsf(fmax[a])) * 2e-6f), 1e-30f);
        org[a] = fmin[a] - pad;
        float need = ((fmax[a] + pad) - org[a]) / 254.0f;                  // 254: one step of head room for the outward rounding
        int e; frexpf(need, &e);                                           // need = m * 2^e, m in [0.5, 1)  ->  2^e >= need
        e = max(-126, min(126, e));
        step[a] = ldexpf(1.0f, e); ebits |= (unsigned)(e + 127) << (8 * a);
        for (int c = 0; c < k; c++) {
            int l = (int)floorf((lo[c][a] - org[a]) / step[a]) , h = (int)ceilf((hi[c][a] - org[a]) / step[a]);
            l = max(0, min(255, l)); h = max(0, min(255, h));
            while (l > 0 && fmaf((float)l, step[a], org[a]) > lo[c][a]) l--;       // decoded box must contain the real one
            while (h < 255 && fmaf((float)h, step[a], org[a]) < hi[c][a]) h++;
            qlo[a] |= (unsigned)l << (8 * c); qhi[a] |= (unsigned)h << (8 * c);
        }
    }
    float4* w = wide + PT_WIDE_QUADS * (int64_t)(wide_base + rank[dfs[i]]);
    (void)ebits;                                    // the steps are powers of two; they are stored as floats (the record has the room, a visit saves the decode)
    w[0] = make_float4(org[0], org[1], org[2], step[0]);
    w[1] = make_float4(__int_as_float(ref[0]), __int_as_float(ref[1]), __int_as_float(ref[2]), __int_as_float(ref[3]));
    w[2] = make_float4(__uint_as_float(qlo[0]), __uint_as_float(qlo[1]), __uint_as_float(qlo[2]), __uint_as_float(qhi[0]));
    w[3] = make_float4(__uint_as_float(qhi[1]), __uint_as_float(qhi[2]), step[1], step[2]);
}
void ptk_collapse_wide(const float4* blas_nodes, int64_t offset, int64_t n, const int32_t* dfs, const uint8_t* depth, const int32_t* rank,
                       int32_t wide_base, float4* wide, hipStream_t s) {
    hipLaunchKernelGGL(k_collapse_wide, grid_for(n), dim3(PT_BLOCK), 0, s, blas_nodes, offset, n, dfs, depth, rank, wide_base, wide);
}
void ptk_pack_objects(const void* instances, const int32_t* perm, const int32_t* mesh_tri_offset, int n, float4* objects, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_objects, grid_for(n), dim3(PT_BLOCK), 0, s, (const float*)instances, perm, mesh_tri_offset, n, objects);
}
