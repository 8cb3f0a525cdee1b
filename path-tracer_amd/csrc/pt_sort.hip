// pt_sort.hip -- the LBVH's key sort on gfx950 (radix-sort.js:162-500, WGSL :502-895).
//
// The reference sorts the 30-bit Morton keys with a 16-pass, 2-bit LSD radix sort that carries an index permutation (64
// dispatches, a host sync per pass).  Its one observable effect is a STABLE sort by key with the original index as the value,
// and that is what this file produces, in 4 passes of 8 bits over the full 32-bit key:
//   k_sort_hist     per tile of 4096 keys: 256-bin digit histogram (LDS atomics)                     -> hist[digit][tile]
//   k_sort_scan     one workgroup per digit: exclusive scan of that digit's row over the tiles        -> hist in place, totals[digit]
//   k_sort_scatter  per tile: every wave ranks its 1024 consecutive keys 64 at a time -- the lanes that hold the same digit are
//                   found with 8 wave64 ballots, a lane's rank among them is a popcount below its lane, a per-wave running
//                   counter in LDS carries the count from one group of 64 to the next -- then digit bases (totals scanned in LDS)
//                   + the tile's row entry + the counts of the waves in front give every key its place.  Order inside a tile is
//                   wave, group, lane = key order, so the pass is stable.
// Pass 0 makes the values on the fly (value = position: the reference's init_idx, radix-sort.js:526-534), so no iota buffer
// exists; the passes go input -> tmp -> out -> tmp -> out, and the caller's input keys are left alone.
// All of it is HBM-streaming integer work: no MFMA, LDS only for the counters.
#include "pt_internal.h"

#define SORT_BLOCK 256
#define SORT_WAVES (SORT_BLOCK / 64)
#define SORT_ITEMS 16                               // keys per thread
#define SORT_TILE (SORT_BLOCK * SORT_ITEMS)         // 4096 keys per workgroup
#define SORT_RADIX 256

__device__ __forceinline__ int sort_lane() { return threadIdx.x & 63; }

__global__ __launch_bounds__(SORT_BLOCK) void k_sort_hist(const uint32_t* __restrict__ keys, int64_t n, int shift, uint32_t* __restrict__ hist, int ntiles) {
    __shared__ uint32_t h[SORT_RADIX];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; i++) {
        int64_t idx = base + i * SORT_BLOCK + threadIdx.x;
        if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & (SORT_RADIX - 1)], 1u);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of v over the 256 threads of the workgroup; *total gets the sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave /* SORT_WAVES */, uint32_t* total) {
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t o = __shfl_up(inc, off); if (sort_lane() >= off) inc += o; }
    const int w = threadIdx.x >> 6;
    __syncthreads();                                 // s_wave may still be read by the previous call
    if (sort_lane() == 63) s_wave[w] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < SORT_WAVES; k++) { uint32_t t = s_wave[k]; if (k < w) before += t; all += t; }
    *total = all;
    return before + inc - v;
}

__global__ __launch_bounds__(SORT_BLOCK) void k_sort_scan(uint32_t* __restrict__ hist, int ntiles, uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_wave[SORT_WAVES];
    uint32_t* row = hist + (int64_t)blockIdx.x * ntiles;
    uint32_t running = 0;
    for (int c = 0; c < ntiles; c += SORT_BLOCK) {
        int j = c + threadIdx.x;
        uint32_t v = j < ntiles ? row[j] : 0u, chunk;
        uint32_t ex = block_exclusive_scan(v, s_wave, &chunk);
        if (j < ntiles) row[j] = running + ex;
        running += chunk;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = running;
}

template <bool IOTA>
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_scatter(const uint32_t* __restrict__ kin, const int32_t* __restrict__ vin, uint32_t* __restrict__ kout,
                                                             int32_t* __restrict__ vout, int64_t n, int shift, const uint32_t* __restrict__ hist,
                                                             const uint32_t* __restrict__ totals, int ntiles) {
    __shared__ volatile uint32_t cnt[SORT_WAVES][SORT_RADIX];      // a wave's running digit counts, then its digits' first output places
    __shared__ uint32_t s_wave[SORT_WAVES];
    const int w = threadIdx.x >> 6, l = sort_lane();
#pragma unroll
    for (int k = 0; k < SORT_WAVES; k++) cnt[k][threadIdx.x] = 0u;
    __syncthreads();
    uint32_t key[SORT_ITEMS]; int32_t val[SORT_ITEMS]; uint32_t rank[SORT_ITEMS];
    const int64_t wave_base = (int64_t)blockIdx.x * SORT_TILE + (int64_t)w * (64 * SORT_ITEMS);
    const unsigned long long below = (1ull << l) - 1ull;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        const int64_t idx = wave_base + r * 64 + l;
        const bool valid = idx < n;
        key[r] = valid ? kin[idx] : 0u;
        val[r] = valid ? (IOTA ? (int32_t)idx : vin[idx]) : 0;
        const uint32_t d = (key[r] >> shift) & (SORT_RADIX - 1);
        unsigned long long same = __ballot(valid);                  // lanes that hold my digit
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            same &= bit ? m : ~m;
        }
        const uint32_t prev = cnt[w][d];
        rank[r] = prev + (uint32_t)__popcll(same & below);
        if (valid && (same & below) == 0ull) cnt[w][d] = prev + (uint32_t)__popcll(same);      // one lane per digit
    }
    __syncthreads();
    {   // thread = digit: first output place of the digit's keys for each wave of this tile
        const int d = threadIdx.x;
        uint32_t sum;
        const uint32_t digit_base = block_exclusive_scan(totals[d], s_wave, &sum);
        uint32_t place = digit_base + hist[(int64_t)d * ntiles + blockIdx.x];
#pragma unroll
        for (int k = 0; k < SORT_WAVES; k++) { uint32_t c = cnt[k][d]; cnt[k][d] = place; place += c; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; r++) {
        const int64_t idx = wave_base + r * 64 + l;
        if (idx < n) {
            const uint32_t pos = cnt[w][(key[r] >> shift) & (SORT_RADIX - 1)] + rank[r];
            kout[pos] = key[r]; vout[pos] = val[r];
        }
    }
}

static size_t sort_align(size_t b) { return (b + 255) / 256 * 256; }

// Stable sort of n keys; vout[i] = original position of the key that lands at i.  kin is not modified.
// tmp == null: only *tmp_bytes is set.  n <= 2^30 (checked by the callers: positions are int32).
hipError_t pt_sort_keys_index(void* tmp, size_t* tmp_bytes, const uint32_t* kin, uint32_t* kout, int32_t* vout, size_t n, hipStream_t s) {
    const int ntiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
    const size_t key_b = sort_align(n * 4), hist_b = sort_align((size_t)SORT_RADIX * (size_t)(ntiles > 0 ? ntiles : 1) * 4), tot_b = sort_align(SORT_RADIX * 4);
    if (!tmp) { *tmp_bytes = 2 * key_b + hist_b + tot_b; return hipSuccess; }
    if (n == 0) return hipSuccess;
    uint32_t* tk = (uint32_t*)tmp; int32_t* tv = (int32_t*)((char*)tmp + key_b);
    uint32_t* hist = (uint32_t*)((char*)tmp + 2 * key_b); uint32_t* totals = (uint32_t*)((char*)tmp + 2 * key_b + hist_b);
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 8 * pass;
        const uint32_t* ki = pass == 0 ? kin : (pass & 1) ? tk : kout;
        const int32_t* vi = pass == 0 ? nullptr : (pass & 1) ? tv : vout;
        uint32_t* ko = (pass & 1) ? kout : tk; int32_t* vo = (pass & 1) ? vout : tv;
        hipLaunchKernelGGL(k_sort_hist, dim3(ntiles), dim3(SORT_BLOCK), 0, s, ki, (int64_t)n, shift, hist, ntiles);
        hipLaunchKernelGGL(k_sort_scan, dim3(SORT_RADIX), dim3(SORT_BLOCK), 0, s, hist, ntiles, totals);
        if (pass == 0) hipLaunchKernelGGL(k_sort_scatter<true>, dim3(ntiles), dim3(SORT_BLOCK), 0, s, ki, vi, ko, vo, (int64_t)n, shift, hist, totals, ntiles);
        else hipLaunchKernelGGL(k_sort_scatter<false>, dim3(ntiles), dim3(SORT_BLOCK), 0, s, ki, vi, ko, vo, (int64_t)n, shift, hist, totals, ntiles);
    }
    return hipGetLastError();
}
