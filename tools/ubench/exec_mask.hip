// exec_mask.hip -- does a wave64 vector instruction cost less on gfx950 when only part of the wave is active?
// Eight independent chains of v_fma_f32 / v_min_f32 / v_cndmask per wave, 8 waves per SIMD, under exec masks that enable
// all 64 lanes, the low 32, the low 16, 16 lanes spread one in four, and one lane.  Prints cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long mask) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, m = 1.0000001f;
    const bool on = (mask >> (threadIdx.x & 63)) & 1ull;
    if (on) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
                if (OP == 1) asm volatile("v_min_f32 %0, %1, %2\n v_min_f32 %1, %2, %3\n v_min_f32 %2, %3, %4\n v_min_f32 %3, %4, %5\n v_min_f32 %4, %5, %6\n v_min_f32 %5, %6, %7\n v_min_f32 %6, %7, %0\n v_min_f32 %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
                if (OP == 2) asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP> void run(const char* name, float* d, unsigned long long mask, const char* mname) {
    const int iters = 2000, w = 8, blocks = 256 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 100, mask);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, mask); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s %-28s %7.3f ms -> %5.2f cycles per instruction per SIMD (2.4 GHz, 8 waves per SIMD)\n", name, mname, ms, ms * 1e6 / ((double)iters * 64 * w) * 2.4);
    fflush(stdout);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    struct { unsigned long long m; const char* n; } masks[] = {{~0ull, "all 64 lanes"}, {0xffffffffull, "lanes 0-31"}, {0xffffffff00000000ull, "lanes 32-63"}, {0xffffull, "lanes 0-15"},
        {0x1111111111111111ull, "16 lanes, one in four"}, {0x00000000ffff0000ull, "lanes 16-31"}, {1ull, "lane 0"}};
    for (auto& mk : masks) { run<0>("v_fma_f32", d, mk.m, mk.n); run<1>("v_min_f32", d, mk.m, mk.n); run<2>("v_mul_f32", d, mk.m, mk.n); }
    return 0;
}
