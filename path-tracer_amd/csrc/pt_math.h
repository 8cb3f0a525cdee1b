// pt_math.h -- device-side arithmetic for the gfx950 kernels.
//
// The kernels compute in strict IEEE-754 binary32: this translation unit is compiled with
// -ffp-contract=off (no FMA contraction) and correctly rounded divide/sqrt, and every
// compound operation below fixes its association order.  The order is the one DESIGN.md
// ("Arithmetic contract") states; the CPU oracle under oracle/ restates the same contract
// independently, which is what makes hit indices and path state comparable bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 F3(float4 v) { return F3(v.x, v.y, v.z); }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float length3(f3 a) { return __builtin_sqrtf(dot3(a, a)); }
__device__ __forceinline__ f3 normalize3(f3 a) { return a / length3(a); }
__device__ __forceinline__ float fmin_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ f3 min3(f3 a, f3 b) { return F3(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)); }
__device__ __forceinline__ f3 max3(f3 a, f3 b) { return F3(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)); }
__device__ __forceinline__ f3 rcp3(f3 d) { return F3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }

// Three matrix rows (each float4 = row r of a 3x4 affine map) applied to (v, w).
// Reference: the mat4x4f * vec4f products of scene.js:208-216, 337-345, 366-374.
__device__ __forceinline__ f3 xform(float4 r0, float4 r1, float4 r2, f3 v, float w) {
    return F3(((r0.x * v.x + r0.y * v.y) + r0.z * v.z) + r0.w * w,
              ((r1.x * v.x + r1.y * v.y) + r1.z * v.z) + r1.w * w,
              ((r2.x * v.x + r2.y * v.y) + r2.z * v.z) + r2.w * w);
}

// WGSL u32(f32): truncate toward zero, saturate; NaN is backend-defined there, 0 here.
__device__ __forceinline__ uint32_t f32_to_u32(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}

// sin and cos of theta in [0, 8): quadrant k = trunc(theta*2/pi + .5), three-term Cody-Waite
// reduction by pi/2, Cephes single-precision minimax polynomials on [-pi/4, pi/4].
__device__ __forceinline__ void pt_sincos(float theta, float& s_out, float& c_out) {
    int   k  = (int)(theta * 0.636619772367581343f + 0.5f);
    float kf = (float)k;
    float r  = ((theta - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
    float z  = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    bool  swap = (k & 1) != 0;
    float s = swap ? pc : ps, c = swap ? ps : pc;
    s_out = (k & 2) ? -s : s;
    c_out = ((k + 1) & 2) ? -c : c;
}

// material.js:154-158 (logic.js:235-239, camera.js:124-128)
__device__ __forceinline__ uint32_t base_hash(uint32_t px, uint32_t py) {
    uint32_t p2x = 1103515245u * ((px >> 1) ^ py);
    uint32_t p2y = 1103515245u * ((py >> 1) ^ px);
    uint32_t h32 = 1103515245u * (p2x ^ (p2y >> 3));
    return h32 ^ (h32 >> 16);
}
// material.js:159-163.  f32(0x7fffffff) is 2^31 after rounding, so the divide is an exact scaling.
__device__ __forceinline__ float2 rand2(float seed) {
    uint32_t n = base_hash(__float_as_uint(seed + 1.0f), __float_as_uint(seed + 2.0f));
    float2 r;
    r.x = (float)(n & 0x7fffffffu) * 4.656612873077392578125e-10f;
    r.y = (float)((n * 48271u) & 0x7fffffffu) * 4.656612873077392578125e-10f;
    return r;
}
