// common.h — shared host/device helpers for libide3d_hip.so (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/ide3d_hip.h"

namespace ide3d {

constexpr int kWave = 64;          // CDNA wavefront width
constexpr int kNumCU = 256;        // MI355X compute units
constexpr int kNumXCD = 8;         // accelerator complex dies (private L2 each)

// ---- error plumbing ------------------------------------------------------------------------

void set_error(const char* fmt, ...);

#define IDE3D_CHECK_ARG(cond, ...)                                   \
    do { if (!(cond)) { ide3d::set_error(__VA_ARGS__); return IDE3D_EINVAL; } } while (0)

#define IDE3D_CHECK_LAUNCH(what)                                     \
    do { hipError_t e_ = hipGetLastError();                          \
         if (e_ != hipSuccess) { ide3d::set_error("%s: %s", what, hipGetErrorString(e_)); \
                                 return IDE3D_ELAUNCH; } } while (0)

// ---- element type <-> fp32 math type ----------------------------------------------------------

template <class T> struct Elem;                       // storage type traits
template <> struct Elem<float> {
    using math_t = float;
    static __device__ __forceinline__ float  ld(const float* p) { return *p; }
    static __device__ __forceinline__ void   st(float* p, float v) { *p = v; }
};
template <> struct Elem<double> {
    using math_t = double;
    static __device__ __forceinline__ double ld(const double* p) { return *p; }
    static __device__ __forceinline__ void   st(double* p, double v) { *p = v; }
};
template <> struct Elem<__half> {
    using math_t = float;
    static __device__ __forceinline__ float  ld(const __half* p) { return __half2float(*p); }
    static __device__ __forceinline__ void   st(__half* p, float v) { *p = __float2half(v); }
};
template <> struct Elem<__hip_bfloat16> {
    using math_t = float;
    static __device__ __forceinline__ float  ld(const __hip_bfloat16* p) { return __bfloat162float(*p); }
    static __device__ __forceinline__ void   st(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }
};

// ---- small device utilities -------------------------------------------------------------------

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// floor division for possibly negative numerators (b > 0).
__host__ __device__ __forceinline__ int floordiv(int a, int b) {
    int q = a / b;
    return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

__host__ __device__ __forceinline__ int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ __forceinline__ int     cdiv(int a, int b) { return (a + b - 1) / b; }

// Grid size for a streaming kernel: enough workgroups to fill 256 CUs x 8 blocks, grid-stride beyond.
inline int stream_grid(int64_t work_items, int per_block) {
    int64_t g = cdiv64(work_items, per_block);
    const int64_t cap = (int64_t)kNumCU * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// XCD-aware remap of a linear workgroup id: hardware places block b on XCD b % 8, so consecutive
// *logical* tiles (which share halo / plane lines) are sent to the same XCD's L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int xcd = bid % kNumXCD;
    const int idx = bid / kNumXCD;
    const int q = nblocks / kNumXCD, r = nblocks % kNumXCD;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace ide3d
