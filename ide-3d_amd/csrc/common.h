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

// Non-default compile-time knobs of the translation units that have any (each unit reports its own: scripts/micro/build_variant.sh recompiles
// single files); '!' marks knobs that change results or drop a safety property.  Joined by ide3d_build_flags() (core.hip).
const char* modconv_build_flags();
const char* triplane_tile_build_flags();
const char* raymarch_build_flags();

// Exclusive residency is a property of the built kernel ON the device it runs on: an LDS-fed bf16 / fp16 matrix loop
// must be the only workgroup on its CU.  It follows from register claims and workgroup sizes; a toolchain that splits the register file
// differently, or a partitioned / CU-masked device, could silently break it.  Every launch of such a kernel goes through
// IDE3D_EXCL_LAUNCH, which asks the runtime ONCE per (kernel instantiation, device) how many workgroups fit a CU; anything but one is
// recorded (ide3d_exclusive_violations(), checked by hip_plugin and the GPU tests) and the launch fails loudly with IDE3D_ELAUNCH.
void note_exclusive_violation(const char* kernel, int blocks_per_cu);
void refuse_launch(const char* kernel);          // this thread's next IDE3D_CHECK_LAUNCH fails
bool take_refused();
int exclusive_violations();
// The answer is cached per (kernel ADDRESS, device, threads, dynamic LDS): round 5 kept it in a function-template static, which every kernel
// instantiation of the same SIGNATURE shared (all modconv_split_kernel<...> variants, all render_rays_kernel<...>): only the first-launched
// variant was ever asked about.  core.hip owns the table.
bool exclusive_checked(const void* kernel, int threads, size_t dyn_lds, const char* name);
template <typename K>
inline bool exclusive_on_this_device(K kernel, int threads, size_t dyn_lds, const char* name) {
#ifdef IDE3D_SP_SHARED_SIMD
    return true;                                   // A/B experiment build without the register claims
#endif
    return exclusive_checked(reinterpret_cast<const void*>(kernel), threads, dyn_lds, name);
}
#define IDE3D_EXCL_LAUNCH(KERNEL, GRID, THREADS, DYN_LDS, ST, ...) \
    do { if (ide3d::exclusive_on_this_device(KERNEL, (THREADS), (DYN_LDS), #KERNEL)) hipLaunchKernelGGL(KERNEL, GRID, dim3(THREADS), DYN_LDS, ST, __VA_ARGS__); } while (0)
#define IDE3D_STR_(x) #x
#define IDE3D_STR(x) IDE3D_STR_(x)

#define IDE3D_CHECK_ARG(cond, ...)                                   \
    do { if (!(cond)) { ide3d::set_error(__VA_ARGS__); return IDE3D_EINVAL; } } while (0)

#define IDE3D_CHECK_LAUNCH(what)                                     \
    do { hipError_t e_ = hipGetLastError();                          \
         if (e_ != hipSuccess) { ide3d::set_error("%s: %s", what, hipGetErrorString(e_)); \
                                 return IDE3D_ELAUNCH; }                \
         if (ide3d::take_refused()) return IDE3D_ELAUNCH; } while (0)

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

// Raise slot (blockIdx.x % IDE3D_AMAX_SLOTS) of image n's amax row to v (v >= 0): called ONCE per workgroup (callers reduce inside the
// workgroup first), a fire-and-forget atomic maximum (no return value, nothing waits for it), 32 cache lines per image.  (Measured on the
// way here: one returning atomic per wave on one word per image = 38 ms per frame instead of 5; per-wave reads of 32 words sharing ONE
// cache line 3 ms; a read-before-atomic per workgroup keeps the workgroup alive for an L2 round trip at its very end.)
__device__ __forceinline__ void amax_raise(float* amax, int n, float v) {
    unsigned* const slot = reinterpret_cast<unsigned*>(amax) + (size_t)n * IDE3D_AMAX_FLOATS + (blockIdx.x % IDE3D_AMAX_SLOTS) * IDE3D_AMAX_STRIDE;
    __hip_atomic_fetch_max(slot, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Workgroup maximum of v (>= 0) through `scratch` (>= blockDim.x / 64 floats of LDS nobody else uses between the two barriers inside),
// then one amax_raise by thread 0.  Every thread of the workgroup must call it.
__device__ __forceinline__ void amax_raise_block(float* amax, int n, float v, float* scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = scratch[0];
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, scratch[i]);
        if (m > 0.f) amax_raise(amax, n, m);          // (+inf included: 0x7f800000 compares above every finite pattern)
    }
}
// amax of a grid-stride element-wise kernel whose threads may see several images: per-workgroup maxima per image in LDS (atomic
// maxima on `s_am[64]`, images folded modulo 64: p_n <= 64 required), then one amax_raise per image and workgroup.
__device__ __forceinline__ void amax_lds_flush(unsigned* s_am, int n, float v) { if (v > 0.f) atomicMax(&s_am[n & 63], __float_as_uint(v)); }
__device__ __forceinline__ void amax_lds_commit(float* amax, const unsigned* s_am, int p_n) {
    __syncthreads();
    if ((int)threadIdx.x < p_n && threadIdx.x < 64 && s_am[threadIdx.x] != 0u) amax_raise(amax, (int)threadIdx.x, __uint_as_float(s_am[threadIdx.x]));
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
