// triplane_tap.h — bilinear tap set-up shared by the stand-alone gather (triplane.hip) and the fused
// ray-marcher (raymarch.hip).  Index math is bit-exact w.r.t. ATen grid_sampler_2d
// (align_corners=False, zeros padding): see DESIGN.md "tap index contract".
//
// Two views of a tap:
//   Tap2     raw integer origin (floor(u), floor(v)), the four bilinear weights and the in-bounds mask —
//            what the parity hook `ide3d_triplane_taps` exports and the backward kernel consumes;
//   TapAddr  what the gather kernels use: element offsets of the four taps with the coordinates clamped into
//            the plane (so every load is unconditional — no exec-mask branches around loads) and the weights
//            of out-of-bounds taps forced to zero (zeros padding).
#pragma once
#include "common.h"

namespace ide3d {

struct Tap2 {
    int ix0, iy0;          // floor(u), floor(v)
    float w00, w01, w10, w11;  // nw (x0,y0), ne (x1,y0), sw (x0,y1), se (x1,y1)
    unsigned mask;         // bit0 nw, bit1 ne, bit2 sw, bit3 se in-bounds
};

struct TapAddr {
    int o00, o01, o10, o11;        // element offsets (clamped in-plane)
    float w00, w01, w10, w11;      // weights, 0 for out-of-bounds taps
};

__device__ __forceinline__ float unnormalize(float c, int size) {
    // ((c + 1) * size - 1) / 2 — explicit rounding at every step (no FMA contraction).
    return __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(c, 1.0f), (float)size), 1.0f), 0.5f);
}

__device__ __forceinline__ Tap2 make_tap(float cx, float cy, int W, int H) {
    Tap2 t;
    const float u = unnormalize(cx, W), v = unnormalize(cy, H);
    const float fu = floorf(u), fv = floorf(v);
    // Saturate so that wild coordinates (inf / huge) cannot overflow the int conversion.
    const float fuc = fminf(fmaxf(fu, -2.0f), (float)W + 1.0f);
    const float fvc = fminf(fmaxf(fv, -2.0f), (float)H + 1.0f);
    t.ix0 = (int)fuc; t.iy0 = (int)fvc;
    const float x1 = __fadd_rn(fu, 1.0f), y1 = __fadd_rn(fv, 1.0f);
    const float ax = __fsub_rn(x1, u), bx = __fsub_rn(u, fu);
    const float ay = __fsub_rn(y1, v), by = __fsub_rn(v, fv);
    t.w00 = __fmul_rn(ax, ay); t.w01 = __fmul_rn(bx, ay);
    t.w10 = __fmul_rn(ax, by); t.w11 = __fmul_rn(bx, by);
    const bool x0ok = t.ix0 >= 0 && t.ix0 < W, x1ok = t.ix0 + 1 >= 0 && t.ix0 + 1 < W;
    const bool y0ok = t.iy0 >= 0 && t.iy0 < H, y1ok = t.iy0 + 1 >= 0 && t.iy0 + 1 < H;
    const bool finite = (u == u) && (v == v) && fu == fuc && fv == fvc;
    t.mask = finite ? ((x0ok && y0ok) ? 1u : 0u) | ((x1ok && y0ok) ? 2u : 0u) |
                      ((x0ok && y1ok) ? 4u : 0u) | ((x1ok && y1ok) ? 8u : 0u) : 0u;
    return t;
}

// Clamped offsets + masked weights.  sH / sW are element strides (they fit 32 bits: checked on the host).
__device__ __forceinline__ TapAddr tap_addr(const Tap2& t, int W, int H, int sH, int sW) {
    TapAddr a;
    const int x0 = min(max(t.ix0, 0), W - 1), x1 = min(max(t.ix0 + 1, 0), W - 1);
    const int y0 = min(max(t.iy0, 0), H - 1), y1 = min(max(t.iy0 + 1, 0), H - 1);
    const int r0 = y0 * sH, r1 = y1 * sH, c0 = x0 * sW, c1 = x1 * sW;
    a.o00 = r0 + c0; a.o01 = r0 + c1; a.o10 = r1 + c0; a.o11 = r1 + c1;
    a.w00 = (t.mask & 1u) ? t.w00 : 0.f;
    a.w01 = (t.mask & 2u) ? t.w01 : 0.f;
    a.w10 = (t.mask & 4u) ? t.w10 : 0.f;
    a.w11 = (t.mask & 8u) ? t.w11 : 0.f;
    return a;
}

__device__ __forceinline__ TapAddr make_tap_addr(float cx, float cy, int W, int H, int sH, int sW) {
    return tap_addr(make_tap(cx, cy, W, H), W, H, sH, sW);
}

__device__ __forceinline__ float4 ld4(const float* p) {
    return *reinterpret_cast<const float4*>(__builtin_assume_aligned(p, 16));
}

__device__ __forceinline__ float4 f4_fma(float4 a, float w, float4 acc) {
    acc.x += a.x * w; acc.y += a.y * w; acc.z += a.z * w; acc.w += a.w * w;
    return acc;
}

// One plane, one 4-channel slice of one sample (channels_last planes): four unconditional 16-byte loads.
__device__ __forceinline__ float4 gather_plane_cl(const float* __restrict__ base, const TapAddr& a) {
    const float4 v00 = ld4(base + a.o00);
    const float4 v01 = ld4(base + a.o01);
    const float4 v10 = ld4(base + a.o10);
    const float4 v11 = ld4(base + a.o11);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    acc = f4_fma(v00, a.w00, acc);
    acc = f4_fma(v01, a.w01, acc);
    acc = f4_fma(v10, a.w10, acc);
    acc = f4_fma(v11, a.w11, acc);
    return acc;
}

}  // namespace ide3d
