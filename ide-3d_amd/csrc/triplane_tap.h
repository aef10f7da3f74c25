// triplane_tap.h — bilinear tap set-up shared by the stand-alone gather (triplane.hip) and the fused
// ray-marcher (raymarch.hip).  Index math is bit-exact w.r.t. ATen grid_sampler_2d
// (align_corners=False, zeros padding): see DESIGN.md "tap index contract".
#pragma once
#include "common.h"

namespace ide3d {

struct Tap2 {
    int ix0, iy0;          // floor(u), floor(v)
    float w00, w01, w10, w11;  // nw (x0,y0), ne (x1,y0), sw (x0,y1), se (x1,y1)
    unsigned mask;         // bit0 nw, bit1 ne, bit2 sw, bit3 se in-bounds
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

__device__ __forceinline__ float4 f4_fma(float4 a, float w, float4 acc) {
    acc.x += a.x * w; acc.y += a.y * w; acc.z += a.z * w; acc.w += a.w * w;
    return acc;
}

// One plane, one 4-channel slice of one sample (channels_last planes).
__device__ __forceinline__ float4 gather_plane_cl(const float* __restrict__ base, int64_t sH, int64_t sW,
                                                  const Tap2& t) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p00 = base + t.iy0 * sH + t.ix0 * sW;
    float4 v00 = (t.mask & 1u) ? *reinterpret_cast<const float4*>(p00) : z;
    float4 v01 = (t.mask & 2u) ? *reinterpret_cast<const float4*>(p00 + sW) : z;
    float4 v10 = (t.mask & 4u) ? *reinterpret_cast<const float4*>(p00 + sH) : z;
    float4 v11 = (t.mask & 8u) ? *reinterpret_cast<const float4*>(p00 + sH + sW) : z;
    float4 acc = z;
    acc = f4_fma(v00, t.w00, acc);
    acc = f4_fma(v01, t.w01, acc);
    acc = f4_fma(v10, t.w10, acc);
    acc = f4_fma(v11, t.w11, acc);
    return acc;
}

}  // namespace ide3d
