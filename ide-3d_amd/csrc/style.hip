// style.hip — per-layer style preparation for the modulated convolutions, one launch per layer.
//
// Replaces the chain of tiny framework kernels that precedes every modulated convolution in inference:
//   styles = affine(w)                       FullyConnectedLayer, inversion/networks.py:152-165 (weight_gain = 1/sqrt(w_dim))
//   dcoefs = rsqrt(sum_{i,k} (W[o,i,k] * styles[n,i])^2 + 1e-8)          networks.py:91-93
// (addmm, square, matmul, add, rsqrt = 5-6 launches of ~4 us each, x 17 layers) and, for the image / semantic heads,
//   W'[n, o, i] = W_head[o, i] * (affine_head(w)[n, i] * weight_gain)    networks.py:700-706 folded for the fused head launch.
// All operands are a few hundred KB and live in L2; the kernels are latency-bound, so the point is launch count.
#include "common.h"

namespace ide3d {

// styles[n, :] = w[n, :] @ A^T * a_gain + b * b_gain (* out_scale).  Each wave owns rows i = wave, wave + nwaves, ...
// and keeps RB rows' loads in flight at once (the dot products are latency-, not bandwidth-bound).
__device__ __forceinline__ void affine_rows(const float* __restrict__ wv, const float* __restrict__ A, const float* __restrict__ b,
                                            int cin, int wdim, float a_gain, float b_gain, float out_scale, float* __restrict__ s_out) {
    constexpr int RB = 8;
    const int lane = lane_id(), wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int i0 = wid * RB; i0 < cin; i0 += nw * RB) {
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = 0.f;
        for (int k = lane; k < wdim; k += kWave) {
            const float wk = wv[k];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int i = min(i0 + r, cin - 1);
                acc[r] += A[(int64_t)i * wdim + k] * wk;
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) acc[r] += __shfl_xor(acc[r], off);
            const int i = i0 + r;
            if (lane == 0 && i < cin) s_out[i] = (acc[r] * a_gain + (b ? b[i] * b_gain : 0.f)) * out_scale;
        }
    }
}

// phase A, grid (ceil(cin / STYLE_ROWS), n): styles[n, rows of this block].
constexpr int STYLE_ROWS = 64;

__global__ void __launch_bounds__(512)
style_affine_kernel(const float* __restrict__ w, int64_t w_stride, const float* __restrict__ A, const float* __restrict__ b,
                    int cin, int wdim, float a_gain, float b_gain, float* __restrict__ styles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_w = lds;
    const int img = blockIdx.y, i0 = blockIdx.x * STYLE_ROWS;
    const int rows = min(STYLE_ROWS, cin - i0);
    for (int k = threadIdx.x; k < wdim; k += blockDim.x) s_w[k] = w[(int64_t)img * w_stride + k];
    __syncthreads();
    affine_rows(s_w, A + (int64_t)i0 * wdim, b ? b + i0 : nullptr, rows, wdim, a_gain, b_gain, 1.0f, styles + (int64_t)img * cin + i0);
}

// phase B, grid (ceil(cout / 64), n): dcoefs[n, o] = rsqrt(sum_i styles[n, i]^2 * wsq_t[i, o] + 1e-8); wsq_t [cin, cout].
__global__ void __launch_bounds__(1024)
style_demod_kernel(const float* __restrict__ styles, const float* __restrict__ wsq_t, int cin, int cout, float* __restrict__ dcoefs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_s = lds;                 // [cin] squared styles
    float* s_p = lds + cin;           // [16][64] partial sums
    const int img = blockIdx.y;
    for (int i = threadIdx.x; i < cin; i += blockDim.x) { const float v = styles[(int64_t)img * cin + i]; s_s[i] = v * v; }
    __syncthreads();
    const int col = threadIdx.x & 63, part = threadIdx.x >> 6;          // 16 slices of the ci range per output column
    const int co = blockIdx.x * 64 + col;
    float acc = 0.f;
    if (co < cout)
        for (int i = part; i < cin; i += 16) acc += s_s[i] * wsq_t[(int64_t)i * cout + co];
    s_p[part * 64 + col] = acc;
    __syncthreads();
    if (part == 0 && co < cout) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += s_p[q * 64 + col];
        dcoefs[(int64_t)img * cout + co] = rsqrtf(t + 1e-8f);
    }
}

// grid: (n, ceil(cin / FOLD_CI)).  Two heads (rgb, seg) sharing w: out[n, o, i] = W_h[o, i] * styles_h[n, i], heads concatenated
// along o.  A block owns FOLD_CI input channels: it needs only its own styles (FOLD_CI affine rows per head, all in flight
// at once) and writes the matching columns of every output row.
constexpr int FOLD_CI = 64;

__global__ void __launch_bounds__(1024)
fold_heads_kernel(const float* __restrict__ w, int64_t w_stride, int cin, int wdim,
                  const float* __restrict__ A0, const float* __restrict__ b0, const float* __restrict__ W0, int cout0, float gain0,
                  const float* __restrict__ A1, const float* __restrict__ b1, const float* __restrict__ W1, int cout1, float gain1,
                  float a_gain, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_w = lds; float* s0 = lds + wdim; float* s1 = s0 + FOLD_CI;
    const int img = blockIdx.x;
    const int ci0 = blockIdx.y * FOLD_CI;
    const int nci = min(FOLD_CI, cin - ci0);
    for (int k = threadIdx.x; k < wdim; k += blockDim.x) s_w[k] = w[(int64_t)img * w_stride + k];
    __syncthreads();
    affine_rows(s_w, A0 + (int64_t)ci0 * wdim, b0 ? b0 + ci0 : nullptr, nci, wdim, a_gain, 1.0f, gain0, s0);
    affine_rows(s_w, A1 + (int64_t)ci0 * wdim, b1 ? b1 + ci0 : nullptr, nci, wdim, a_gain, 1.0f, gain1, s1);
    __syncthreads();
    float* o = out + (int64_t)img * (cout0 + cout1) * cin;
    for (int e = threadIdx.x; e < (cout0 + cout1) * FOLD_CI; e += blockDim.x) {
        const int co = e / FOLD_CI, cl = e - co * FOLD_CI;
        if (cl >= nci) continue;
        const int ci = ci0 + cl;
        o[(int64_t)co * cin + ci] = (co < cout0) ? W0[(int64_t)co * cin + ci] * s0[cl] : W1[(int64_t)(co - cout0) * cin + ci] * s1[cl];
    }
}

}  // namespace ide3d

extern "C" int ide3d_style_demod(const float* w, int64_t w_stride, const float* affine_w, const float* affine_b, const float* wsq_t,
                                 int32_t n, int32_t cin, int32_t cout, int32_t wdim, float affine_gain, float bias_gain,
                                 float* styles, float* dcoefs, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(w && affine_w && styles, "style_demod: null pointer");
    IDE3D_CHECK_ARG(dcoefs == nullptr || wsq_t != nullptr, "style_demod: dcoefs needs wsq_t");
    IDE3D_CHECK_ARG(n > 0 && cin > 0 && wdim > 0 && (dcoefs == nullptr || cout > 0), "style_demod: bad shape");
    IDE3D_CHECK_ARG((size_t)wdim * sizeof(float) <= 60 * 1024 && ((size_t)cin + 1024) * sizeof(float) <= 60 * 1024,
                    "style_demod: w_dim / cin too large for LDS staging");
    hipLaunchKernelGGL(style_affine_kernel, dim3(cdiv(cin, STYLE_ROWS), n), dim3(512), (size_t)wdim * sizeof(float), (hipStream_t)stream,
                       w, w_stride, affine_w, affine_b, cin, wdim, affine_gain, bias_gain, styles);
    if (dcoefs)
        hipLaunchKernelGGL(style_demod_kernel, dim3(cdiv(cout, 64), n), dim3(1024), ((size_t)cin + 1024) * sizeof(float), (hipStream_t)stream,
                           styles, wsq_t, cin, cout, dcoefs);
    IDE3D_CHECK_LAUNCH("style_demod");
    return IDE3D_OK;
}

extern "C" int ide3d_fold_heads(const float* w, int64_t w_stride, int32_t n, int32_t cin, int32_t wdim, float affine_gain,
                                const float* a0, const float* b0, const float* w0, int32_t cout0, float gain0,
                                const float* a1, const float* b1, const float* w1, int32_t cout1, float gain1,
                                float* out, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(w && a0 && w0 && a1 && w1 && out, "fold_heads: null pointer");
    IDE3D_CHECK_ARG(n > 0 && cin > 0 && wdim > 0 && cout0 > 0 && cout1 > 0, "fold_heads: bad shape");
    const size_t lds = ((size_t)wdim + 2 * FOLD_CI) * sizeof(float);
    IDE3D_CHECK_ARG(lds <= 60 * 1024, "fold_heads: w_dim + cin too large for LDS staging");
    hipLaunchKernelGGL(fold_heads_kernel, dim3(n, cdiv(cin, FOLD_CI)), dim3(1024), lds, (hipStream_t)stream,
                       w, w_stride, cin, wdim, a0, b0, w0, cout0, gain0, a1, b1, w1, cout1, gain1, affine_gain, out);
    IDE3D_CHECK_LAUNCH("fold_heads");
    return IDE3D_OK;
}
