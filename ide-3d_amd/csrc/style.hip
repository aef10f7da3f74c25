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

// styles[n, :] = w[n, :] @ A^T * a_gain + b * b_gain (* out_scale).  Each wave owns rows i = wave*RB .. wave*RB + RB-1,
// then + nwaves*RB, ...  The dot products are latency-, not bandwidth-bound (the weights come from HBM / the infinity
// cache, ~2 us per dependent round trip), so a wave requests all RB rows of a 256-column slab as 16-byte loads before
// it touches any of them: RB loads in flight per lane, wdim / 256 round trips per row group.
template <int RB>
__device__ __forceinline__ void affine_rows(const float* __restrict__ wv, const float* __restrict__ A, const float* __restrict__ b,
                                            int cin, int wdim, float a_gain, float b_gain, float out_scale, float* __restrict__ s_out) {
    const int lane = lane_id(), wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const bool vec = (wdim % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    for (int i0 = wid * RB; i0 < cin; i0 += nw * RB) {
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = 0.f;
        if (vec) {
#pragma unroll 2
            for (int k = lane * 4; k < wdim; k += kWave * 4) {
                float4 a[RB];
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    a[r] = *reinterpret_cast<const float4*>(A + (int64_t)min(i0 + r, cin - 1) * wdim + k);
                const float4 wk = *reinterpret_cast<const float4*>(wv + k);
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] += (a[r].x * wk.x + a[r].y * wk.y) + (a[r].z * wk.z + a[r].w * wk.w);
            }
        } else {
            for (int k = lane; k < wdim; k += kWave) {
                const float wk = wv[k];
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] += A[(int64_t)min(i0 + r, cin - 1) * wdim + k] * wk;
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

// Same, run by one half (8 waves) of a 16-wave workgroup: rows wave%8 * 8 ...
template <int HALF>
__device__ __forceinline__ void affine_rows_half(const float* __restrict__ wv, const float* __restrict__ A, const float* __restrict__ b,
                                                 int cin, int wdim, float a_gain, float out_scale, float* __restrict__ s_out) {
    constexpr int RB = 8;
    const int lane = lane_id(), wid = (threadIdx.x >> 6) - HALF * 8;
    const bool vec = (wdim % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    for (int i0 = wid * RB; i0 < cin; i0 += 8 * RB) {
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = 0.f;
        if (vec) {
#pragma unroll 2
            for (int k = lane * 4; k < wdim; k += kWave * 4) {
                float4 a[RB];
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    a[r] = *reinterpret_cast<const float4*>(A + (int64_t)min(i0 + r, cin - 1) * wdim + k);
                const float4 wk = *reinterpret_cast<const float4*>(wv + k);
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] += (a[r].x * wk.x + a[r].y * wk.y) + (a[r].z * wk.z + a[r].w * wk.w);
            }
        } else {
            for (int k = lane; k < wdim; k += kWave) {
                const float wk = wv[k];
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] += A[(int64_t)min(i0 + r, cin - 1) * wdim + k] * wk;
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) acc[r] += __shfl_xor(acc[r], off);
            const int i = i0 + r;
            if (lane == 0 && i < cin) s_out[i] = (acc[r] * a_gain + (b ? b[i] : 0.f)) * out_scale;
        }
    }
}

// phase A, grid (ceil(cin / STYLE_ROWS), n): styles[n, rows of this block].
constexpr int STYLE_ROWS = 32;           // 4 waves x 8 rows: cin / 32 x n workgroups (64 for cin = 512, n = 4)

__device__ __forceinline__ void style_affine_body(const float* __restrict__ w, int64_t w_stride, const float* __restrict__ A, const float* __restrict__ b,
                                                  int cin, int wdim, float a_gain, float b_gain, float* __restrict__ styles, int bx, int img) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_w = lds;
    const int i0 = bx * STYLE_ROWS;
    const int rows = min(STYLE_ROWS, cin - i0);
    for (int k = threadIdx.x; k < wdim; k += blockDim.x) s_w[k] = w[(int64_t)img * w_stride + k];
    __syncthreads();
    affine_rows<8>(s_w, A + (int64_t)i0 * wdim, b ? b + i0 : nullptr, rows, wdim, a_gain, b_gain, 1.0f, styles + (int64_t)img * cin + i0);
}

__global__ void __launch_bounds__(256)
style_affine_kernel(const float* __restrict__ w, int64_t w_stride, const float* __restrict__ A, const float* __restrict__ b,
                    int cin, int wdim, float a_gain, float b_gain, float* __restrict__ styles) {
    style_affine_body(w, w_stride, A, b, cin, wdim, a_gain, b_gain, styles, blockIdx.x, blockIdx.y);
}

// phase B, grid ceil(cout / 32): dcoefs[img, o] = rsqrt(sum_i styles[img, i]^2 * wsq_t[i, o] + 1e-8) for every image; wsq_t
// [cin, cout].  A workgroup owns 32 output columns: thread = (column, one of 32 interleaved slices of the ci range); a
// thread requests its whole slice of wsq_t (up to DEMOD_MAX_CI / 32 values) before using any, and every value serves
// all images (wsq_t is read once, not once per image).
constexpr int DEMOD_COLS = 32, DEMOD_PARTS = 32, DEMOD_MAX_CI = 512, DEMOD_MAX_N = 8;

__device__ __forceinline__ void style_demod_body(const float* __restrict__ styles, const float* __restrict__ wsq_t, int n, int cin, int cout,
                                                 float* __restrict__ dcoefs, int bx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_s = lds;                                   // [n][cin] squared styles
    float* s_p = lds + (size_t)n * cin;                 // [DEMOD_PARTS][DEMOD_COLS] partial sums of one image
    for (int i = threadIdx.x; i < n * cin; i += blockDim.x) { const float v = styles[i]; s_s[i] = v * v; }
    __syncthreads();
    const int col = threadIdx.x & (DEMOD_COLS - 1), part = threadIdx.x / DEMOD_COLS;
    const int co = bx * DEMOD_COLS + col, coc = min(co, cout - 1);
    float acc[DEMOD_MAX_N];
#pragma unroll
    for (int g = 0; g < DEMOD_MAX_N; ++g) acc[g] = 0.f;
    for (int c0 = 0; c0 < cin; c0 += DEMOD_MAX_CI) {
        constexpr int PER = DEMOD_MAX_CI / DEMOD_PARTS;
        float v[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = c0 + part + q * DEMOD_PARTS;
            v[q] = wsq_t[(int64_t)min(i, cin - 1) * cout + coc];
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = c0 + part + q * DEMOD_PARTS;
            if (i < cin) {
#pragma unroll
                for (int g = 0; g < DEMOD_MAX_N; ++g)
                    if (g < n) acc[g] += s_s[g * cin + i] * v[q];
            }
        }
    }
    for (int g = 0; g < n; ++g) {
        s_p[part * DEMOD_COLS + col] = acc[g];
        __syncthreads();
        if (part == 0 && co < cout) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < DEMOD_PARTS; ++q) t += s_p[q * DEMOD_COLS + col];
            dcoefs[(int64_t)g * cout + co] = rsqrtf(t + 1e-8f);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(1024)
style_demod_kernel(const float* __restrict__ styles, const float* __restrict__ wsq_t, int n, int cin, int cout, float* __restrict__ dcoefs) {
    style_demod_body(styles, wsq_t, n, cin, cout, dcoefs, blockIdx.x);
}

// grid: (n, ceil(cin / FOLD_CI)).  Two heads (rgb, seg) sharing w: out[n, o, i] = W_h[o, i] * styles_h[n, i], heads concatenated
// along o.  A block owns FOLD_CI input channels: it needs only its own styles (FOLD_CI affine rows per head, all in flight
// at once) and writes the matching columns of every output row.
constexpr int FOLD_CI = 64;

__device__ __forceinline__ void fold_heads_body(const float* __restrict__ w, int64_t w_stride, int cin, int wdim,
                  const float* __restrict__ A0, const float* __restrict__ b0, const float* __restrict__ W0, int cout0, float gain0,
                  const float* __restrict__ A1, const float* __restrict__ b1, const float* __restrict__ W1, int cout1, float gain1,
                  float a_gain, float* __restrict__ out, int img, int by) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_w = lds; float* s0 = lds + wdim; float* s1 = s0 + FOLD_CI;
    const int ci0 = by * FOLD_CI;
    const int nci = min(FOLD_CI, cin - ci0);
    for (int k = threadIdx.x; k < wdim; k += blockDim.x) s_w[k] = w[(int64_t)img * w_stride + k];
    __syncthreads();
    // 16 waves: waves 0-7 take head 0 (8 rows each), waves 8-15 head 1 — both heads' rows are in flight together
    if ((threadIdx.x >> 6) < 8) affine_rows_half<0>(s_w, A0 + (int64_t)ci0 * wdim, b0 ? b0 + ci0 : nullptr, nci, wdim, a_gain, gain0, s0);
    else                        affine_rows_half<1>(s_w, A1 + (int64_t)ci0 * wdim, b1 ? b1 + ci0 : nullptr, nci, wdim, a_gain, gain1, s1);
    __syncthreads();
    float* o = out + (int64_t)img * (cout0 + cout1) * cin;
    for (int e = threadIdx.x; e < (cout0 + cout1) * FOLD_CI; e += blockDim.x) {
        const int co = e / FOLD_CI, cl = e - co * FOLD_CI;
        if (cl >= nci) continue;
        const int ci = ci0 + cl;
        o[(int64_t)co * cin + ci] = (co < cout0) ? W0[(int64_t)co * cin + ci] * s0[cl] : W1[(int64_t)(co - cout0) * cin + ci] * s1[cl];
    }
}

__global__ void __launch_bounds__(1024)
fold_heads_kernel(const float* __restrict__ w, int64_t w_stride, int cin, int wdim,
                  const float* __restrict__ A0, const float* __restrict__ b0, const float* __restrict__ W0, int cout0, float gain0,
                  const float* __restrict__ A1, const float* __restrict__ b1, const float* __restrict__ W1, int cout1, float gain1,
                  float a_gain, float* __restrict__ out) {
    fold_heads_body(w, w_stride, cin, wdim, A0, b0, W0, cout0, gain0, A1, b1, W1, cout1, gain1, a_gain, out, blockIdx.x, blockIdx.y);
}

// ---- all layers of a pass in three launches (round 3) ------------------------------------------------------------------------
// The 43 launches above (13 + 7 pairs and 7 + 3 head foldings of a 64 -> 512 pass) are 6 - 17 us each and depend only on ws: as
// separate nodes of the captured graph they run, back to back, BEFORE the first convolution (the graph executor does not start the
// convolution branch beside them: scripts/step_timeline.py) — ~0.4 ms of a 4.2 ms pass.  The batched forms take the same per-layer
// arguments as a table in the kernel arguments and map a flat block index to (layer, block): same code per block, same results.
struct StyleBatch { ide3d_style_job job[IDE3D_STYLE_BATCH_MAX]; int blk0[IDE3D_STYLE_BATCH_MAX + 1]; int njobs; };
struct FoldBatch { ide3d_fold_job job[IDE3D_STYLE_BATCH_MAX]; int blk0[IDE3D_STYLE_BATCH_MAX + 1]; int njobs; };

template <class B>
__device__ __forceinline__ int batch_job(const B& b, int bx) {
    int j = 0;
    while (j + 1 < b.njobs && bx >= b.blk0[j + 1]) ++j;
    return j;
}

__global__ void __launch_bounds__(256)
style_affine_batch_kernel(const StyleBatch b, int wdim) {
    const int j = batch_job(b, blockIdx.x);
    const ide3d_style_job& q = b.job[j];
    style_affine_body(q.w, q.w_stride, q.affine_w, q.affine_b, q.cin, wdim, q.affine_gain, q.bias_gain, q.styles, blockIdx.x - b.blk0[j], blockIdx.y);
}

__global__ void __launch_bounds__(1024)
style_demod_batch_kernel(const StyleBatch b, int n) {
    const int j = batch_job(b, blockIdx.x);
    const ide3d_style_job& q = b.job[j];
    style_demod_body(q.styles, q.wsq_t, n, q.cin, q.cout, q.dcoefs, blockIdx.x - b.blk0[j]);
}

__global__ void __launch_bounds__(1024)
fold_heads_batch_kernel(const FoldBatch b, int wdim) {
    const int j = batch_job(b, blockIdx.y);
    const ide3d_fold_job& q = b.job[j];
    fold_heads_body(q.w, q.w_stride, q.cin, wdim, q.a0, q.b0, q.w0, q.cout0, q.gain0, q.a1, q.b1, q.w1, q.cout1, q.gain1, q.affine_gain, q.out,
                    blockIdx.x, blockIdx.y - b.blk0[j]);
}

}  // namespace ide3d

extern "C" int ide3d_style_demod(const float* w, int64_t w_stride, const float* affine_w, const float* affine_b, const float* wsq_t,
                                 int32_t n, int32_t cin, int32_t cout, int32_t wdim, float affine_gain, float bias_gain,
                                 float* styles, float* dcoefs, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(w && affine_w && styles, "style_demod: null pointer");
    IDE3D_CHECK_ARG(dcoefs == nullptr || wsq_t != nullptr, "style_demod: dcoefs needs wsq_t");
    IDE3D_CHECK_ARG(n > 0 && cin > 0 && wdim > 0 && (dcoefs == nullptr || cout > 0), "style_demod: bad shape");
    IDE3D_CHECK_ARG((size_t)wdim * sizeof(float) <= 60 * 1024, "style_demod: w_dim too large for LDS staging");
    IDE3D_CHECK_ARG(dcoefs == nullptr || ((size_t)DEMOD_MAX_N * cin + DEMOD_PARTS * DEMOD_COLS) * sizeof(float) <= 60 * 1024,
                    "style_demod: cin too large for LDS staging");
    hipLaunchKernelGGL(style_affine_kernel, dim3(cdiv(cin, STYLE_ROWS), n), dim3(256), (size_t)wdim * sizeof(float), (hipStream_t)stream,
                       w, w_stride, affine_w, affine_b, cin, wdim, affine_gain, bias_gain, styles);
    if (dcoefs)
        for (int g0 = 0; g0 < n; g0 += DEMOD_MAX_N) {            // DEMOD_MAX_N images per launch share one pass over wsq_t
            const int ng = min(DEMOD_MAX_N, n - g0);
            hipLaunchKernelGGL(style_demod_kernel, dim3(cdiv(cout, DEMOD_COLS)), dim3(1024),
                               ((size_t)ng * cin + DEMOD_PARTS * DEMOD_COLS) * sizeof(float), (hipStream_t)stream,
                               styles + (int64_t)g0 * cin, wsq_t, ng, cin, cout, dcoefs + (int64_t)g0 * cout);
        }
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

extern "C" int ide3d_style_demod_batch(const ide3d_style_job* jobs, int32_t njobs, int32_t n, int32_t wdim, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(jobs && njobs > 0 && njobs <= IDE3D_STYLE_BATCH_MAX, "style_demod_batch: 1..%d jobs", IDE3D_STYLE_BATCH_MAX);
    IDE3D_CHECK_ARG(n > 0 && n <= DEMOD_MAX_N && wdim > 0, "style_demod_batch: 1..%d images", DEMOD_MAX_N);
    IDE3D_CHECK_ARG((size_t)wdim * sizeof(float) <= 60 * 1024, "style_demod_batch: w_dim too large for LDS staging");
    StyleBatch a{}, d{};
    int na = 0, nd = 0, max_cin = 0;
    for (int j = 0; j < njobs; ++j) {
        const ide3d_style_job& q = jobs[j];
        IDE3D_CHECK_ARG(q.w && q.affine_w && q.styles && q.cin > 0, "style_demod_batch: job %d: null pointer / bad shape", j);
        IDE3D_CHECK_ARG(q.dcoefs == nullptr || (q.wsq_t != nullptr && q.cout > 0), "style_demod_batch: job %d: dcoefs needs wsq_t", j);
        a.job[j] = q; a.blk0[j] = na; na += cdiv(q.cin, STYLE_ROWS);
        if (q.dcoefs) { d.job[d.njobs] = q; d.blk0[d.njobs] = nd; nd += cdiv(q.cout, DEMOD_COLS); ++d.njobs; if (q.cin > max_cin) max_cin = q.cin; }
    }
    a.njobs = njobs; a.blk0[njobs] = na; d.blk0[d.njobs] = nd;
    IDE3D_CHECK_ARG(((size_t)n * max_cin + DEMOD_PARTS * DEMOD_COLS) * sizeof(float) <= 60 * 1024, "style_demod_batch: cin too large for LDS staging");
    hipLaunchKernelGGL(style_affine_batch_kernel, dim3(na, n), dim3(256), (size_t)wdim * sizeof(float), (hipStream_t)stream, a, wdim);
    if (d.njobs)
        hipLaunchKernelGGL(style_demod_batch_kernel, dim3(nd), dim3(1024), ((size_t)n * max_cin + DEMOD_PARTS * DEMOD_COLS) * sizeof(float),
                           (hipStream_t)stream, d, n);
    IDE3D_CHECK_LAUNCH("style_demod_batch");
    return IDE3D_OK;
}

extern "C" int ide3d_fold_heads_batch(const ide3d_fold_job* jobs, int32_t njobs, int32_t n, int32_t wdim, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(jobs && njobs > 0 && njobs <= IDE3D_STYLE_BATCH_MAX, "fold_heads_batch: 1..%d jobs", IDE3D_STYLE_BATCH_MAX);
    IDE3D_CHECK_ARG(n > 0 && wdim > 0, "fold_heads_batch: bad shape");
    const size_t lds = ((size_t)wdim + 2 * FOLD_CI) * sizeof(float);
    IDE3D_CHECK_ARG(lds <= 60 * 1024, "fold_heads_batch: w_dim too large for LDS staging");
    FoldBatch b{};
    int nb = 0;
    for (int j = 0; j < njobs; ++j) {
        const ide3d_fold_job& q = jobs[j];
        IDE3D_CHECK_ARG(q.w && q.a0 && q.w0 && q.a1 && q.w1 && q.out && q.cin > 0 && q.cout0 > 0 && q.cout1 > 0, "fold_heads_batch: job %d: null pointer / bad shape", j);
        b.job[j] = q; b.blk0[j] = nb; nb += cdiv(q.cin, FOLD_CI);
    }
    b.njobs = njobs; b.blk0[njobs] = nb;
    hipLaunchKernelGGL(fold_heads_batch_kernel, dim3(n, nb), dim3(1024), lds, (hipStream_t)stream, b, wdim);
    IDE3D_CHECK_LAUNCH("fold_heads_batch");
    return IDE3D_OK;
}
