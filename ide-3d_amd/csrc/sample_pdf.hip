// sample_pdf.hip — inverse-CDF depth resampling of the hierarchical pass (training/volumetric_rendering.py:224-265).
//
//   pdf = (w + eps) / sum(w + eps);  cdf = [0, cumsum(pdf)];  i = searchsorted_left(cdf, u)
//   lo = max(i - 1, 0), hi = min(i, K);  d = cdf[hi] - cdf[lo] (d < eps -> 1)
//   sample = bins[lo] + (u - cdf[lo]) / d * (bins[hi] - bins[lo])
//
// One wave per ray: the K weights are split into 64 contiguous runs (lane = run), the row sum and the prefix sums are
// accumulated in double and rounded to float per element (what ATen's CPU `sum` / `cumsum` produce up to the last ulp),
// the K + 1 cdf values go to LDS, and each lane then resolves its draws u_j with a binary search.  All the float
// arithmetic after the cdf is the reference's expression, operation by operation, without contraction.
// A ray moves (2K + 1 + 2 n_importance) * 4 bytes; at the benchmark shape (16384 rays, K = 94, 96 draws) that is 25 MB
// and the launch is latency-, not bandwidth-bound (a few microseconds).
#include "common.h"

namespace ide3d {
namespace {

constexpr int kPdfWaves = 4;          // rays per workgroup and step
constexpr int kPdfMaxK = 2048;        // LDS: 4 * 2049 floats = 32.8 KB

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

__global__ void __launch_bounds__(kPdfWaves * kWave)
sample_pdf_kernel(const float* __restrict__ bins, const float* __restrict__ weights, const float* __restrict__ u,
                  int64_t u_ray_stride, int64_t rays, int k, int n_imp, float eps, float* __restrict__ samples) {
    extern __shared__ float s_cdf[];                       // [kPdfWaves][k + 1]
    const int lane = lane_id(), wid = threadIdx.x / kWave;
    float* cdf = s_cdf + (size_t)wid * (k + 1);
    const int per = (k + kWave - 1) / kWave;               // run length of a lane
    const int i0 = min(lane * per, k), i1 = min(i0 + per, k);

    for (int64_t base = (int64_t)blockIdx.x * kPdfWaves; base < rays; base += (int64_t)gridDim.x * kPdfWaves) {
        const int64_t ray = base + wid;
        const bool live = ray < rays;
        if (live) {
            const float* w = weights + ray * k;
            double part = 0.0;
            for (int i = i0; i < i1; ++i) part += (double)__fadd_rn(w[i], eps);
            const float total = (float)wave_sum(part);
            // exclusive prefix of the lane totals of pdf (in double), then the run itself
            double run = 0.0;
            for (int i = i0; i < i1; ++i) run += (double)__fdiv_rn(__fadd_rn(w[i], eps), total);
            double incl = run;
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const double up = __shfl_up(incl, o, kWave);
                if (lane >= o) incl += up;
            }
            double acc = incl - run;
            for (int i = i0; i < i1; ++i) {
                acc += (double)__fdiv_rn(__fadd_rn(w[i], eps), total);
                cdf[i + 1] = (float)acc;
            }
            if (lane == 0) cdf[0] = 0.f;
        }
        __syncthreads();
        if (live) {
            const float* b = bins + ray * (k + 1);
            const float* ur = u + ray * u_ray_stride;
            float* o = samples + ray * n_imp;
            for (int j = lane; j < n_imp; j += kWave) {
                const float uj = ur[j];
                int lo = 0, hi = k + 1;                    // first index with cdf[index] >= u  (torch.searchsorted, right=False)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (cdf[mid] < uj) lo = mid + 1; else hi = mid;
                }
                const int below = max(lo - 1, 0), above = min(lo, k);
                const float c0 = cdf[below], c1 = cdf[above];
                float d = __fsub_rn(c1, c0);
                d = d < eps ? 1.f : d;
                const float t = __fdiv_rn(__fsub_rn(uj, c0), d);
                o[j] = __fadd_rn(b[below], __fmul_rn(t, __fsub_rn(b[above], b[below])));
            }
        }
        __syncthreads();                                   // cdf is rewritten by the next ray of this wave
    }
}

}  // namespace
}  // namespace ide3d

extern "C" int ide3d_sample_pdf(const float* bins, const float* weights, const float* u, int64_t u_ray_stride,
                                int64_t rays, int32_t k, int32_t n_importance, float eps, float* samples, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(rays >= 0 && k >= 1 && k <= kPdfMaxK && n_importance >= 1, "sample_pdf: bad shape (1 <= bins - 1 <= %d)", kPdfMaxK);
    IDE3D_CHECK_ARG(u_ray_stride == 0 || u_ray_stride >= n_importance, "sample_pdf: u rows overlap");
    if (rays == 0) return IDE3D_OK;
    IDE3D_CHECK_ARG(bins && weights && u && samples, "sample_pdf: null pointer");
    const int grid = stream_grid(rays, kPdfWaves);
    hipLaunchKernelGGL(sample_pdf_kernel, dim3(grid), dim3(kPdfWaves * kWave), (size_t)kPdfWaves * (k + 1) * sizeof(float),
                       (hipStream_t)stream, bins, weights, u, u_ray_stride, rays, (int)k, (int)n_importance, eps, samples);
    IDE3D_CHECK_LAUNCH("sample_pdf");
    return IDE3D_OK;
}
