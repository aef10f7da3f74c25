// composite.hip — depth-ordered alpha compositing of per-sample (features, sigma) along rays.
//
// Drop-in for `training.volumetric_rendering.fancy_integration`
// (training/volumetric_rendering.py:34-74):
//   delta_i = (z_{i+1} - z_i) * ||d_cam||,  delta_last = 1e10
//   alpha_i = 1 - exp(-delta_i * clamp(sigma_i + noise_i)),  clamp = softplus | relu
//   T_i     = prod_{j<i} (1 - alpha_j + 1e-10),  w_i = alpha_i * T_i
//   rgb = sum_i w_i c_i,  depth = sum_i w_i z_i  (+ last_back / white_back / max_depth / fill_mode)
//
// MI355X mapping: one 64-lane wavefront per ray, two phases.
//   1. lanes = depth samples: alpha per lane, exclusive prefix product of (1 - alpha + 1e-10) across
//      the wave with shuffles (Hillis-Steele, 6 steps), chunks of 64 samples carry the running
//      transmittance; weights go to LDS (and to the optional `weights` output).
//   2. lanes = channels: each lane walks the samples and accumulates w_i * c_i[lane]; every step is
//      one contiguous (ch+1)*4-byte row read, so the dominant stream (rgb_sigma) is read coalesced
//      and (through L1, same lines as phase 1's sigma reads) exactly once from HBM.
// HBM roofline: rays*steps*(ch+2)*4 bytes in, rays*(ch+1+steps)*4 out.
#include "common.h"
#include "knobs.h"
#include <stdlib.h>

namespace ide3d {

constexpr int kMaxSteps = 1024;    // LDS weights buffer per wave

__device__ __forceinline__ float softplus_f(float x) {
    // torch.nn.functional.softplus(beta=1, threshold=20)
    return (x > 20.f) ? x : log1pf(expf(x));
}

// Exclusive prefix product across the 64 lanes of a wave; returns (exclusive, total).
__device__ __forceinline__ float wave_excl_prod(float v, float& total) {
    const int lane = lane_id();
    float incl = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const float o = __shfl_up(incl, off);
        if (lane >= off) incl *= o;
    }
    total = __shfl(incl, kWave - 1);
    const float prev = __shfl_up(incl, 1);
    return lane == 0 ? 1.0f : prev;
}

__global__ void __launch_bounds__(256)
composite_kernel(const float* __restrict__ rgb_sigma, const float* __restrict__ z_vals,
                 const float* __restrict__ dir_norm, const float* __restrict__ noise,
                 int64_t rays, int steps, int ch, int clamp_mode, int last_back, int white_back,
                 float max_depth, int fill_mode,
                 float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ weights) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];   // [4 waves][steps]
    const int lane = lane_id();
    const int wid = threadIdx.x / kWave;
    float* sw = s_w + wid * steps;
    const int row = ch + 1;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t ray = (int64_t)blockIdx.x * 4 + wid; ray < rays; ray += nwaves) {
        const float* rs = rgb_sigma + ray * steps * row;
        const float* zv = z_vals + ray * steps;
        const float dn = dir_norm[ray];
        // ---- phase 1: weights ----
        float carry = 1.0f, wsum = 0.f, dsum = 0.f;
        for (int s0 = 0; s0 < steps; s0 += kWave) {
            const int s = s0 + lane;
            float alpha = 0.f, z = 0.f;
            if (s < steps) {
                z = zv[s];
                const float delta = (s + 1 < steps) ? (zv[s + 1] - z) * dn : 1e10f;
                float sg = rs[s * row + ch];
                if (noise) sg += noise[ray * steps + s];
                const float dens = clamp_mode == 0 ? softplus_f(sg) : fmaxf(sg, 0.f);
                alpha = 1.0f - expf(-delta * dens);
            }
            const float f = (s < steps) ? (1.0f - alpha + 1e-10f) : 1.0f;
            float tot;
            const float excl = wave_excl_prod(f, tot);
            const float w = alpha * (carry * excl);
            carry *= tot;
            if (s < steps) sw[s] = w;
            wsum += w; dsum += w * z;
        }
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
            wsum += __shfl_xor(wsum, off);
            dsum += __shfl_xor(dsum, off);
        }
        if (last_back) {
            // weights[:, :, -1] += 1 - weights_sum   (applied before the colour / depth sums)
            const float extra = 1.0f - wsum;
            if (lane == 0) sw[steps - 1] += extra;
            dsum += extra * zv[steps - 1];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (weights)
            for (int s = lane; s < steps; s += kWave) weights[ray * steps + s] = sw[s];
        // ---- phase 2: channels ----
        for (int c0 = 0; c0 < ch; c0 += kWave) {
            const int c = c0 + lane;
            float acc = 0.f;
            if (c < ch) {
                for (int s = 0; s < steps; ++s) acc += sw[s] * rs[s * row + c];
                if (white_back) acc = acc + 1.0f - wsum;
                if (fill_mode == 1 && wsum < 0.9f) acc = (c == 0) ? 1.0f : 0.0f;
                if (fill_mode == 2) acc = wsum;
                rgb[ray * ch + c] = acc;
            }
        }
        if (lane == 0 && depth) {
            float d = dsum;
            if (max_depth != 0.f) d += (1.0f - wsum) * max_depth;
            depth[ray] = d;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// LDS-staged variant: the (steps x (ch + 1)) block of a ray is contiguous in memory (20 KB at 96 x 53), so a wave copies it
// into LDS with 16-byte loads that are ALL issued before the first is consumed, and both phases then read LDS.  The kernel
// above touches the block twice from global memory (phase 1 reads sigma with a (ch + 1) * 4-byte lane stride — every lane its
// own 128-byte line — and phase 2 re-reads the rows after the working set of the CU's waves has long left L1 / L2): measured
// 191 us = 22 % of the HBM roofline at [4, 4096, 96, 52]; this one moves every byte once.
template <int NLD>     // 16-byte loads per lane: block floats <= NLD * 256
__global__ void __launch_bounds__(256)
composite_lds_kernel(const float* __restrict__ rgb_sigma, const float* __restrict__ z_vals,
                     const float* __restrict__ dir_norm, const float* __restrict__ noise,
                     int64_t rays, int steps, int ch, int clamp_mode, int last_back, int white_back,
                     float max_depth, int fill_mode,
                     float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ weights) {
    extern __shared__ __attribute__((aligned(16))) float s_all[];   // [4 waves][block + steps]
    const int lane = lane_id();
    const int wid = threadIdx.x / kWave;
    const int row = ch + 1;
    const int block = steps * row, nvec = block / 4;
    float* sb = s_all + (size_t)wid * (block + ((steps + 3) & ~3));
    float* sw = sb + block;
    const int wpb = blockDim.x / kWave;                       // waves per workgroup (1 or 4)
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    for (int64_t ray = (int64_t)blockIdx.x * wpb + wid; ray < rays; ray += nwaves) {
        const float4* src = reinterpret_cast<const float4*>(rgb_sigma + ray * block);
        float4 v[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) { const int i = lane + k * kWave; v[k] = src[min(i, nvec - 1)]; }
        const float* zv = z_vals + ray * steps;
        const float dn = dir_norm[ray];
#pragma unroll
        for (int k = 0; k < NLD; ++k) { const int i = lane + k * kWave; if (i < nvec) reinterpret_cast<float4*>(sb)[i] = v[k]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- phase 1: weights (lanes = samples; sigma column read with an odd-pitch stride: conflict-free when ch is even) ----
        float carry = 1.0f, wsum = 0.f, dsum = 0.f;
        for (int s0 = 0; s0 < steps; s0 += kWave) {
            const int s = s0 + lane;
            float alpha = 0.f, z = 0.f;
            if (s < steps) {
                z = zv[s];
                const float delta = (s + 1 < steps) ? (zv[s + 1] - z) * dn : 1e10f;
                float sg = sb[s * row + ch];
                if (noise) sg += noise[ray * steps + s];
                const float dens = clamp_mode == 0 ? softplus_f(sg) : fmaxf(sg, 0.f);
                alpha = 1.0f - expf(-delta * dens);
            }
            const float f = (s < steps) ? (1.0f - alpha + 1e-10f) : 1.0f;
            float tot;
            const float excl = wave_excl_prod(f, tot);
            const float w = alpha * (carry * excl);
            carry *= tot;
            if (s < steps) sw[s] = w;
            wsum += w; dsum += w * z;
        }
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
            wsum += __shfl_xor(wsum, off);
            dsum += __shfl_xor(dsum, off);
        }
        if (last_back) {
            const float extra = 1.0f - wsum;
            if (lane == 0) sw[steps - 1] += extra;
            dsum += extra * zv[steps - 1];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (weights)
            for (int s = lane; s < steps; s += kWave) weights[ray * steps + s] = sw[s];
        // ---- phase 2: channels (lanes = channels, rows from LDS: consecutive lanes on consecutive banks) ----
        for (int c0 = 0; c0 < ch; c0 += kWave) {
            const int c = c0 + lane;
            float acc = 0.f;
            if (c < ch) {
#pragma unroll 8
                for (int s = 0; s < steps; ++s) acc += sw[s] * sb[s * row + c];
                if (white_back) acc = acc + 1.0f - wsum;
                if (fill_mode == 1 && wsum < 0.9f) acc = (c == 0) ? 1.0f : 0.0f;
                if (fill_mode == 2) acc = wsum;
                rgb[ray * ch + c] = acc;
            }
        }
        if (lane == 0 && depth) {
            float d = dsum;
            if (max_depth != 0.f) d += (1.0f - wsum) * max_depth;
            depth[ray] = d;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace ide3d

extern "C" int ide3d_composite(const float* rgb_sigma, const float* z_vals, const float* dir_norm,
                               const float* noise, int64_t rays, int32_t steps, int32_t ch,
                               int clamp_mode, int last_back, int white_back, float max_depth,
                               int fill_mode, float* rgb, float* depth, float* weights, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(rgb_sigma && z_vals && dir_norm && rgb, "composite: null pointer");
    IDE3D_CHECK_ARG(rays >= 0 && steps >= 1 && steps <= kMaxSteps && ch >= 1, "composite: bad shape (steps <= %d)", kMaxSteps);
    IDE3D_CHECK_ARG(clamp_mode == 0 || clamp_mode == 1, "composite: Need to choose clamp mode");
    IDE3D_CHECK_ARG(fill_mode != 1 || ch == 3, "composite: fill_mode 'debug' needs 3 colour channels");
    if (rays == 0) return IDE3D_OK;
    const int grid = stream_grid(rays, 4);
    {
        // LDS-staged path: contiguous, 16-byte aligned ray blocks that fit four to a workgroup
        const int64_t block = (int64_t)steps * (ch + 1);
        const size_t lds = (size_t)4 * (block + ((steps + 3) & ~3)) * sizeof(float);
        const bool off = knobs().composite_no_lds;
        if (!off && block % 4 == 0 && block <= 40 * 256 && lds <= 160 * 1024 && ((reinterpret_cast<uintptr_t>(rgb_sigma) & 15) == 0)) {
            const int nld = (int)((block / 4 + kWave - 1) / kWave);
            // persistent workgroups (as many as fit the LDS of the chip): every wave walks several rays, which is what the
            // request-ahead pipeline needs
            // one-wave workgroups: the LDS footprint is per wave (block + weights), so 64-thread workgroups pack the CU's 160 KB
            // with as many waves as fit (7 at 96 x 53) instead of one 4-wave workgroup
            const int mode = knobs().composite_mode;
            const int wpb = (mode == 1 || mode == 3) ? 4 : 1;
            const size_t lds_w = lds / 4 * wpb;
            int64_t pgrid = (mode >= 2) ? (int64_t)kNumCU * ((160 * 1024) / (int64_t)lds_w) : cdiv64(rays, wpb);
            if (pgrid > cdiv64(rays, wpb)) pgrid = cdiv64(rays, wpb);
            if (pgrid > 0x7fffffff) pgrid = 0x7fffffff;
            auto go = [&](auto kern) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w);
                hipLaunchKernelGGL(kern, dim3((unsigned)pgrid), dim3(64 * wpb), lds_w, (hipStream_t)stream,
                                   rgb_sigma, z_vals, dir_norm, noise, rays, steps, ch, clamp_mode, last_back, white_back,
                                   max_depth, fill_mode, rgb, depth, weights);
            };
            if (nld <= 8) go(composite_lds_kernel<8>); else if (nld <= 20) go(composite_lds_kernel<20>); else go(composite_lds_kernel<40>);
            IDE3D_CHECK_LAUNCH("composite");
            return IDE3D_OK;
        }
    }
    hipLaunchKernelGGL(composite_kernel, dim3(grid), dim3(256), (size_t)4 * steps * sizeof(float), (hipStream_t)stream,
                       rgb_sigma, z_vals, dir_norm, noise, rays, steps, ch, clamp_mode, last_back, white_back,
                       max_depth, fill_mode, rgb, depth, weights);
    IDE3D_CHECK_LAUNCH("composite");
    return IDE3D_OK;
}
