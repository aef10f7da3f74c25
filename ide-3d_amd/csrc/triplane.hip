// triplane.hip — tri-plane bilinear feature gather (and its backward) for gfx950.
//
// Drop-in for `dnnlib.util.sample_from_triplane` (dnnlib/util.py:580-617): three
// grid_sample(bilinear, zeros, align_corners=False) look-ups on the xy / yz / xz planes of a
// [n, 3C, H, W] tensor, summed.  Plane p, axes (a -> W, b -> H): p0 = (x, y), p1 = (y, z),
// p2 = (x, z).  Output row index = n * m + sample.
//
// MI355X mapping (fast path, channels_last planes, C % 4 == 0):
//   * one bilinear tap = one contiguous C*4-byte read (128 B at C = 32) instead of C scattered
//     4-byte reads: C/4 lanes cooperate on a sample with one global_load_dwordx4 each, so a
//     wavefront retires 64/(C/4) samples per tap instruction and every fetched line is fully used;
//   * consecutive samples are consecutive depth steps of one ray, so the 12 taps of a wave's
//     samples hit neighbouring lines (L1/L2 reuse); workgroups take *contiguous* sample chunks and
//     the chunk order is XCD-remapped so that neighbouring rays share one XCD's L2;
//   * output rows are written as 16-byte vectors, fully coalesced (C*4 bytes per sample).
// HBM roofline: (3*C*H*W + 3*m + C*m) * 4 bytes per image (planes + coords + output).
//
// Index math is bit-exact w.r.t. ATen grid_sampler_unnormalize (align_corners=False):
//   u = ((c + 1) * size - 1) / 2 evaluated add, mul, sub, mul(0.5) in fp32 without contraction.
#include "common.h"
#include "knobs.h"
#include "triplane_tap.h"

namespace ide3d {

// Fast path: channels_last planes (stride of the channel axis == 1), C % 4 == 0, (C/4) | 64.
template <int LPS>   // lanes per sample = C / 4
__global__ void __launch_bounds__(256)
triplane_sample_cl_kernel(const float* __restrict__ planes, int64_t sN, int sH, int sW,
                          int C, int H, int W, const float* __restrict__ coords, int64_t m,
                          int64_t rows, float* __restrict__ out, int64_t rows_per_block) {
    constexpr int SPW = kWave / LPS;                     // samples per wave-iteration
    constexpr int SPB = SPW * 4;                         // samples per workgroup-iteration (4 waves)
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row_begin = (int64_t)blk * rows_per_block;
    int64_t row_end = row_begin + rows_per_block;
    if (row_end > rows) row_end = rows;
    if (row_begin >= row_end) return;
    const int sub = threadIdx.x / LPS;                   // sample slot inside the workgroup
    const int cl = threadIdx.x % LPS;                    // 4-channel slice
    // image index of the current row, tracked incrementally (no 64-bit division in the loop)
    int64_t n = row_begin / m;
    int64_t n_end = (n + 1) * m;                         // first row of image n + 1
    const float* pb = planes + n * sN + cl * 4;
    // software-pipelined coordinate fetch: the next iteration's (x, y, z) is in flight during the gathers
    int64_t row = row_begin + sub;
    const int64_t last = row_end - 1;
    const float* cp = coords + (row < last ? row : last) * 3;
    float cx = cp[0], cy = cp[1], cz = cp[2];
    for (; row < row_end; row += SPB) {
        const int64_t nxt = row + SPB;
        const float* np_ = coords + (nxt < last ? nxt : last) * 3;
        const float ncx = np_[0], ncy = np_[1], ncz = np_[2];
        while (row >= n_end) { n_end += m; pb += sN; }
        const TapAddr t0 = make_tap_addr(cx, cy, W, H, sH, sW);
        const TapAddr t1 = make_tap_addr(cy, cz, W, H, sH, sW);
        const TapAddr t2 = make_tap_addr(cx, cz, W, H, sH, sW);
        const float4 a0 = gather_plane_cl(pb, t0);
        const float4 a1 = gather_plane_cl(pb + C, t1);
        const float4 a2 = gather_plane_cl(pb + 2 * C, t2);
        float4 r;
        r.x = (a0.x + a1.x) + a2.x; r.y = (a0.y + a1.y) + a2.y;
        r.z = (a0.z + a1.z) + a2.z; r.w = (a0.w + a1.w) + a2.w;
        *reinterpret_cast<float4*>(__builtin_assume_aligned(out + row * C + cl * 4, 16)) = r;
        cx = ncx; cy = ncy; cz = ncz;
    }
}

// ------------------------------------------------------------------------------------------------
// v2 fast path: tap set-up once per sample, shared through LDS; 32-bit buffer addressing.
//
// In the kernel above the C/4 lanes of a sample each redo the (identical) coordinate -> tap arithmetic and 64-bit
// address math, which makes it VALU-bound (~350 VALU per 8 samples).  Here a wavefront takes 64 consecutive samples:
//   phase A  lane = sample: one coalesced 12-byte coordinate load, three taps, results (4 byte-offsets + 4 masked
//            weights per plane = 24 dwords) parked in LDS, [sample][24] (conflict-free for the phase-B broadcast reads);
//   phase B  lane = (sample slot, 4-channel slice): 64 / SPW rounds; per round six ds_read_b128 fetch the sample's
//            taps, twelve buffer_load_dwordx4 (SGPR base + 32-bit offset, no 64-bit VALU address math) fetch the
//            C*4-byte tap lines, 24 packed FMAs blend them, one 16-byte store writes the output slice.
// ------------------------------------------------------------------------------------------------

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    const f32x4_t f = __builtin_bit_cast(f32x4_t, v);
    return make_float4(f.x, f.y, f.z, f.w);
}

template <int LPS>
__global__ void __launch_bounds__(256)
triplane_sample_cl2_kernel(const float* __restrict__ planes, unsigned sN_bytes, int sH, int sW, unsigned plane_bytes,
                           int C, int H, int W, const float* __restrict__ coords, unsigned m,
                           unsigned rows, float* __restrict__ out, unsigned rows_per_block) {
    constexpr int SPW = kWave / LPS;                     // samples per phase-B round
    constexpr int ROUNDS = kWave / SPW;                  // = LPS
    __shared__ __attribute__((aligned(16))) unsigned s_tap[4][kWave][24];
    const int lane = lane_id(), wid = threadIdx.x >> 6;
    const int slot = lane / LPS, cl = lane % LPS;
    unsigned (*tap)[24] = s_tap[wid];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)planes, 0, (int)plane_bytes, 0x00020000);

    const unsigned blk = (unsigned)xcd_remap(blockIdx.x, gridDim.x);
    const unsigned row_begin = blk * rows_per_block;
    unsigned row_end = row_begin + rows_per_block;
    if (row_end > rows) row_end = rows;
    const unsigned ch_bytes = (unsigned)cl * 16u;

    for (unsigned base = row_begin + wid * kWave; base < row_end; base += 4 * kWave) {
        // ---- phase A ----
        {
            const unsigned row = min(base + lane, rows - 1);
            const float* cp = coords + (size_t)row * 3;
            const float cx = cp[0], cy = cp[1], cz = cp[2];
            const unsigned img = (row / m) * sN_bytes;
            const TapAddr t0 = make_tap_addr(cx, cy, W, H, sH, sW);
            const TapAddr t1 = make_tap_addr(cy, cz, W, H, sH, sW);
            const TapAddr t2 = make_tap_addr(cx, cz, W, H, sH, sW);
            const unsigned p1 = img + (unsigned)C * 4u, p2 = img + (unsigned)C * 8u;
            u32x4* dst = reinterpret_cast<u32x4*>(tap[lane]);
            dst[0] = u32x4{img + (unsigned)t0.o00 * 4u, img + (unsigned)t0.o01 * 4u, img + (unsigned)t0.o10 * 4u, img + (unsigned)t0.o11 * 4u};
            dst[1] = u32x4{__float_as_uint(t0.w00), __float_as_uint(t0.w01), __float_as_uint(t0.w10), __float_as_uint(t0.w11)};
            dst[2] = u32x4{p1 + (unsigned)t1.o00 * 4u, p1 + (unsigned)t1.o01 * 4u, p1 + (unsigned)t1.o10 * 4u, p1 + (unsigned)t1.o11 * 4u};
            dst[3] = u32x4{__float_as_uint(t1.w00), __float_as_uint(t1.w01), __float_as_uint(t1.w10), __float_as_uint(t1.w11)};
            dst[4] = u32x4{p2 + (unsigned)t2.o00 * 4u, p2 + (unsigned)t2.o01 * 4u, p2 + (unsigned)t2.o10 * 4u, p2 + (unsigned)t2.o11 * 4u};
            dst[5] = u32x4{__float_as_uint(t2.w00), __float_as_uint(t2.w01), __float_as_uint(t2.w10), __float_as_uint(t2.w11)};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- phase B ----
#pragma unroll 2
        for (int r = 0; r < ROUNDS; ++r) {
            const int s = r * SPW + slot;
            const unsigned row = base + s;
            const u32x4* src = reinterpret_cast<const u32x4*>(tap[s]);
            float4 acc[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const u32x4 o = src[2 * pl];
                const u32x4 wq = src[2 * pl + 1];
                const float4 v00 = buf_ld4(rsrc, o.x + ch_bytes);
                const float4 v01 = buf_ld4(rsrc, o.y + ch_bytes);
                const float4 v10 = buf_ld4(rsrc, o.z + ch_bytes);
                const float4 v11 = buf_ld4(rsrc, o.w + ch_bytes);
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                a = f4_fma(v00, __uint_as_float(wq.x), a);
                a = f4_fma(v01, __uint_as_float(wq.y), a);
                a = f4_fma(v10, __uint_as_float(wq.z), a);
                a = f4_fma(v11, __uint_as_float(wq.w), a);
                acc[pl] = a;
            }
            if (row < row_end) {
                f32x4_t res = {(acc[0].x + acc[1].x) + acc[2].x, (acc[0].y + acc[1].y) + acc[2].y,
                               (acc[0].z + acc[1].z) + acc[2].z, (acc[0].w + acc[1].w) + acc[2].w};
                __builtin_nontemporal_store(res, reinterpret_cast<f32x4_t*>(out + (size_t)row * C + cl * 4));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// General strided path (NCHW or anything else): one lane per sample, loop over channels; the
// output tile of the workgroup is transposed through LDS so that stores stay coalesced.
__global__ void __launch_bounds__(256)
triplane_sample_strided_kernel(const float* __restrict__ planes, int64_t sN, int64_t sC, int64_t sH, int64_t sW,
                               int C, int H, int W, const float* __restrict__ coords, int64_t m,
                               int64_t rows, float* __restrict__ out) {
    __shared__ float s_out[256 * 33];                    // [sample][channel chunk of 32] padded
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t row0 = (int64_t)blockIdx.x * 256; row0 < rows; row0 += stride) {
        const int64_t row = row0 + threadIdx.x;
        const bool live = row < rows;
        Tap2 t[3];
        const float* pb = planes;
        if (live) {
            const float* cp = coords + row * 3;
            const float cx = cp[0], cy = cp[1], cz = cp[2];
            t[0] = make_tap(cx, cy, W, H); t[1] = make_tap(cy, cz, W, H); t[2] = make_tap(cx, cz, W, H);
            pb = planes + (row / m) * sN;
        }
        for (int c0 = 0; c0 < C; c0 += 32) {
            const int cn = (C - c0 < 32) ? C - c0 : 32;
            if (live) {
                for (int c = 0; c < cn; ++c) {
                    float tot[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const float* p00 = pb + ((int64_t)pl * C + c0 + c) * sC + t[pl].iy0 * sH + t[pl].ix0 * sW;
                        float acc = 0.f;
                        if (t[pl].mask & 1u) acc += p00[0] * t[pl].w00;
                        if (t[pl].mask & 2u) acc += p00[sW] * t[pl].w01;
                        if (t[pl].mask & 4u) acc += p00[sH] * t[pl].w10;
                        if (t[pl].mask & 8u) acc += p00[sH + sW] * t[pl].w11;
                        tot[pl] = acc;
                    }
                    s_out[threadIdx.x * 33 + c] = (tot[0] + tot[1]) + tot[2];
                }
            }
            __syncthreads();
            // Coalesced write-out: consecutive lanes -> consecutive channels of a sample.
            for (int i = threadIdx.x; i < 256 * cn; i += 256) {
                const int s = i / cn, c = i - s * cn;
                if (row0 + s < rows) out[(row0 + s) * C + c0 + c] = s_out[s * 33 + c];
            }
            __syncthreads();
        }
    }
}

__global__ void __launch_bounds__(256)
triplane_taps_kernel(int H, int W, const float* __restrict__ coords, int64_t rows, int32_t* __restrict__ taps) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += stride) {
        const float* cp = coords + row * 3;
        const float cx = cp[0], cy = cp[1], cz = cp[2];
        const Tap2 t0 = make_tap(cx, cy, W, H), t1 = make_tap(cy, cz, W, H), t2 = make_tap(cx, cz, W, H);
        int32_t* o = taps + row * 9;
        o[0] = t0.ix0; o[1] = t0.iy0; o[2] = (int)t0.mask;
        o[3] = t1.ix0; o[4] = t1.iy0; o[5] = (int)t1.mask;
        o[6] = t2.ix0; o[7] = t2.iy0; o[8] = (int)t2.mask;
    }
}

// Backward: scatter-add of grad_out into grad_planes (+ optional coordinate gradients).
__global__ void __launch_bounds__(256)
triplane_backward_kernel(const float* __restrict__ grad_out, const float* __restrict__ planes,
                         int64_t sN, int64_t sC, int64_t sH, int64_t sW,
                         int C, int H, int W, const float* __restrict__ coords, int64_t m, int64_t rows,
                         float* __restrict__ grad_planes, int64_t gN, int64_t gC, int64_t gH, int64_t gW,
                         float* __restrict__ grad_coords) {
    // One wave per sample-group: lanes run over channels so atomics on channels_last gradients coalesce.
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) / kWave;
    for (int64_t row = wave; row < rows; row += nwaves) {
        const float* cp = coords + row * 3;
        const float cx = cp[0], cy = cp[1], cz = cp[2];
        const int64_t n = row / m;
        Tap2 t[3] = { make_tap(cx, cy, W, H), make_tap(cy, cz, W, H), make_tap(cx, cz, W, H) };
        float gu[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f};   // d/du, d/dv per plane (partial over lanes)
        for (int c = lane; c < C; c += kWave) {
            const float go = grad_out[row * C + c];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const int64_t ch = (int64_t)pl * C + c;
                float* g00 = grad_planes + n * gN + ch * gC + t[pl].iy0 * gH + t[pl].ix0 * gW;
                if (t[pl].mask & 1u) atomicAdd(g00, go * t[pl].w00);
                if (t[pl].mask & 2u) atomicAdd(g00 + gW, go * t[pl].w01);
                if (t[pl].mask & 4u) atomicAdd(g00 + gH, go * t[pl].w10);
                if (t[pl].mask & 8u) atomicAdd(g00 + gH + gW, go * t[pl].w11);
                if (grad_coords) {
                    const float* p00 = planes + n * sN + ch * sC + t[pl].iy0 * sH + t[pl].ix0 * sW;
                    const float v00 = (t[pl].mask & 1u) ? p00[0] : 0.f;
                    const float v01 = (t[pl].mask & 2u) ? p00[sW] : 0.f;
                    const float v10 = (t[pl].mask & 4u) ? p00[sH] : 0.f;
                    const float v11 = (t[pl].mask & 8u) ? p00[sH + sW] : 0.f;
                    // weights: w00 = ax*ay, w01 = bx*ay, w10 = ax*by, w11 = bx*by with ax = x1-u, bx = u-x0.
                    const float u = unnormalize(pl == 1 ? cy : cx, W), v = unnormalize(pl == 0 ? cy : cz, H);
                    const float bx = u - floorf(u), ax = 1.f - bx, by = v - floorf(v), ay = 1.f - by;
                    gu[pl] += go * ((v01 - v00) * ay + (v11 - v10) * by);
                    gv[pl] += go * ((v10 - v00) * ax + (v11 - v01) * bx);
                }
            }
        }
        if (grad_coords) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                for (int off = kWave / 2; off > 0; off >>= 1) {
                    gu[pl] += __shfl_xor(gu[pl], off);
                    gv[pl] += __shfl_xor(gv[pl], off);
                }
            if (lane == 0) {
                // du/dc = size / 2 (align_corners=False).  x feeds planes 0,2 (u); y feeds plane 0 (v) and
                // plane 1 (u); z feeds planes 1,2 (v).
                const float hx = 0.5f * W, hy = 0.5f * H;
                grad_coords[row * 3 + 0] = (gu[0] + gu[2]) * hx;
                grad_coords[row * 3 + 1] = gv[0] * hy + gu[1] * hx;
                grad_coords[row * 3 + 2] = (gv[1] + gv[2]) * hy;
            }
        }
    }
}

// v2 launcher: images are processed in groups whose planes span < 2 GiB so that byte offsets fit 32 bits.
template <int LPS>
static bool launch_cl2(const float* planes, const int64_t* s, int n, int C, int H, int W,
                       const float* coords, int64_t m, float* out, hipStream_t st) {
    const int64_t sN_bytes = s[0] * 4;
    if (m <= 0 || m >= (1LL << 31) || sN_bytes <= 0 || sN_bytes >= (1LL << 31)) return false;
    if ((int64_t)H * s[2] * 4 > sN_bytes || s[0] < 0) return false;
    int group = (int)((((1LL << 31) - 1) / sN_bytes));
    if (group < 1) return false;
    while ((int64_t)group * m >= (1LL << 31)) group >>= 1;
    if (group < 1) return false;
    for (int n0 = 0; n0 < n; n0 += group) {
        const int cnt = (n - n0 < group) ? n - n0 : group;
        const unsigned rows = (unsigned)((int64_t)cnt * m);
        int64_t nblk = kNumCU * 6;
        unsigned rpb = (unsigned)(cdiv64(cdiv64(rows, nblk), 256) * 256);
        if (rpb < 256) rpb = 256;
        nblk = cdiv64(rows, rpb);
        hipLaunchKernelGGL((triplane_sample_cl2_kernel<LPS>), dim3((unsigned)nblk), dim3(256), 0, st,
                           planes + (int64_t)n0 * s[0], (unsigned)sN_bytes, (int)s[2], (int)s[3], (unsigned)(cnt * sN_bytes),
                           C, H, W, coords + (int64_t)n0 * m * 3, (unsigned)m, rows, out + (int64_t)n0 * m * C, rpb);
    }
    return true;
}

template <int LPS>
static void launch_cl(const float* planes, const int64_t* s, int n, int C, int H, int W,
                      const float* coords, int64_t m, float* out, hipStream_t st) {
    if (launch_cl2<LPS>(planes, s, n, C, H, W, coords, m, out, st)) return;
    const int64_t rows = (int64_t)n * m;
    constexpr int SPB = (kWave / LPS) * 4;
    // ~8 workgroups per CU; contiguous chunks, multiple of the per-iteration sample count.
    int64_t nblk = kNumCU * 8;
    int64_t rpb = cdiv64(cdiv64(rows, nblk), SPB) * SPB;
    if (rpb < SPB) rpb = SPB;
    nblk = cdiv64(rows, rpb);
    hipLaunchKernelGGL((triplane_sample_cl_kernel<LPS>), dim3((unsigned)nblk), dim3(256), 0, st,
                       planes, s[0], (int)s[2], (int)s[3], C, H, W, coords, m, rows, out, rpb);
}

}  // namespace ide3d

extern "C" int ide3d_triplane_sample(const float* planes, const int64_t plane_stride[4],
                                     int32_t n, int32_t C, int32_t H, int32_t W,
                                     const float* coords, int64_t m, float* out, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(planes && coords && out && plane_stride, "triplane_sample: null pointer");
    IDE3D_CHECK_ARG(n > 0 && C > 0 && H > 0 && W > 0 && m >= 0, "triplane_sample: bad shape");
    if (m == 0) return IDE3D_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t* s = plane_stride;
    const bool cl = (s[1] == 1) && (C % 4 == 0) && (s[0] % 4 == 0) && (s[2] % 4 == 0) && (s[3] % 4 == 0) &&
                    (s[2] > 0 && s[3] > 0 && s[2] * H < 0x7fffffffLL && s[3] * W < 0x7fffffffLL) &&
                    ((reinterpret_cast<uintptr_t>(planes) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    const int lps = C / 4;
    if (cl && lps >= 1 && lps <= 64 && (64 % lps) == 0) {
        switch (lps) {
        case 1:  launch_cl<1>(planes, s, n, C, H, W, coords, m, out, st); break;
        case 2:  launch_cl<2>(planes, s, n, C, H, W, coords, m, out, st); break;
        case 4:  launch_cl<4>(planes, s, n, C, H, W, coords, m, out, st); break;
        case 8:  launch_cl<8>(planes, s, n, C, H, W, coords, m, out, st); break;
        case 16: launch_cl<16>(planes, s, n, C, H, W, coords, m, out, st); break;
        case 32: launch_cl<32>(planes, s, n, C, H, W, coords, m, out, st); break;
        case 64: launch_cl<64>(planes, s, n, C, H, W, coords, m, out, st); break;
        }
    } else {
        const int64_t rows = (int64_t)n * m;
        hipLaunchKernelGGL(triplane_sample_strided_kernel, dim3(stream_grid(rows, 256)), dim3(256), 0, st,
                           planes, s[0], s[1], s[2], s[3], C, H, W, coords, m, rows, out);
    }
    IDE3D_CHECK_LAUNCH("triplane_sample");
    return IDE3D_OK;
}

namespace ide3d {
bool launch_triplane_tile(const float* planes, const int64_t* s, int n, int C, int H, int W, const float* coords, int64_t m,
                          float* out, int rays_h, int rays_w, int steps, hipStream_t st);   // triplane_tile.hip
}

extern "C" int ide3d_triplane_sample_rays(const float* planes, const int64_t plane_stride[4],
                                          int32_t n, int32_t C, int32_t H, int32_t W,
                                          const float* coords, int64_t m, float* out,
                                          int32_t rays_h, int32_t rays_w, int32_t steps, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(planes && coords && out && plane_stride, "triplane_sample_rays: null pointer");
    IDE3D_CHECK_ARG(n > 0 && C > 0 && H > 0 && W > 0 && m >= 0, "triplane_sample_rays: bad shape");
    IDE3D_CHECK_ARG(rays_h > 0 && rays_w > 0 && steps > 0 && (int64_t)rays_h * rays_w * steps == m,
                    "triplane_sample_rays: m must equal rays_h * rays_w * steps");
    const bool no_tile = knobs().gather_no_tile;
    const bool aligned = ((reinterpret_cast<uintptr_t>(planes) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                         (plane_stride[0] % 4 == 0) && (plane_stride[2] % 4 == 0) && (plane_stride[3] % 4 == 0);
    if (!no_tile && aligned &&
        launch_triplane_tile(planes, plane_stride, n, C, H, W, coords, m, out, rays_h, rays_w, steps, (hipStream_t)stream)) {
        IDE3D_CHECK_LAUNCH("triplane_sample_rays");
        return IDE3D_OK;
    }
    return ide3d_triplane_sample(planes, plane_stride, n, C, H, W, coords, m, out, stream);
}

extern "C" int ide3d_triplane_taps(int32_t H, int32_t W, const float* coords, int64_t rows,
                                   int32_t* taps, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(coords && taps && H > 0 && W > 0 && rows >= 0, "triplane_taps: bad argument");
    if (rows == 0) return IDE3D_OK;
    hipLaunchKernelGGL(triplane_taps_kernel, dim3(stream_grid(rows, 256)), dim3(256), 0, (hipStream_t)stream,
                       H, W, coords, rows, taps);
    IDE3D_CHECK_LAUNCH("triplane_taps");
    return IDE3D_OK;
}

extern "C" int ide3d_triplane_sample_backward(const float* grad_out, const float* planes,
                                              const int64_t plane_stride[4],
                                              int32_t n, int32_t C, int32_t H, int32_t W,
                                              const float* coords, int64_t m,
                                              float* grad_planes, const int64_t grad_plane_stride[4],
                                              float* grad_coords, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(grad_out && planes && coords && grad_planes && plane_stride && grad_plane_stride,
                    "triplane_sample_backward: null pointer");
    IDE3D_CHECK_ARG(n > 0 && C > 0 && H > 0 && W > 0 && m >= 0, "triplane_sample_backward: bad shape");
    if (m == 0) return IDE3D_OK;
    const int64_t rows = (int64_t)n * m;
    const int64_t* s = plane_stride; const int64_t* g = grad_plane_stride;
    hipLaunchKernelGGL(triplane_backward_kernel, dim3(stream_grid(rows * kWave, 256)), dim3(256), 0, (hipStream_t)stream,
                       grad_out, planes, s[0], s[1], s[2], s[3], C, H, W, coords, m, rows,
                       grad_planes, g[0], g[1], g[2], g[3], grad_coords);
    IDE3D_CHECK_LAUNCH("triplane_sample_backward");
    return IDE3D_OK;
}
