// modconv.hip — StyleGAN2 modulated convolution as an fp32 MFMA implicit GEMM with fused epilogue.
//
// Replaces, for stride-1 k x k (k in {1, 3}) layers, the ATen conv behind
// `conv2d_gradfix.conv2d` (torch_utils/ops/conv2d_gradfix.py:35) as used by `modulated_conv2d`
// (inversion/networks.py:55-130) together with the epilogue of `SynthesisLayer.forward`
// (networks.py:457-512: `+ noise`, `bias_act(lrelu, gain, clamp)`) and of `ToRGBLayer.forward`
// (networks.py:700-707: no demodulation, linear, clamp):
//   y[n,o,p] = act( d[n,o] * sum_{i,t} w[o,i,t] * s[n,i] * x[n,i,p+t] + ns*noise[p] + b[o] ) * gain
// (the un-fused formulation of networks.py:99-114, which needs no per-sample weight tensor).
//
// GEMM view per image: M = cout, N = pixels, K = cin*k*k on v_mfma_f32_32x32x2_f32 (fp32 in, fp32
// accumulate, exact fp32 products -> same numerics class as the reference's fp32 conv).
//   * workgroup = 4 waves; output tile BM x 128 pixels (an 8 x 16 pixel patch); two shapes:
//       BIG   BM = 128: waves 2(M) x 2(N), each 64 couts x 64 pixels  (4 accumulators of 32x32)
//       SMALL BM = 32 : waves 1(M) x 4(N), each 32 couts x 32 pixels  (toRGB / toSeg heads)
//   * K is walked in chunks of KC input channels: the chunk's weights are staged in LDS already
//     multiplied by the styles (A operand, [k][cout] with a +1 pad -> conflict-free both ways), the
//     chunk's input halo patch ((8+k-1) x (16+k-1) per channel, zero padded) is staged once and read
//     k*k times with shifted addresses (B operand) — im2col never materialises;
//   * epilogue in registers: demodulation, noise, bias, lrelu, gain, clamp, then NCHW stores.
// The fp32 MFMA rate (157 TFLOP/s peak) bounds this kernel; LDS/L2 traffic is ~5 B/clk/CU.
#include "common.h"

namespace ide3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KS, int BIG>
struct McCfg {
    static constexpr int PH = 8, PW = 16;                 // pixel patch (BN = 128)
    static constexpr int WM = BIG ? 2 : 1;                // waves along M
    static constexpr int WN = 4 / WM;                     // waves along N
    static constexpr int MTW = BIG ? 2 : 1;               // 32-row M tiles per wave
    static constexpr int NTW = (PH * PW / 32) / WN;       // 32-pixel N tiles per wave
    static constexpr int BM = WM * MTW * 32;
    static constexpr int KC = (KS == 3) ? 4 : 16;         // input channels per K chunk
    static constexpr int TAPS = KS * KS;
    static constexpr int KK = KC * TAPS;                  // K elements per chunk (even)
    static constexpr int HP = PH + KS - 1, HW = PW + KS - 1;
    static constexpr int XW = HW + 2;                     // LDS row pitch of the halo patch
    static constexpr int XS = HP * XW;                    // per-channel pitch
    static constexpr int WP = BM + 1;                     // LDS pitch of a weight k-row
    static constexpr int LDS_W = KK * WP;
    static constexpr int LDS_X = KC * XS;
};

template <int KS, int BIG>
__global__ void __launch_bounds__(256, 2)
modconv_kernel(ide3d_modconv_params p, int tiles_x, int tiles_y, int mblocks) {
    using K = McCfg<KS, BIG>;
    __shared__ float s_w[K::LDS_W];
    __shared__ float s_x[K::LDS_X];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / K::WN, wn = wid % K::WN;
    const int half = lane >> 5, l32 = lane & 31;

    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mb = bid % mblocks; bid /= mblocks;
    const int txi = bid % tiles_x; bid /= tiles_x;
    const int tyi = bid % tiles_y; bid /= tiles_y;
    const int n = bid;
    const int co0 = mb * K::BM;
    const int y0 = tyi * K::PH, x0 = txi * K::PW;
    constexpr int PAD = KS / 2;

    const float* __restrict__ xin = p.x + (int64_t)n * p.cin * p.h * p.w_;
    const float* __restrict__ sty = p.styles + (int64_t)n * p.cin;

    f32x16 acc[K::MTW][K::NTW];
#pragma unroll
    for (int i = 0; i < K::MTW; ++i)
#pragma unroll
        for (int j = 0; j < K::NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // B-operand base address of this lane inside the halo patch, per N tile.
    int boff[K::NTW];
#pragma unroll
    for (int j = 0; j < K::NTW; ++j) {
        const int py = (wn * K::NTW + j) * 2 + (l32 >> 4), px = l32 & 15;
        boff[j] = py * K::XW + px;
    }

    for (int ci0 = 0; ci0 < p.cin; ci0 += K::KC) {
        // ---- stage weights (x styles) : s_w[(tap*KC + cil)][co] ----
        for (int e = tid; e < K::BM * K::KK; e += 256) {
            const int co = e / K::KK, r = e - co * K::KK;
            const int cil = r / K::TAPS, tap = r - cil * K::TAPS;
            float v = 0.f;
            if (co0 + co < p.cout && ci0 + cil < p.cin)
                v = p.w[((int64_t)(co0 + co) * p.cin + ci0 + cil) * K::TAPS + tap] * sty[ci0 + cil];
            s_w[(tap * K::KC + cil) * K::WP + co] = v;
        }
        // ---- stage input halo patch ----
        for (int e = tid; e < K::KC * K::HP * K::HW; e += 256) {
            const int cil = e / (K::HP * K::HW), r = e - cil * (K::HP * K::HW);
            const int ry = r / K::HW, rx = r - ry * K::HW;
            const int yy = y0 - PAD + ry, xx = x0 - PAD + rx;
            float v = 0.f;
            if (ci0 + cil < p.cin && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w_)
                v = xin[((int64_t)(ci0 + cil) * p.h + yy) * p.w_ + xx];
            s_x[cil * K::XS + ry * K::XW + rx] = v;
        }
        __syncthreads();
        // ---- MFMA over the chunk: k-step = (tap, channel pair), lane half selects the channel ----
#pragma unroll
        for (int tap = 0; tap < K::TAPS; ++tap) {
            const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
            for (int cp = 0; cp < K::KC / 2; ++cp) {
                const int cil = cp * 2 + half;
                float a[K::MTW], b[K::NTW];
#pragma unroll
                for (int i = 0; i < K::MTW; ++i)
                    a[i] = s_w[(tap * K::KC + cil) * K::WP + (wm * K::MTW + i) * 32 + l32];
#pragma unroll
                for (int j = 0; j < K::NTW; ++j)
                    b[j] = s_x[cil * K::XS + boff[j] + ky * K::XW + kx];
#pragma unroll
                for (int i = 0; i < K::MTW; ++i)
#pragma unroll
                    for (int j = 0; j < K::NTW; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue ----
    float* __restrict__ yout = p.y + (int64_t)n * p.cout * p.h * p.w_;
#pragma unroll
    for (int j = 0; j < K::NTW; ++j) {
        const int py = (wn * K::NTW + j) * 2 + (l32 >> 4), px = l32 & 15;
        const int yy = y0 + py, xx = x0 + px;
        const bool pix_ok = yy < p.h && xx < p.w_;
        const float nz = (p.noise && pix_ok) ? p.noise[yy * p.w_ + xx] * p.noise_strength : 0.f;
#pragma unroll
        for (int i = 0; i < K::MTW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int co = co0 + (wm * K::MTW + i) * 32 + row;
                if (!pix_ok || co >= p.cout) continue;
                float v = acc[i][j][r];
                if (p.dcoefs) v *= p.dcoefs[(int64_t)n * p.cout + co];
                v += nz;
                if (p.bias) v += p.bias[co];
                if (p.act == 3) v = (v > 0.f) ? v : v * p.alpha;
                v *= p.gain;
                if (p.clamp >= 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
                yout[((int64_t)co * p.h + yy) * p.w_ + xx] = v;
            }
    }
}

template <int KS, int BIG>
static int launch_modconv(const ide3d_modconv_params& p, hipStream_t st) {
    using K = McCfg<KS, BIG>;
    const int tiles_x = cdiv(p.w_, K::PW), tiles_y = cdiv(p.h, K::PH), mblocks = cdiv(p.cout, K::BM);
    const int64_t nblocks = (int64_t)tiles_x * tiles_y * mblocks * p.n;
    if (nblocks > 0x7fffffff) { set_error("modconv2d: grid too large"); return IDE3D_EINVAL; }
    hipLaunchKernelGGL((modconv_kernel<KS, BIG>), dim3((unsigned)nblocks), dim3(256), 0, st, p, tiles_x, tiles_y, mblocks);
    IDE3D_CHECK_LAUNCH("modconv2d");
    return IDE3D_OK;
}

}  // namespace ide3d

extern "C" int ide3d_modconv2d(const ide3d_modconv_params* pp, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(pp != nullptr, "modconv2d: null params");
    const ide3d_modconv_params& p = *pp;
    IDE3D_CHECK_ARG(p.x && p.w && p.styles && p.y, "modconv2d: null tensor pointer");
    IDE3D_CHECK_ARG(p.n > 0 && p.cin > 0 && p.cout > 0 && p.h > 0 && p.w_ > 0, "modconv2d: bad shape");
    IDE3D_CHECK_ARG(p.k == 1 || p.k == 3, "modconv2d: kernel size must be 1 or 3 (got %d)", p.k);
    IDE3D_CHECK_ARG(p.act == 1 || p.act == 3, "modconv2d: act must be linear (1) or lrelu (3)");
    hipStream_t st = (hipStream_t)stream;
    const bool big = p.cout > 96;
    if (p.k == 3) return big ? launch_modconv<3, 1>(p, st) : launch_modconv<3, 0>(p, st);
    return big ? launch_modconv<1, 1>(p, st) : launch_modconv<1, 0>(p, st);
}
