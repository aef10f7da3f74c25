// modconv.hip — StyleGAN2 modulated convolutions as implicit GEMMs on the matrix cores with fused epilogues: an fp32-MFMA loop for
// every shape (described first) and a split-bf16 loop (bf16x6 / bf16x3 on v_mfma_f32_32x32x16_bf16, further down) that the big
// shared-weight 3x3 and transposed 3x3 layers take by default.
//
// Replaces the ATen convolutions behind `conv2d_gradfix.conv2d / conv_transpose2d`
// (torch_utils/ops/conv2d_gradfix.py:35,40) for the three shapes the generator uses, as called from
// `modulated_conv2d` (inversion/networks.py:55-130) via `conv2d_resample` (conv2d_resample.py:112-134):
//   mode 0  3x3, stride 1, pad 1 (correlation)          SynthesisLayer conv1 / up = 1
//   mode 1  1x1                                         ToRGBLayer / toSeg heads
//   mode 2  3x3 transposed, stride 2, pad 0 -> (2H+1)x(2W+1)   first half of an up-sampling SynthesisLayer
//           (the 4x4 FIR that follows is csrc/upfirdn2d.hip)
// with the un-fused modulation algebra of networks.py:99-114: the INPUT patch is scaled by the styles s[n,ci]
// while it is staged, the weights are shared by the whole batch, demodulation d[n,co] is applied to the
// accumulators, and (modes 0/1) noise + bias + leaky-ReLU + gain + clamp finish in registers
// (networks.py:457-512, :700-707).
//
// GEMM view: M = cout, N = batch x pixels, K = cin x taps on v_mfma_f32_32x32x2_f32 (exact fp32 products).
//   * 256-thread workgroup, BM x 128 output tile: BIG = 128 couts (waves 2 x 2, 4 accumulators each) or SMALL = 32
//     couts (waves 1 x 4).  The 128 "pixels" of a tile are TI images x PH x PW (1x8x16, 2x8x8 or 8x4x4) so that
//     low-resolution layers still fill the N dimension.
//   * weights are pre-packed once per call into [m-block][k-chunk][tap][ci][co] (ide3d_modconv_pack): a K chunk is
//     one contiguous slab copied with 16-byte loads and stored linearly in LDS (A operand, conflict-free);
//   * the input halo patch ((PH+2) x (PW+2) per channel, zero padded, x style) is staged once per chunk and read
//     once per tap with a shifted address (B operand): im2col never materialises;
//   * transposed convolution: the four output parity classes (oy%2, ox%2) are separate tile sets, each walking
//     only its own taps (4 / 2 / 2 / 1 of the 9), so no multiplications by inserted zeros are issued;
//   * double buffering: chunk c+1 is fetched (weights by LDS-DMA, input patch through registers) while chunk c feeds
//     the MFMAs, one barrier per chunk.  The MFMA operands are read from LDS by hand-issued `ds_read_b32` with
//     explicit `s_waitcnt lgkmcnt(n)` (tap t+1 in flight while tap t multiplies): the compiler then sees no LDS load
//     in the loop and no longer drains `vmcnt` — the weight DMA it cannot disambiguate — before the first operand
//     read, so the prefetch latency overlaps the MFMAs (+3 % frames/s); phase cycles: scripts/modconv_trace.py;
//   * split-K (low-resolution 512-channel layers have too few tiles to fill 256 CUs): partial sums go to a
//     workspace, `modconv_epilogue_kernel` reduces them and applies the epilogue — deterministic, no atomics.
// Bound: fp32 MFMA (157.3 TFLOP/s peak); LDS and L2 traffic stay below 10 B/clk/CU.  The split-bf16 loop is bound by the bf16 MFMA
// (2.5 PFLOP/s / 6 products) and, below 16 x 16-pixel tiles, by its weight stream.
#include "common.h"
#include "knobs.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace ide3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef IDE3D_MC_TRACE
// Developer aid (make EXTRA=-DIDE3D_MC_TRACE): cycles wave 0 of one block spends in each phase of its chunk loop,
// accumulated in registers and written once after the loop; read back with ide3d_debug_mc().  Not part of the ABI.
__device__ unsigned long long g_mc_dbg[256];
#define IDE3D_MC_TS(k) { const unsigned long long now_ = __builtin_readcyclecounter(); mc_acc[k] += now_ - mc_last; mc_last = now_; }
#else
#define IDE3D_MC_TS(k)
#endif

enum { MODE_CONV3 = 0, MODE_CONV1 = 1, MODE_TCONV3 = 2, MODE_TCONV3A = 3, MODE_CONV3S2 = 4 };

template <int MODE> struct ModeCfg;
template <> struct ModeCfg<MODE_CONV3>  { static constexpr int KC = 4, WTAPS = 9, MAXT = 9; };
template <> struct ModeCfg<MODE_CONV1>  { static constexpr int KC = 16, WTAPS = 1, MAXT = 1; };
// 3x3 stride-2 convolution without padding (the conv after the low-pass filter of a down-sampling Conv2dLayer,
// conv2d_resample.py:100-103): output (h - 3) / 2 + 1; a P x Q output tile reads a (2P + 1) x (2Q + 1) input patch.
template <> struct ModeCfg<MODE_CONV3S2> { static constexpr int KC = 4, WTAPS = 9, MAXT = 9; };
template <> struct ModeCfg<MODE_TCONV3> { static constexpr int KC = 4, WTAPS = 9, MAXT = 4; };
// all-class transposed conv: one block computes the four output parity classes of its grid tile from ONE staged input
// patch and all 9 taps (each tap feeds exactly one class), i.e. 4x the MFMA work per staging step / barrier of mode 2.
template <> struct ModeCfg<MODE_TCONV3A> { static constexpr int KC = 4, WTAPS = 9, MAXT = 9; };

template <int MODE, int BIG, int TI, int PH, int PW, int NWV = 4, int KCO = 0>      // KCO: input channels per K chunk (0 = the mode's default)
struct McCfg {
    static constexpr int NT = 64 * NWV;                     // threads per workgroup (4 or 8 waves)
    static constexpr int BN = TI * PH * PW;
    static_assert(BN == 64 || BN == 128 || BN == 256, "pixel tile must hold 64, 128 or 256 pixels");
    static constexpr int NCLS = (MODE == MODE_TCONV3A) ? 4 : 1;   // accumulator sets (output parity classes)
    using MC = ModeCfg<MODE>;
    static constexpr int KC = KCO ? KCO : MC::KC, WTAPS = MC::WTAPS, MAXT = MC::MAXT;
    // BIG: 1 -> BM 128 (waves 2 x 2, two M tiles each); 2 -> BM 64 (waves 2 x 2, one M tile each); 0 -> BM 32 (waves 1 x 4)
    static constexpr int WM = BIG ? 2 : 1, WN = NWV / WM;
    static constexpr int MTW = (BIG == 1) ? 2 : 1;          // 32-row M tiles per wave
    static constexpr int NTW = (BN / 32) / WN;              // 32-pixel N tiles per wave
    static constexpr int BM = WM * MTW * 32;
    static constexpr int STRIDE = (MODE == MODE_CONV3S2) ? 2 : 1;
    static constexpr int HALO = (MODE == MODE_CONV1 || MODE == MODE_CONV3S2) ? 0 : 1;
    static constexpr int HP = (MODE == MODE_CONV3S2) ? 2 * PH + 1 : PH + 2 * HALO;   // (halo) patch
    static constexpr int HW = (MODE == MODE_CONV3S2) ? 2 * PW + 1 : PW + 2 * HALO;
    // LDS row pitch: with 16-pixel rows a pitch = 16 (mod 32) puts the two pixel rows of an MFMA N tile on disjoint banks
    static constexpr int XW = (MODE == MODE_CONV1) ? PW : (MODE == MODE_CONV3S2) ? ((HW + 3) & ~3) : ((PW == 16) ? 48 : HW + 2);
    static constexpr int XS = HP * XW;                      // per-channel pitch
    static constexpr int XI = KC * XS;                      // per-image pitch
    static constexpr int LDS_W = MAXT * KC * BM;            // floats, one buffer
    static constexpr int LDS_X = TI * XI;
    static constexpr int NW4 = (LDS_W / 4 + NT - 1) / NT;   // float4 weight loads per thread per chunk
    // flat 1x1 tiles (one row of PW pixels, no halo) stage the patch in 16-byte pieces
    static constexpr int VW = (MODE == MODE_CONV1 && PH == 1 && PW % 4 == 0) ? 4 : 1;
    static constexpr int NXE = (TI * KC * HP * HW / VW + NT - 1) / NT;   // input elements (VW floats each) per thread per chunk
};

template <int N>
__device__ __forceinline__ int sel(const int (&a)[N], int t) {
    int v = a[0];
#pragma unroll
    for (int i = 1; i < N; ++i) v = (t == i) ? a[i] : v;
    return v;
}

__host__ __device__ inline int mc_bm(int cout) { return cout > 96 ? 128 : (cout > 32 ? 64 : 32); }
__host__ __device__ inline int mc_kc(int k) { return k == 3 ? 4 : 16; }

// ------------------------------------------------------------------------------------------------
// weight packing: w [cout, cin, k, k] -> [mb][cc][tap][cil][co]  (zero padded to full blocks)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
modconv_pack_kernel(const float* __restrict__ w, int64_t w_batch_stride, int nbatch, int cout, int cin, int taps, int bm, int kc,
                    int mblocks, int cchunks, float* __restrict__ out) {
    const int64_t total = (int64_t)nbatch * mblocks * cchunks * taps * kc * bm;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int64_t r = i;
        const int co_l = (int)(r % bm); r /= bm;
        const int cil = (int)(r % kc); r /= kc;
        const int tap = (int)(r % taps); r /= taps;
        const int cc = (int)(r % cchunks); r /= cchunks;
        const int mb = (int)(r % mblocks); r /= mblocks;
        const int nb = (int)r;
        const int co = mb * bm + co_l, ci = cc * kc + cil;
        out[i] = (co < cout && ci < cin) ? w[nb * w_batch_stride + ((int64_t)co * cin + ci) * taps + tap] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// main kernel
// ------------------------------------------------------------------------------------------------
struct ConvGeom {
    int tiles_x[4], tiles_y[4];     // per parity class (class 0 only for modes 0 / 1)
    int tile_base[5];               // prefix sum of tiles per class (per image group)
    int img_groups;                 // ceil(n / TI)
    int mblocks, cchunks, split_k, chunks_per_split;
    int oh, ow;                     // output size
    int debug;                      // experiments only: 1 = no staging after the first chunk, 2 = staging but no MFMA
};

// ---- hand-scheduled LDS operand reads ----------------------------------------------------------------------------
// The MFMA operands are read from LDS with `ds_read_b32` issued from inline assembly and waited for with explicit
// `s_waitcnt lgkmcnt(n)`: (1) the reads of tap t+1 are in flight while the MFMAs of tap t issue (hipcc otherwise places
// every read right before its use and waits for it); (2) the compiler sees no LDS load in the loop, so it does not
// drain `vmcnt` — the LDS-DMA weight copy of the NEXT chunk, which may alias in its view — before the first operand
// read of THIS chunk: the global latency of a chunk's prefetch now overlaps the MFMAs instead of preceding them.
template <int OFF_BYTES>
__device__ __forceinline__ float lds_read_async(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF_BYTES));
    return v;
}
template <int PENDING>
__device__ __forceinline__ void lds_wait(float& first) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(first) : "n"(PENDING)); }
__device__ __forceinline__ void lds_pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(size_t)p; }

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// patch offset (floats) and accumulator set of tap T
template <int MODE, int XW>
__host__ __device__ constexpr int tap_patch_offset(int t) {
    return (MODE == MODE_CONV1) ? 0
         : (MODE == MODE_TCONV3A) ? ((t / 3 == 2) ? 0 : 1) * XW + ((t % 3 == 2) ? 0 : 1)
         : (t / 3) * XW + (t % 3);
}
template <int MODE>
__host__ __device__ constexpr int tap_class(int t) { return (MODE == MODE_TCONV3A) ? ((t / 3) & 1) * 2 + ((t % 3) & 1) : 0; }

// ---- epilogue (shared by the fp32-MFMA and the split-bf16 main loops) ------------------------------------------------
// acc[class][m tile][n tile] in the 32x32 MFMA C layout; `lp` = the pixel (0..31, in tile order) that this lane's MFMA column
// holds inside each 32-pixel N tile (the lane's own index for the fp32 path; a permutation for the split-bf16 path).
// `s_w`: LDS scratch of SCRATCH floats that the main loop has finished with (the caller's last barrier covers it).
// `row_unscale` / `x_unscale`: the f16x3 loop computes with weights scaled per output row and the patch scaled per image (exact powers
// of two); both are undone here together with the demodulation (also for split-K partials, which are stored unscaled).
// `p.y_amax` (optional): max |finished value| per image, for the consumer's f16x3 scale (wave reduction + one atomic per wave and tile).
// running maximum of |v|: ONE v_max_f32 with the |.| source modifier per value (these epilogues are issue-bound).  A NaN does not raise
// it (v_max returns the other operand); an inf does: the consumer then sees a non-finite bound and computes that image without range
// scaling (f16_scale) — an image that holds an inf is broken in every arithmetic.
__device__ __forceinline__ void amax_acc(float& m, float v) { m = fmaxf(m, fabsf(v)); }
__device__ __forceinline__ void amax_commit(float* y_amax, int n, float v, bool) {       // several images per tile (tiny layers): per lane
    if (v > 0.f) amax_raise(y_amax, n, v);
}

// ROWPAR >= 0 (all-class transposed form): acc holds the two column classes of output rows 2 gy + ROWPAR only
template <int MODE, int TI, int PH, int PW, int NWV, int NCLS, int MTW, int NTW, int BM, int SCRATCH, int ROWPAR = -1>
__device__ __forceinline__ void modconv_finish(const ide3d_modconv_params& p, float* __restrict__ partial, const ConvGeom& g,
                                               f32x16 (&acc)[NCLS][MTW][NTW], float* s_w, int mb, int n0, int y0, int x0, int split, int cls,
                                               int wm, int wn, int lp, const float* __restrict__ row_unscale = nullptr, float x_unscale = 1.f,
                                               int team_tid = -1, float* amax_scratch = nullptr) {
    // team_tid >= 0: this tile belongs to one of two 64 * NWV-thread teams of the workgroup (modconv_split_kernel<..., TEAMS = 2>): thread
    // index inside the team; s_w is the team's own LDS, amax_scratch the workgroup's (both teams work on the same image)
    constexpr int NT = 64 * NWV;
    const int tid = team_tid >= 0 ? team_tid : (int)threadIdx.x, lane = tid & 63, wid = tid >> 6, half = lane >> 5;
    const int cpy = cls >> 1, cpx = cls & 1;
    const bool raw = (g.split_k > 1);
    // Demodulation coefficients and biases of this block's BM output channels go to LDS first (the K loop has finished with
    // s_w): a lane's 16 x MTW accumulator rows are 16 x MTW different channels, and one global load per row in front of each
    // store exposes its latency that many times (measured: the epilogue took as long as 7 - 60 K chunks).
    float* const s_dm = s_w;                       // [TI][BM]
    float* const s_bi = s_w + TI * BM;          // [BM]
    for (int e = tid; e < TI * BM; e += NT) {
        const int rl = e % BM, co = mb * BM + rl, n = min(n0 + e / BM, p.n - 1);
        float d = (!raw && p.dcoefs && co < p.cout) ? p.dcoefs[(int64_t)n * p.cout + co] : 1.f;
        if (row_unscale) d *= row_unscale[co] * x_unscale;        // exact: powers of two
        s_dm[e] = d;
        if (e < BM) s_bi[e] = (!raw && p.bias && co < p.cout) ? p.bias[co] : 0.f;
    }
    __syncthreads();
    float* const dst = raw ? partial + (int64_t)split * p.n * p.cout * g.oh * g.ow : p.y;
    // row pitch of the destination: the finished output may have padded rows (p.y_pitch: the (2h + 1)-wide transposed-convolution result
    // with 16-byte aligned rows for the FIR that reads it next); split-K partials are dense
    const int64_t pitch = (!raw && p.y_pitch > 0) ? p.y_pitch : g.ow;
    // Branch-free finish: absent terms are identities (d = 1, b = 0 in LDS; slope 1 = linear; clamp +inf; split-K partials
    // take all of them), so the 128 values of a lane do not cost four uniform branches each.
    const float e_alpha = (!raw && p.act == 3) ? p.alpha : 1.f, e_gain = raw ? 1.f : p.gain;
    const float e_clamp = (!raw && p.clamp >= 0.f) ? p.clamp : __builtin_inff();
    const float e_nstr = (!raw && p.noise) ? p.noise_strength : 0.f;
    auto finish = [&](float v, float d, float nz, float bb) {
        v *= d; v += nz; v += bb;
        v = (v > 0.f) ? v : v * e_alpha;
        v *= e_gain;
        return fminf(fmaxf(v, -e_clamp), e_clamp);
    };
    const bool want_amax = !raw && p.y_amax != nullptr;
    float amax_tile = 0.f;                                   // TI == 1: one image per tile, reduced once at the end
    // Vector epilogue: a lane holds ONE pixel of 16 channels per accumulator, i.e. 4-byte stores and 16 different
    // demodulation / bias values.  RR channel rows of raw accumulators go through a wave-private LDS tile ([row][pixel], the
    // two x-parity classes of the all-class transposed convolution interleaved); a lane then takes 4 consecutive output
    // pixels of ONE channel, finishes them (one d / b pair, one 16-byte noise load) and stores 16 bytes.
    constexpr int QX = (MODE == MODE_TCONV3A) ? 2 : 1, QY = NCLS / QX;
    static_assert(ROWPAR < 0 || (MODE == MODE_TCONV3A && NCLS == 2), "one row parity = two column classes");
    constexpr int TW = 32 * QX, TP = TW + 8;                     // staged row: floats, pitch (rows r and r + 4 on disjoint banks)
    constexpr int AVAIL = SCRATCH - (TI + 1) * BM;
    constexpr int RR = (MODE == MODE_TCONV3 || PW % 4 != 0) ? 0 : (AVAIL >= NWV * 32 * TP) ? 32 : (AVAIL >= NWV * 16 * TP) ? 16 : (AVAIL >= NWV * 8 * TP) ? 8 : 0;
    if constexpr (RR > 0) {
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
        float* const T = s_w + (TI + 1) * BM + wid * (RR * TP);
#pragma unroll
        for (int qy = 0; qy < QY; ++qy)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int pbase = (wn * NTW + j) * 32;
            // what a lane stores depends on (row, pixel group); the pixel group c4 = lane % (TW / 4) is the same in every round
            const int c4 = lane % (TW / 4), row0 = lane / (TW / 4);
            const int pix = pbase + (c4 * 4) / QX, ti = pix / (PH * PW), rem = pix % (PH * PW);
            const int n = n0 + ti, gy = y0 + rem / PW, gx = x0 + rem % PW;
            const int oy = (MODE == MODE_TCONV3A) ? 2 * gy + (ROWPAR >= 0 ? ROWPAR : qy) : gy, ox = (MODE == MODE_TCONV3A) ? 2 * gx : gx;
            const bool px_ok = n < p.n && oy < g.oh && ox < g.ow, full = ox + 3 < g.ow;
            const int64_t pofs = (int64_t)oy * pitch + ox, nofs = (int64_t)oy * g.ow + ox;
            f32x4u nz = {0.f, 0.f, 0.f, 0.f};
            if (e_nstr != 0.f && px_ok) {
                if (full) nz = *reinterpret_cast<const f32x4u*>(p.noise + nofs) * e_nstr;
                else for (int e = 0; e < 4; ++e) if (ox + e < g.ow) nz[e] = p.noise[nofs + e] * e_nstr;
            }
            float* const o_px = dst + (int64_t)n * p.cout * ((int64_t)g.oh * pitch) + pofs;
            float amax_j = 0.f;
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int g0 = 0; g0 < 4; g0 += RR / 8) {
#pragma unroll
                for (int qx = 0; qx < QX; ++qx)
#pragma unroll
                    for (int gg = 0; gg < RR / 8; ++gg)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            T[(gg * 8 + rr + 4 * half) * TP + lp * QX + qx] = acc[qy * QX + qx][i][j][(g0 + gg) * 4 + rr];
#pragma unroll
                for (int k = 0; k < RR * (TW / 4) / 64; ++k) {
                    const int rowl = row0 + k * (256 / TW);
                    f32x4u v4 = *reinterpret_cast<const f32x4u*>(T + rowl * TP + c4 * 4);
                    const int rl = (wm * MTW + i) * 32 + g0 * 8 + rowl, co = mb * BM + rl;
                    if (!px_ok || co >= p.cout) continue;
                    const float d = s_dm[ti * BM + rl], bb = s_bi[rl];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = finish(v4[e], d, nz[e], bb);
                    float* o = o_px + (int64_t)co * ((int64_t)g.oh * pitch);
                    if (full) {
                        *reinterpret_cast<f32x4u*>(o) = v4;
                        if (want_amax) { amax_acc(amax_j, v4[0]); amax_acc(amax_j, v4[1]); amax_acc(amax_j, v4[2]); amax_acc(amax_j, v4[3]); }
                    } else for (int e = 0; e < 4; ++e) if (ox + e < g.ow) { o[e] = v4[e]; amax_acc(amax_j, v4[e]); }
                }
            }
            if (want_amax) { if (TI == 1) amax_tile = fmaxf(amax_tile, amax_j); else if (px_ok) amax_commit(p.y_amax, n, amax_j, false); }
        }
    } else {
#pragma unroll
    for (int q = 0; q < NCLS; ++q)
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int pix = (wn * NTW + j) * 32 + lp;
        const int ti = pix / (PH * PW), rem = pix % (PH * PW);
        const int n = n0 + ti;
        const int gy = y0 + rem / PW, gx = x0 + rem % PW;              // (class-)grid coordinates
        const int oy = (MODE == MODE_TCONV3) ? 2 * gy + cpy : gy, ox = (MODE == MODE_TCONV3) ? 2 * gx + cpx : gx;
        const bool ok = n < p.n && oy < g.oh && ox < g.ow;
        if (!ok) continue;
        const float nz = (e_nstr != 0.f) ? p.noise[oy * g.ow + ox] * e_nstr : 0.f;
        float amax_j = 0.f;
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (wm * MTW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;      // row inside the block
                const int co = mb * BM + rl;
                if (co >= p.cout) continue;
                const float v = finish(acc[q][i][j][r], s_dm[ti * BM + rl], nz, s_bi[rl]);
                *(dst + (((int64_t)n * p.cout + co) * g.oh + oy) * pitch + ox) = v;
                amax_acc(amax_j, v);
            }
        if (want_amax) { if (TI == 1) amax_tile = fmaxf(amax_tile, amax_j); else amax_commit(p.y_amax, n, amax_j, false); }
    }
    }
    if (TI == 1 && p.y_amax != nullptr && !raw && n0 < p.n)        // (uniform over the workgroup) one read / atomic per tile
        amax_raise_block(p.y_amax, n0, amax_tile, amax_scratch ? amax_scratch : s_w);          // its first barrier: every thread has finished reading s_dm / s_bi / T
}

// One output tile: `tl` = index inside its tile set (row-major, `tiles_x` per row), `cls` = output parity class
// (MODE_TCONV3 only).  s_w / s_x: two buffers of K::LDS_W / K::LDS_X floats.
template <int MODE, int BIG, int TI, int PH, int PW, int NWV, int KCO>
__device__ __forceinline__ void modconv_tile(const ide3d_modconv_params& p, const float* __restrict__ wp, float* __restrict__ partial,
                                             const ConvGeom& g, float* s_w, float* s_x, int mb, int tl, int grp, int split, int cls,
                                             int tiles_x) {
    using K = McCfg<MODE, BIG, TI, PH, PW, NWV, KCO>;
#ifdef IDE3D_MC_TRACE
    const unsigned long long mc_t0 = __builtin_readcyclecounter();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / K::WN, wn = wid % K::WN;
    const int half = lane >> 5, l32 = lane & 31;
    const int txi = tl % tiles_x, tyi = tl / tiles_x;
    const int y0 = tyi * PH, x0 = txi * PW;                       // tile origin in (class-)grid coordinates
    const int n0 = grp * TI;
    const int cpy = cls >> 1, cpx = cls & 1;                      // output parity of this class (tconv)

    // ---- tap table of this block: weight index and patch offsets (dy, dx) ----
    int ntaps, t_widx[K::MAXT], t_off[K::MAXT];
    if (MODE == MODE_CONV3 || MODE == MODE_CONV3S2) {
        ntaps = 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) { t_widx[t] = t; t_off[t] = (t / 3) * K::XW + (t % 3); }
    } else if (MODE == MODE_CONV1) {
        ntaps = 1; t_widx[0] = 0; t_off[0] = 0;
    } else if (MODE == MODE_TCONV3A) {
        ntaps = 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t % 3;
            t_widx[t] = t; t_off[t] = ((ky == 2) ? 0 : 1) * K::XW + ((kx == 2) ? 0 : 1);
        }
    } else {
        // out[2i+ky] += x[i] w[ky]:  even output rows take ky = 0 (i = c) and ky = 2 (i = c-1); odd rows ky = 1 (i = c).
        // patch row of input i for grid row c (patch origin = y0 - 1): (c - y0) + 1 + (i - c).
        const int nky = cpy ? 1 : 2, nkx = cpx ? 1 : 2;
        ntaps = nky * nkx;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int a = t / nkx, b = t % nkx;
            const int ky = cpy ? 1 : (a == 0 ? 0 : 2), kx = cpx ? 1 : (b == 0 ? 0 : 2);
            const int dy = (ky == 2) ? 0 : 1, dx = (kx == 2) ? 0 : 1;
            t_widx[t] = ky * 3 + kx; t_off[t] = dy * K::XW + dx;
        }
    }

    // ---- staging plan (fixed across chunks) ----
    const float* __restrict__ wsrc = wp + (int64_t)mb * g.cchunks * (K::WTAPS * K::KC * K::BM)
                                     + (p.w_batch_stride ? (int64_t)n0 * g.mblocks * g.cchunks * (K::WTAPS * K::KC * K::BM) : 0);
    int x_src[K::NXE], x_dst[K::NXE], x_img[K::NXE];
#pragma unroll
    for (int i = 0; i < K::NXE; ++i) {
        const int e = tid + i * K::NT;
        x_src[i] = -1; x_dst[i] = -1; x_img[i] = 0;
        if (e < TI * K::KC * K::HP * K::HW / K::VW) {
            int r = e;
            const int rx = (r % (K::HW / K::VW)) * K::VW; r /= (K::HW / K::VW);
            const int ry = r % K::HP; r /= K::HP;
            const int cil = r % K::KC; r /= K::KC;
            const int ti = r;
            const int yy = y0 * K::STRIDE - K::HALO + ry, xx = x0 * K::STRIDE - K::HALO + rx;
            x_dst[i] = ti * K::XI + cil * K::XS + ry * K::XW + rx;
            x_img[i] = ti * K::KC + cil;                               // (image, channel-in-chunk)
            if (n0 + ti < p.n && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w_)
                x_src[i] = ((n0 + ti) * p.cin + cil) * (p.h * p.w_) + yy * p.w_ + xx;   // + ci0 * h * w per chunk
        }
    }
    const int hw = p.h * p.w_;

    f32x16 acc[K::NCLS][K::MTW][K::NTW];
#pragma unroll
    for (int q = 0; q < K::NCLS; ++q)
#pragma unroll
        for (int i = 0; i < K::MTW; ++i)
#pragma unroll
            for (int j = 0; j < K::NTW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;

    // B-operand base address of this lane per N tile: pixel j -> (ti, py, px)
    int boff[K::NTW];
#pragma unroll
    for (int j = 0; j < K::NTW; ++j) {
        const int pix = (wn * K::NTW + j) * 32 + l32;
        const int ti = pix / (PH * PW), rem = pix % (PH * PW);
        boff[j] = ti * K::XI + (rem / PW) * K::STRIDE * K::XW + (rem % PW) * K::STRIDE + half * K::XS;
    }
    const int aoff = half * K::BM + wm * K::MTW * 32 + l32;

    const int c_begin = split * g.chunks_per_split;
    const int c_end = min(c_begin + g.chunks_per_split, g.cchunks);

    // Staging is split into an issue-only half (unconditional loads from clamped addresses: no exec-mask
    // branches, no vmcnt(0) between loads) and a commit half that runs after the MFMAs of the current chunk.
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    using XV = std::conditional_t<K::VW == 4, f32x4v, float>;
    XV xreg[K::NXE];
    float sreg[K::NXE];
    const int img_last = p.n - 1;
    constexpr int PIECE = (K::KC * K::BM < 256) ? K::KC * K::BM : 256;   // floats per LDS-DMA piece (<= 64 lanes x 16 B)
    constexpr int PPT = K::KC * K::BM / PIECE;                            // pieces per tap slab
    const int w_pieces = ntaps * PPT;
    // Weights: global -> LDS by DMA (global_load_lds_dwordx4): the packed slab is already the LDS image, so there is no
    // VGPR round trip and no ds_write; each wave moves every 4th piece.  Input patch: issue-only loads into
    // registers (clamped addresses, no branches); scaling by the styles and zero fill happen in commit().
#ifdef IDE3D_MC_TRACE
    unsigned long long mc_acc[6] = {0, 0, 0, 0, 0, 0}, mc_last = 0;
#endif
    auto fetch = [&](int c, int buf) {
        const float* ws = wsrc + (int64_t)c * (K::WTAPS * K::KC * K::BM);
        for (int i = wid; i < w_pieces; i += NWV) {
            const int t = i / PPT, r = (i - t * PPT) * PIECE;
            const float* src = ws + ((MODE == MODE_TCONV3) ? sel(t_widx, t) : t) * (K::KC * K::BM) + r + lane * 4;
            if (lane * 4 < PIECE)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(s_w + buf * K::LDS_W + t * (K::KC * K::BM) + r), 16, 0, 0);
        }
        IDE3D_MC_TS(5)
        const int ci0 = c * K::KC;
#pragma unroll
        for (int i = 0; i < K::NXE; ++i) {
            const int cil = x_img[i] % K::KC;
            const int ci = min(ci0 + cil, p.cin - 1);
            const int64_t src = (x_src[i] >= 0) ? (int64_t)x_src[i] + (int64_t)(ci - cil) * hw : 0;
            xreg[i] = *reinterpret_cast<const XV*>(p.x + src);
            sreg[i] = p.styles ? p.styles[min(n0 + x_img[i] / K::KC, img_last) * p.cin + ci] : 1.0f;
        }
    };
    auto commit = [&](int buf, int c) {
        const int ci0 = c * K::KC;
#pragma unroll
        for (int i = 0; i < K::NXE; ++i) {
            const int cil = x_img[i] % K::KC;
            const bool live = x_src[i] >= 0 && ci0 + cil < p.cin;
            XV v = xreg[i] * sreg[i];
            if (!live) __builtin_memset(&v, 0, sizeof(v));
            if (x_dst[i] >= 0) *reinterpret_cast<XV*>(s_x + buf * K::LDS_X + x_dst[i]) = v;
        }
    };

    if (c_begin < c_end) {
        fetch(c_begin, 0);
        commit(0, c_begin);
    }
    __syncthreads();
#ifdef IDE3D_MC_TRACE
    mc_last = __builtin_readcyclecounter();
    const unsigned long long mc_prologue = mc_last - mc_t0;
#endif
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        IDE3D_MC_TS(0)
        if (c + 1 < c_end && g.debug != 1) fetch(c + 1, buf ^ 1);
        IDE3D_MC_TS(1)
        const float* sw = s_w + buf * K::LDS_W;
        const float* sx = s_x + buf * K::LDS_X;
        auto tap_body = [&](int t, int off, auto cls_tag) {
            constexpr int q = decltype(cls_tag)::value;
#pragma unroll
            for (int cp = 0; cp < K::KC / 2; ++cp) {
                float a[K::MTW], b[K::NTW];
#pragma unroll
                for (int i = 0; i < K::MTW; ++i) a[i] = sw[(t * K::KC + cp * 2) * K::BM + aoff + i * 32];
#pragma unroll
                for (int j = 0; j < K::NTW; ++j) b[j] = sx[boff[j] + cp * 2 * K::XS + off];
#pragma unroll
                for (int i = 0; i < K::MTW; ++i)
#pragma unroll
                    for (int j = 0; j < K::NTW; ++j)
                        acc[q][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[q][i][j], 0, 0, 0);
            }
        };
        using C0 = std::integral_constant<int, 0>;
        if constexpr (MODE == MODE_TCONV3) {
            for (int t = 0; t < ntaps; ++t) tap_body(t, sel(t_off, t), C0{});
        } else {
            // software pipeline over steps = (tap, pair of k-steps): operand set (s & 1) feeds the MFMAs of step s while set
            // ((s + 1) & 1) is loading
            constexpr int KP = K::KC / 2, G = 2, SPT = KP / G, NSTEP = K::MAXT * SPT, NRD = G * (K::MTW + K::NTW);
            static_assert(KP % G == 0 && NRD <= 15, "lgkmcnt is a 4-bit counter");
            const unsigned a_base = lds_addr(sw + aoff);
            unsigned b_base[K::NTW];
#pragma unroll
            for (int j = 0; j < K::NTW; ++j) b_base[j] = lds_addr(sx + boff[j]);
            float av[2][G][K::MTW], bv[2][G][K::NTW];
            auto issue = [&](auto ss) {
                constexpr int S = decltype(ss)::value, T = S / SPT, CP0 = (S % SPT) * G, B = S & 1;
                static_for<G>([&](auto cc) {
                    constexpr int CP = CP0 + decltype(cc)::value, CI = decltype(cc)::value;
                    static_for<K::MTW>([&](auto ii) {
                        constexpr int I = decltype(ii)::value;
                        av[B][CI][I] = lds_read_async<((T * K::KC + CP * 2) * K::BM + I * 32) * 4>(a_base);
                    });
                    static_for<K::NTW>([&](auto jj) {
                        constexpr int J = decltype(jj)::value;
                        bv[B][CI][J] = lds_read_async<(CP * 2 * K::XS + tap_patch_offset<MODE, K::XW>(T)) * 4>(b_base[J]);
                    });
                });
            };
            issue(C0{});
            static_for<NSTEP>([&](auto ss) {
                constexpr int S = decltype(ss)::value, B = S & 1, Q = tap_class<MODE>(S / SPT);
                if constexpr (S + 1 < NSTEP) issue(std::integral_constant<int, S + 1>{});
                lds_wait<(S + 1 < NSTEP) ? NRD : 0>(av[B][0][0]);
#pragma unroll
                for (int cp = 0; cp < G; ++cp) {
#pragma unroll
                    for (int i = 0; i < K::MTW; ++i) lds_pin(av[B][cp][i]);
#pragma unroll
                    for (int j = 0; j < K::NTW; ++j) lds_pin(bv[B][cp][j]);
                }
#pragma unroll
                for (int cp = 0; cp < G; ++cp)
#pragma unroll
                    for (int i = 0; i < K::MTW; ++i)
#pragma unroll
                        for (int j = 0; j < K::NTW; ++j)
                            acc[Q][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[B][cp][i], bv[B][cp][j], acc[Q][i][j], 0, 0, 0);
            });
        }
        IDE3D_MC_TS(2)
        if (c + 1 < c_end && g.debug != 1) commit(buf ^ 1, c + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the LDS-DMA weight slab of chunk c + 1 has landed
        IDE3D_MC_TS(3)
        if (g.debug != 1) __syncthreads();
        IDE3D_MC_TS(4)
    }

#ifdef IDE3D_MC_TRACE
    const unsigned long long mc_t1 = __builtin_readcyclecounter();
#endif
    modconv_finish<MODE, TI, PH, PW, NWV, K::NCLS, K::MTW, K::NTW, K::BM, 2 * K::LDS_W>(p, partial, g, acc, s_w, mb, n0, y0, x0, split, cls, wm, wn, l32);
#ifdef IDE3D_MC_TRACE
    if (blockIdx.x == 100 && threadIdx.x == 0) {
        for (int k = 0; k < 6; ++k) g_mc_dbg[k] = mc_acc[k];
        g_mc_dbg[7] = (unsigned long long)(c_end - c_begin);
        g_mc_dbg[8] = mc_prologue;
        g_mc_dbg[9] = __builtin_readcyclecounter() - mc_t1;      // epilogue: issue of the stores (not their completion)
    }
#endif
}


// ------------------------------------------------------------------------------------------------
// split-bf16 main loop (3x3 and all-class transposed 3x3, 8x16 / 16x16 / 4x16 pixel tiles)
// ------------------------------------------------------------------------------------------------
// fp32 has no fast matrix path on gfx950 (v_mfma_f32_32x32x2_f32 runs at the vector rate, 1/16 of the bf16 rate), so the
// big 3x3 layers can instead split every fp32 operand into PARTS bf16 pieces (a = a0 + a1 [+ a2], round-to-nearest at each
// step, residuals exact in fp32) and sum the products a_i * b_j with i + j < PARTS on v_mfma_f32_32x32x16_bf16 (bf16 products
// are exact in fp32; accumulation is fp32 like the fp32 MFMA's):
//   PARTS = 3: 6 products, dropped terms <= 2^-24 relative (a1 b2 + a2 b1 + ...): fp32-grade products at 6 / 16 of the fp32 MFMA time;
//   PARTS = 2: 3 products, dropped terms ~ 2^-17 relative: 3 / 16 of the time.
// Layout differences from the fp32 loop:
//   * K chunk = 16 input channels = one MFMA K; a lane holds 8 consecutive channels of one pixel / output channel (16 bytes,
//     `ds_read_b128`).  LDS images in 16-byte units: weights [tap][part][k half][co], patch [part][k half][y][x];
//   * weights are staged per (chunk, kernel row) - 3 taps, 96 x PARTS x BM bytes - by LDS-DMA from the pre-split packed copy;
//     the patch is staged once per chunk: wave q loads channel quad q of every patch pixel (its 4 styles are wave-uniform),
//     scales, splits and writes 8 bytes per part;
//   * an MFMA N tile is 2 pixel rows x 16: lane l takes x = l % 16 and row = bit2 ^ bit3 ^ bit4 of l, so that each of the four
//     16-lane groups a `ds_read_b128` is serviced in ({0-3,12-15,20-27}, ...) reads one contiguous 256-byte row segment:
//     conflict-free for every tap shift and any row pitch.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// WBUF: weight stage buffers (2 = the next stage streams in behind the MFMAs; 1 = two workgroups per CU hide each other's staging).
// NWV: 4 waves (2 x 2), or 8 waves (2 x 4, two per SIMD) with time-shifted roles: waves 0-3 stage the patch and start multiplying at
// once, waves 4-7 first issue the weight DMA of the next stage (the issuing wave is stuck for most of the transfer, ~19 B/clk/CU)
// and multiply afterwards: the matrix pipe of every SIMD always has one of the two to take instructions from.
// PAIR (all-class transposed form only): the workgroup computes the two column classes of ONE output-row parity (see modconv_split_pair_kernel)
template <int MODE, int BIG, int PH, int PARTS, int WBUF = 2, int NWV_ = 4, int PAIR = 0>
struct SpCfg {
    static_assert(MODE == MODE_CONV3 || MODE == MODE_TCONV3A, "split-bf16 loop: 3x3 and all-class transposed 3x3 only");
    static_assert(BIG == 1 || BIG == 2, "64- or 128-row blocks");
    static_assert(NWV_ == 4 || NWV_ == 8, "4 or 8 waves");
    static constexpr int PW = 16, NWV = NWV_, NT = 64 * NWV;
    static constexpr int BN = PH * PW;
    static_assert(!PAIR || MODE == MODE_TCONV3A, "row-parity pairs exist in the transposed form only");
    static constexpr int NCLS = (MODE == MODE_TCONV3A) ? (PAIR ? 2 : 4) : 1;
    // wave grid WM x WN over (rows, pixels).  64-row blocks on 32 x 16 pixels (round 4): 1 x 8, i.e. 64 x 64 outputs per wave like the
    // 128-row forms — the 2 x 4 grid of the 16 x 16-pixel form gives 32 x 64 per wave: 3 operand reads per 2 MFMAs instead of 4 per 4
    static constexpr int BM = (BIG == 1) ? 128 : 64;
    static constexpr int WM = (BIG == 2 && PH == 32) ? 1 : 2, WN = NWV / WM;
    static constexpr int MTW = BM / 32 / WM, NTW = BN / 32 / WN;
    static_assert(MTW >= 1 && MTW * 32 * WM == BM, "row block must split into 32-row MFMA tiles per wave");
    static_assert(NTW >= 1 && NTW * 32 * WN == BN, "pixel tile must split into 32-pixel MFMA tiles per wave");
    static constexpr int NLOAD = 4;                                  // waves that stage (patch: waves 0-3; weights: the last four)
    static constexpr int KC = 16;
    // patch with halo: the transposed taps read offsets 0 and 1 only (input i - 1 and i), so their patch has ONE halo row / column
    static constexpr int HALO = (MODE == MODE_TCONV3A) ? 1 : 2;
    static constexpr int HP = PH + HALO, HW = PW + HALO, NSLOT = HP * HW;
    static constexpr int NS = PARTS * (PARTS + 1) / 2;               // products per k step
    static constexpr int W_UNITS = 3 * PARTS * 2 * BM;               // 16-byte units per weight stage (one kernel row)
    static constexpr int X_UNITS = PARTS * 2 * NSLOT;                // per patch buffer
    static constexpr int NXR = (NSLOT + 63) / 64;                    // patch pixels per lane
    static constexpr int LDS_BYTES = 16 * (WBUF * W_UNITS + 2 * X_UNITS);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS of one CU");
};

__host__ __device__ inline int64_t sp_packed_units(int mblocks, int cchunks, int bm, int parts) { return (int64_t)mblocks * cchunks * 9 * parts * 2 * bm; }

// round-to-nearest-even bf16 pair (v_cvt_pk_bf16_f32): low half = a, high half = b
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// a, b -> PARTS packed bf16 (F16: fp16, round to nearest even, v_cvt_pk_f16_f32) pairs whose sums reproduce a and b (residuals are
// exact in fp32)
template <int PARTS, int F16 = 0>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&out)[PARTS]) {
    if constexpr (F16) {
        static_assert(PARTS == 2, "the fp16 split has two pieces");
        const f32x2 v = {a, b};
        const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
        // residuals a - float(hi.lo), b - float(hi.hi) in ONE instruction each: v_fma_mix_f32 reads the fp16 half directly (hipcc emits
        // v_cvt_f32_f16 [+ SDWA] and v_sub_f32 for the plain expression: 6 instead of 4 VALU per pair in the issue-bound patch staging)
        f32x2 r;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(hi), "v"(a));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(hi), "v"(b));
        out[0] = hi; out[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    } else {
#pragma unroll
    for (int q = 0; q < PARTS; ++q) {
        const unsigned pk = pk_bf16(a, b);
        out[q] = pk;
        if (q + 1 < PARTS) { a -= __uint_as_float(pk << 16); b -= __uint_as_float(pk & 0xffff0000u); }
    }
    }
}
template <int F16>
__device__ __forceinline__ f32x16 sp_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// exact power-of-two scale that brings `bound` (> 0, finite) just below 2^15 (fp16 overflows at 65504): (scale, 1 / scale); (1, 1) for
// zero / non-finite bounds.  The exponent is kept inside +-126 so that neither factor leaves the normal fp32 range.
__device__ __forceinline__ void f16_scale(float bound, float& scale, float& unscale) {
    scale = 1.f; unscale = 1.f;
    if (bound > 0.f && bound < __builtin_inff()) {
        int e;
        frexpf(bound, &e);                                  // bound = m * 2^e, m in [0.5, 1)
        e = min(max(15 - e, -126), 126);
        scale = ldexpf(1.f, e); unscale = ldexpf(1.f, -e);
    }
}

// f16x3: per-row scales of the weights w [cout, rowlen]: scale[row] * max |w[row]| in [2^14, 2^15); rows >= cout: 1
__global__ void __launch_bounds__(256)
modconv_row_scale_kernel(const float* __restrict__ w, int cout, int rowlen, int rows_padded, float* __restrict__ scale, float* __restrict__ unscale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows_padded) return;
    float m = 0.f;
    if (row < cout) for (int i = lane; i < rowlen; i += 64) m = fmaxf(m, fabsf(w[(int64_t)row * rowlen + i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) { float sc, us; f16_scale(m, sc, us); scale[row] = sc; unscale[row] = us; }
}

// weights w [cout, cin, 3, 3] -> [mb][chunk][ky][kx][part][k half][co][8 channels] bf16 (zero padded)
template <int PARTS, int F16 = 0>
__global__ void __launch_bounds__(256)
modconv_pack_split_kernel(const float* __restrict__ w, int cout, int cin, int bm, int mblocks, int cchunks, u32x4* __restrict__ out,
                          const float* __restrict__ row_scale = nullptr) {
    const int64_t total = (int64_t)mblocks * cchunks * 9 * 2 * bm;   // one thread per (mb, chunk, tap, k half, co): all parts
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int64_t r = i;
        const int co_l = (int)(r % bm); r /= bm;
        const int kg = (int)(r % 2); r /= 2;
        const int tap = (int)(r % 9); r /= 9;
        const int cc = (int)(r % cchunks); r /= cchunks;
        const int mb = (int)r;
        const int co = mb * bm + co_l;
        const float rs = F16 ? row_scale[co] : 1.f;
        unsigned pk[4][PARTS];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ci = cc * 16 + kg * 8 + e * 2;
            const float a = (co < cout && ci < cin) ? w[((int64_t)co * cin + ci) * 9 + tap] * rs : 0.f;
            const float b = (co < cout && ci + 1 < cin) ? w[((int64_t)co * cin + ci + 1) * 9 + tap] * rs : 0.f;
            split_pair<PARTS, F16>(a, b, pk[e]);
        }
        const int64_t base = (((int64_t)(mb * cchunks + cc) * 9 + tap) * PARTS * 2 + kg) * bm + co_l;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) {
            const u32x4 v = {pk[0][q], pk[1][q], pk[2][q], pk[3][q]};
            out[base + (int64_t)q * 2 * bm] = v;
        }
    }
}

template <int OFF_BYTES>
__device__ __forceinline__ u32x4 lds_read128_async(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF_BYTES));
    return v;
}
template <int PENDING>
__device__ __forceinline__ void lds_wait128(u32x4& first) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(first) : "n"(PENDING)); }
__device__ __forceinline__ void lds_pin128(u32x4& v) { asm volatile("" : "+v"(v)); }

// PAR (all-class transposed form): -1 = all four output classes, three stages (kernel rows) per chunk; 0 / 1 = the two column classes of the even /
// odd output rows only: kernel rows {0, 2} / {1} — out[2 i + ky] += x[i] w[ky] — i.e. two stages / one stage per chunk on twice the positions
template <int MODE, int BIG, int PH, int PARTS, int WBUF, int NWV, int F16, int TEAMS = 1, int PAR = -1>
__device__ __forceinline__ void modconv_split_tile(const ide3d_modconv_params& p, const u32x4* __restrict__ wp, float* __restrict__ partial,
                                                   const ConvGeom& g, unsigned char* smem, int mb, int tl, int grp, int split, int tiles_x,
                                                   const float* __restrict__ row_unscale, float* wg_scratch = nullptr) {
    using K = SpCfg<MODE, BIG, PH, PARTS, WBUF, NWV, (PAR >= 0)>;
    static_assert(!F16 || PARTS == 2, "f16x3 = two fp16 pieces per operand");
    static_assert(PAR < 0 || (MODE == MODE_TCONV3A && WBUF == 2 && TEAMS == 1), "row-parity pairs: transposed form, two weight buffers, one team");
    // stages (kernel rows) of a chunk, in order
    constexpr int NST = (PAR < 0) ? 3 : (PAR == 0 ? 2 : 1);
    auto kys = [](int i) constexpr { return (PAR < 0) ? i : (PAR == 0 ? 2 * i : 1); };
    static_assert(TEAMS == 1 || (TEAMS == 2 && NWV == 4), "two teams of four waves, or one team");
    constexpr int PW = K::PW;
    // TEAMS = 2: two 4-wave teams in one 8-wave workgroup, each with its own tile (adjacent output-channel blocks of the same pixels, so
    // both run the same number of stages) and its own LDS; the workgroup barriers couple them, the code below is per team
    const int tid = (TEAMS == 2) ? (int)(threadIdx.x & 255u) : (int)threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / K::WN, wn = wid % K::WN;
    const int half = lane >> 5, l32 = lane & 31;
    const int lp = ((((l32 >> 2) ^ (l32 >> 3) ^ (l32 >> 4)) & 1) << 4) | (l32 & 15);     // pixel of this lane inside an N tile
    const int txi = tl % tiles_x, tyi = tl / tiles_x;
    const int y0 = tyi * PH, x0 = txi * PW;
    const int n0 = grp;                                                                 // one image per tile
    u32x4* const s_w = reinterpret_cast<u32x4*>(smem);                                  // [WBUF][W_UNITS]
    u32x4* const s_x = s_w + WBUF * K::W_UNITS;                                         // [2][X_UNITS]

    f32x16 acc[K::NCLS][K::MTW][K::NTW];
#pragma unroll
    for (int q = 0; q < K::NCLS; ++q)
#pragma unroll
        for (int i = 0; i < K::MTW; ++i)
#pragma unroll
            for (int j = 0; j < K::NTW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;

    // ---- patch staging plan: wave `wid` owns channel quad `wid` of the chunk; lane -> patch pixels lane + 64 r ----
    const int hw = p.h * p.w_;
    int x_src[K::NXR];                       // offset of the pixel inside a channel plane, -1 = outside the image / the patch
#pragma unroll
    for (int r = 0; r < K::NXR; ++r) {
        const int sidx = lane + 64 * r;
        const int ry = sidx / K::HW, rx = sidx % K::HW;
        const int yy = y0 - 1 + ry, xx = x0 - 1 + rx;
        x_src[r] = (sidx < K::NSLOT && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w_) ? yy * p.w_ + xx : -1;
    }
    const float* __restrict__ ximg = p.x + (int64_t)n0 * p.cin * hw;
    float xreg[K::NXR][4];
    float sty[4];
    // f16x3: scale of this image's patch (exact power of two, joins the styles when a patch is committed)
    float xs = 1.f, xus = 1.f;
    if constexpr (F16) {
        // every wave for itself (no LDS, no barrier, no extra launch): 64 lanes stride over the cin styles, lanes 0-31 read the amax
        // slots; the result is first needed when the first patch is committed, i.e. behind that patch's own global loads
        float mm = p.styles ? 0.f : 1.f;
        if (p.styles) for (int ci = lane; ci < p.cin; ci += 64) mm = fmaxf(mm, fabsf(p.styles[(int64_t)n0 * p.cin + ci]));
        float xm = (lane < IDE3D_AMAX_SLOTS) ? p.x_amax[(int64_t)n0 * IDE3D_AMAX_FLOATS + lane * IDE3D_AMAX_STRIDE] : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mm = fmaxf(mm, __shfl_xor(mm, off)); xm = fmaxf(xm, __shfl_xor(xm, off)); }
        f16_scale(mm * xm, xs, xus);
    }
    const bool stages_patch = (NWV == 4) || wid < 4;
    auto fetch_patch = [&](int c) {
        if (!stages_patch) return;
        const int ci0 = c * K::KC + wid * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ci = min(ci0 + k, p.cin - 1);
            sty[k] = (ci0 + k < p.cin) ? (p.styles ? p.styles[(int64_t)n0 * p.cin + ci] : 1.f) : 0.f;
#pragma unroll
            for (int r = 0; r < K::NXR; ++r) xreg[r][k] = ximg[(int64_t)ci * hw + max(x_src[r], 0)];
        }
    };
    auto commit_patch = [&](int buf) {
        if (!stages_patch) return;
        unsigned char* const dst = reinterpret_cast<unsigned char*>(s_x + buf * K::X_UNITS) + ((wid >> 1) * K::NSLOT) * 16 + (wid & 1) * 8;
        // f16x3: the image scale joins the styles HERE, where the (scalar) style loads are waited for anyway — multiplied in fetch_patch it
        // forces `s_waitcnt lgkmcnt(0)` right behind the loads, in the middle of the hand-counted operand-read pipeline (scalar loads and
        // LDS reads share that counter): measured 10 - 30 % per launch
        const float sc[4] = {F16 ? sty[0] * xs : sty[0], F16 ? sty[1] * xs : sty[1], F16 ? sty[2] * xs : sty[2], F16 ? sty[3] * xs : sty[3]};
#pragma unroll
        for (int r = 0; r < K::NXR; ++r) {
            const int sidx = lane + 64 * r;
            const bool live = x_src[r] >= 0;
            unsigned lo[PARTS], hi[PARTS];
            split_pair<PARTS, F16>(live ? xreg[r][0] * sc[0] : 0.f, live ? xreg[r][1] * sc[1] : 0.f, lo);
            split_pair<PARTS, F16>(live ? xreg[r][2] * sc[2] : 0.f, live ? xreg[r][3] * sc[3] : 0.f, hi);
            if (sidx < K::NSLOT) {
#pragma unroll
                for (int q = 0; q < PARTS; ++q) {
                    const u32x2 v = {lo[q], hi[q]};
                    *reinterpret_cast<u32x2*>(dst + (q * 2 * K::NSLOT + sidx) * 16) = v;
                }
            }
        }
    };
    // ---- weight stage (chunk c, kernel row ky): one contiguous slab, already the LDS image ----
    const u32x4* __restrict__ wsrc = wp + (int64_t)mb * g.cchunks * 3 * K::W_UNITS;
    constexpr int W_PIECES = K::W_UNITS / 64;                        // 1 KB pieces (64 lanes x 16 bytes)
    static_assert(K::W_UNITS % 64 == 0, "weight stage is a whole number of 1 KB pieces");
    // An LDS-DMA piece costs the issuing wave 200-350 cycles (the stream runs at ~3-5 B/clk per issuing wave), so the 8-wave form
    // splits the pieces in time: waves 4-7 issue the first W_EARLY of their column before they multiply (waves 0-3 multiply
    // meanwhile), waves 0-3 issue the rest after their MFMAs and the patch commit (waves 4-7 multiply meanwhile).
    constexpr int W_COL = (W_PIECES + 3) / 4;                        // pieces per wave column (wid & 3)
    constexpr int W_EARLY = (NWV == 8) ? (W_COL * 5 + 8) / 9 : W_COL;
    auto fetch_weights = [&](int stage, int buf, bool late = false) {
        const u32x4* src = wsrc + (int64_t)stage * K::W_UNITS;
        if (NWV == 4 && late) return;
        if (NWV == 8 && (late != (wid < 4))) return;
        const int m0 = late ? W_EARLY : 0, m1 = late ? W_COL : W_EARLY;
        for (int m = m0; m < m1; ++m) {
            const int i = (wid & 3) + 4 * m;
            if (i < W_PIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 64 + lane),
                                                 (__attribute__((address_space(3))) void*)(s_w + buf * K::W_UNITS + i * 64), 16, 0, 0);
        }
    };

    // ---- operand addresses ----
    const unsigned a_base0 = (unsigned)(size_t)(s_w + half * K::BM + wm * K::MTW * 32 + l32);
    unsigned b_base0[K::NTW];
#pragma unroll
    for (int j = 0; j < K::NTW; ++j) {
        const int pix = (wn * K::NTW + j) * 32 + lp;
        b_base0[j] = (unsigned)(size_t)(s_x + half * K::NSLOT + (pix / PW) * K::HW + (pix % PW));
    }

    const int c_begin = split * g.chunks_per_split;
    const int c_end = min(c_begin + g.chunks_per_split, g.cchunks);
#ifdef IDE3D_MC_TRACE
    unsigned long long mc_acc[6] = {0, 0, 0, 0, 0, 0}, mc_last = 0;
    const unsigned long long mc_t0 = __builtin_readcyclecounter();
#endif
    if (c_begin < c_end) {
        if (WBUF == 2) { fetch_weights(c_begin * 3 + kys(0), 0); if (NWV == 8) fetch_weights(c_begin * 3 + kys(0), 0, true); }
        fetch_patch(c_begin);
        commit_patch(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int wbuf = 0;
#ifdef IDE3D_MC_TRACE
    mc_last = __builtin_readcyclecounter();
    const unsigned long long mc_prologue = mc_last - mc_t0;
#endif
    for (int c = c_begin; c < c_end; ++c) {
        const int xbuf = (c - c_begin) & 1;
        const bool more = c + 1 < c_end;
        static_for<NST>([&](auto sii) {
            constexpr int SI = decltype(sii)::value, KY = kys(SI);
            constexpr bool LAST = (SI == NST - 1), MID = (SI == (NST == 3 ? 1 : 0));          // MID: the stage behind which the next chunk's patch is loaded
            const int next_stage = LAST ? (c + 1) * 3 + kys(0) : c * 3 + kys(LAST ? 0 : SI + 1);
            IDE3D_MC_TS(0)
            if constexpr (WBUF == 2) {
                if (!LAST || more) fetch_weights(next_stage, wbuf ^ 1);
                if (MID && more) fetch_patch(c + 1);
            } else {
                // one weight buffer: this stage's slab is fetched now (everybody left the previous stage at the barrier below); the
                // other workgroup of the CU multiplies meanwhile
                fetch_weights(c * 3 + KY, 0);
                if (MID && more) fetch_patch(c + 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            IDE3D_MC_TS(1)
            // ---- MFMAs of kernel row KY: taps pipelined (operands of tap kx + 1 in flight while tap kx multiplies) ----
            const unsigned a_base = a_base0 + ((WBUF == 2) ? wbuf * (K::W_UNITS * 16) : 0);
            unsigned b_base[K::NTW];
#pragma unroll
            for (int j = 0; j < K::NTW; ++j) b_base[j] = b_base0[j] + xbuf * (K::X_UNITS * 16);
            // Software pipeline over (tap, piece) groups: the A and B pieces q of tap kx are one group of MTW + NTW 16-byte reads;
            // group s + 1 is in flight while the products that group s completes - (qa, qb) with max(qa, qb) = q and
            // qa + qb < PARTS - multiply.  At most two groups (<= 12 reads) are outstanding: lgkmcnt is a 4-bit counter.
            // AHEAD groups in flight (SP_AHEAD = 1: the round-2 pipeline).  Round 4 measured 2 (bf16x6; f16x3 with
            // a third tap buffer): 338 vs 330 us at 128 -> 128 @256, 255 vs 251 at the transposed 256 -> 128, 218 vs 211 in f16x3 - with two
            // waves per SIMD the LDS latency is already covered, more reads in flight only delay the first product.  The operand registers of a tap are reused by the tap
            // two taps later (NBUF = 2) or three (NBUF = 3): group s + AHEAD may only be issued once every product of the tap it overwrites
            // has been ISSUED at least one group earlier, i.e. AHEAD <= (NBUF - 1) * PARTS - 1; and (AHEAD + 1) * GRP <= 15 reads outstanding.
            constexpr int GRP = K::MTW + K::NTW, NGRP = 3 * PARTS;
            constexpr int SP_AHEAD = 1;
            constexpr int NBUF = (PARTS == 2 && SP_AHEAD > 1) ? 3 : 2;
            constexpr int AHEAD_CAP = 15 / GRP - 1, AHEAD_SAFE = (NBUF - 1) * PARTS - 1;
            constexpr int AHEAD = (SP_AHEAD < AHEAD_CAP ? SP_AHEAD : AHEAD_CAP) < AHEAD_SAFE ? (SP_AHEAD < AHEAD_CAP ? SP_AHEAD : AHEAD_CAP) : AHEAD_SAFE;
            static_assert(AHEAD >= 1 && (AHEAD + 1) * GRP <= 15, "lgkmcnt is a 4-bit counter");
            // All-class transposed form: the patch offset of tap (ky, kx) is ((ky == 2) ? 0 : 1, (kx == 2) ? 0 : 1), i.e. the taps kx = 0 and
            // kx = 1 of a kernel row multiply the SAME patch fragment: it is read once (slot 0; kx = 2 reads into slot 1), which takes a third
            // of the patch reads out of the LDS-bound loop
            constexpr bool B_REUSE = (MODE == MODE_TCONV3A) && NBUF == 2;
            u32x4 av[NBUF][K::MTW][PARTS], bv[NBUF][K::NTW][PARTS];
            auto issue = [&](auto ss) {
                constexpr int S = decltype(ss)::value, KX = S / PARTS, Q = S % PARTS, B = KX % NBUF, T = KY * 3 + KX;
                constexpr int BS = B_REUSE ? (KX == 2 ? 1 : 0) : B;
                static_for<K::MTW>([&](auto ii) {
                    constexpr int I = decltype(ii)::value;
                    av[B][I][Q] = lds_read128_async<(((KX * PARTS + Q) * 2) * K::BM + I * 32) * 16>(a_base);
                });
                if constexpr (!(B_REUSE && KX == 1))
                static_for<K::NTW>([&](auto jj) {
                    constexpr int J = decltype(jj)::value;
                    bv[BS][J][Q] = lds_read128_async<(Q * 2 * K::NSLOT + tap_patch_offset<MODE, K::HW>(T)) * 16>(b_base[J]);
                });
            };
            static_for<AHEAD>([&](auto ss) { issue(ss); });
            static_for<NGRP>([&](auto ss) {
                constexpr int S = decltype(ss)::value, KX = S / PARTS, Q = S % PARTS, B = KX % NBUF, QC = (PAR >= 0) ? (KX & 1) : tap_class<MODE>(KY * 3 + KX);
                constexpr int BS = B_REUSE ? (KX == 2 ? 1 : 0) : B;
                if constexpr (S + AHEAD < NGRP) issue(std::integral_constant<int, S + AHEAD>{});
                // reads issued after group S's own: those of the next min(AHEAD, groups left) groups
                constexpr int PENDING = [] { int n = 0; for (int s2 = S + 1; s2 <= S + AHEAD && s2 < NGRP; ++s2) n += K::MTW + ((B_REUSE && s2 / PARTS == 1) ? 0 : K::NTW); return n; }();
                lds_wait128<PENDING>(av[B][0][Q]);
#pragma unroll
                for (int i = 0; i < K::MTW; ++i) lds_pin128(av[B][i][Q]);
                if constexpr (!(B_REUSE && KX == 1)) {
#pragma unroll
                for (int j = 0; j < K::NTW; ++j) lds_pin128(bv[BS][j][Q]);
                }
#pragma unroll
                for (int qa = 0; qa <= Q; ++qa)
#pragma unroll
                    for (int qb = 0; qb <= Q; ++qb) {
                        if ((qa == Q || qb == Q) && qa + qb < PARTS) {
#pragma unroll
                            for (int i = 0; i < K::MTW; ++i)
#pragma unroll
                                for (int j = 0; j < K::NTW; ++j)
                                    acc[QC][i][j] = sp_mfma<F16>(av[B][i][qa], bv[BS][j][qb], acc[QC][i][j]);
                        }
                    }
            });
            IDE3D_MC_TS(2)
            if (LAST && more) commit_patch(xbuf ^ 1);
            if constexpr (WBUF == 2 && NWV == 8) { if (!LAST || more) fetch_weights(next_stage, wbuf ^ 1, true); }
            IDE3D_MC_TS(3)
            if (WBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            IDE3D_MC_TS(4)
            __syncthreads();
            IDE3D_MC_TS(5)
            wbuf ^= 1;
        });
    }
#ifdef IDE3D_MC_TRACE
    const unsigned long long mc_t1 = __builtin_readcyclecounter();
#endif
    modconv_finish<MODE, 1, PH, PW, K::NWV, K::NCLS, K::MTW, K::NTW, K::BM, K::LDS_BYTES / 4, PAR>(p, partial, g, acc, reinterpret_cast<float*>(smem), mb, n0, y0, x0, split, 0, wm, wn, lp,
                                                                                                F16 ? row_unscale : nullptr, xus, (TEAMS == 2) ? tid : -1, wg_scratch);
#ifdef IDE3D_MC_TRACE
    if (blockIdx.x == 100 && (threadIdx.x == 0 || threadIdx.x == 256)) {          // wave 0 (and wave 4 of an 8-wave workgroup, at [16..])
        const int o = threadIdx.x ? 16 : 0;
        for (int k = 0; k < 6; ++k) g_mc_dbg[o + k] = mc_acc[k];
        g_mc_dbg[o + 7] = (unsigned long long)(c_end - c_begin) * 3;
        g_mc_dbg[o + 8] = mc_prologue;
        g_mc_dbg[o + 9] = __builtin_readcyclecounter() - mc_t1;
    }
#endif
}



// ------------------------------------------------------------------------------------------------
// per-image 1x1 heads on the split-bf16 arithmetic
// ------------------------------------------------------------------------------------------------
// The dual toRGB + toSeg heads (per-image folded weights [n, O, C], bias, clamp; networks.py:1109,1130) are fp32-MFMA bound on the
// loop above (56 % pipe busy at O = 192: 2.9x their HBM time).  Here a workgroup owns ALL O rows (MT x 32, MT = 1 or 6) of 128
// consecutive pixels, so the activations are read once: wave w takes the B operand of its 32 pixels straight from global memory
// (lane = pixel, 8 channels of its k half = 8 coalesced dword loads), splits it into bf16 pieces in registers and multiplies it
// with the A fragments of all MT row tiles, which the four waves share through LDS (pre-split packed weights [chunk][piece][k half]
// [row][8], double-buffered LDS-DMA).  The accumulator layout has lane = pixel, so every store instruction writes 128-byte runs.
// Exclusive residency (see kSpExclusive below): 8 waves = two of this workgroup's waves per SIMD, 256 pixels per workgroup, one workgroup
// per CU; every wave stays until the barrier behind the K loop.
// MT = 1 (<= 32 outputs: HBM-bound, ~100 registers): 16 waves of 128 registers, four per SIMD, 512 pixels per workgroup - the loads of twice
// as many waves in flight (116 -> see profiles/round4 at 64 -> 22 @512 with 8 waves; the 4-wave form of round 3 ran two workgroups per CU).
constexpr int head_waves(int mt) { return mt == 1 ? 16 : 8; }
template <int PARTS, int MT>
struct HeadCfg {
    static constexpr int ROWS = MT * 32;
    static constexpr int A_UNITS = PARTS * 2 * ROWS;                  // 16-byte units per 16-channel chunk
    static constexpr int NWV = head_waves(MT);
    static constexpr int BN = 32 * NWV;
};
__host__ __device__ inline int64_t head_packed_units(int n, int cchunks, int parts, int mt) { return (int64_t)n * cchunks * parts * 2 * mt * 32; }

template <int PARTS>
__global__ void __launch_bounds__(256)
head_pack_split_kernel(const float* __restrict__ w, int64_t w_batch_stride, int n, int cout, int cin, int rows, int cchunks, u32x4* __restrict__ out) {
    const int64_t total = (int64_t)n * cchunks * 2 * rows;            // one thread per (image, chunk, k half, row): all pieces
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int64_t r = i;
        const int row = (int)(r % rows); r /= rows;
        const int kg = (int)(r % 2); r /= 2;
        const int cc = (int)(r % cchunks); r /= cchunks;
        const int nb = (int)r;
        unsigned pk[4][PARTS];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ci = cc * 16 + kg * 8 + e * 2;
            const float a = (row < cout && ci < cin) ? w[nb * w_batch_stride + (int64_t)row * cin + ci] : 0.f;
            const float b = (row < cout && ci + 1 < cin) ? w[nb * w_batch_stride + (int64_t)row * cin + ci + 1] : 0.f;
            split_pair<PARTS>(a, b, pk[e]);
        }
        const int64_t base = (((int64_t)nb * cchunks + cc) * PARTS * 2 + kg) * rows + row;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) {
            const u32x4 v = {pk[0][q], pk[1][q], pk[2][q], pk[3][q]};
            out[base + (int64_t)q * 2 * rows] = v;
        }
    }
}

template <int PARTS, int MT>
__global__ void __launch_bounds__(64 * head_waves(MT), head_waves(MT) / 4)        // 2 (4) waves per SIMD, all of this workgroup, 256 (128) registers each
head_split_kernel(ide3d_modconv_params p, const u32x4* __restrict__ wp, int cchunks, int tiles_per_image) {
    using K = HeadCfg<PARTS, MT>;
    if constexpr (K::NWV == 16) asm volatile("" ::: "v127"); else asm volatile("" ::: "v255");      // the workgroup's waves hold their SIMDs' whole register file: no foreign wave fits
    constexpr int S_MIN = K::NWV * 32 * 40 / 4 + 2 * K::ROWS / 4 + 16;       // the epilogue's staging tiles: 32 rows x (32 + 8) floats per wave
    constexpr int S_UNITS = (2 * K::A_UNITS > S_MIN) ? 2 * K::A_UNITS : S_MIN;
    __shared__ __attribute__((aligned(16))) u32x4 s_a[S_UNITS];              // weights of chunk c and c + 1 (L2-resident, one chunk of look-ahead)
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = bid / tiles_per_image, tile = bid % tiles_per_image;
    const int hw = p.h * p.w_;
    const int pix = tile * K::BN + wid * 32 + l32;
    const int pix_c = min(pix, hw - 1);
    const float* __restrict__ xl = p.x + (int64_t)n0 * p.cin * hw + (int64_t)(8 * half) * hw + pix_c;     // this lane's k half, channel 0 of chunk 0
    const u32x4* __restrict__ wsrc = wp + (int64_t)n0 * cchunks * K::A_UNITS;

    f32x16 acc[1][MT][1];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][m][0][r] = 0.f;

    constexpr int A_PIECES = K::A_UNITS / 64;
    static_assert(K::A_UNITS % 64 == 0, "whole 1 KB pieces");
    auto fetch_a = [&](int c, int buf) {
        for (int i = wid; i < A_PIECES; i += K::NWV)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (int64_t)c * K::A_UNITS + i * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(s_a + buf * K::A_UNITS + i * 64), 16, 0, 0);
    };
    // B operand ring: three register slots (chunk c in use, c + 1 and c + 2 in flight); the chunk loop is unrolled by three so that the
    // slot of every access is a compile-time constant (a run-time slot index sends the ring to scratch memory)
    float xr[3][8];
    auto fetch_b = [&](int c, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = c * 16 + 8 * half + e;
            const float v = xl[(int64_t)(min(ch, p.cin - 1) - 8 * half) * hw];      // clamped address, zero beyond cin (like the packed weights)
            xr[SLOT][e] = (ch < p.cin) ? v : 0.f;
        }
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;
    // issue order: A0 B0 B1 | A1 B2 | A2 B3 | ...: the activations (HBM) run two chunks ahead in registers, the weights one chunk ahead in LDS
    fetch_a(0, 0); fetch_b(0, S0{});
    if (cchunks > 1) fetch_b(1, S1{});
    auto chunk = [&](int c, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value, NSLOT = (SLOT + 2) % 3;
        // A(c) and B(c) have landed once at most the 8 loads of B(c + 1), issued after them, are outstanding
        if (c + 1 < cchunks) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                     // ... for every wave; and nobody still reads the buffer refilled below
        if (c + 1 < cchunks) fetch_a(c + 1, (c + 1) & 1);
        if (c + 2 < cchunks) fetch_b(c + 2, std::integral_constant<int, NSLOT>{});
        u32x4 bfrag[PARTS];
        {
            unsigned pk[4][PARTS];
#pragma unroll
            for (int e = 0; e < 4; ++e) split_pair<PARTS>(xr[SLOT][2 * e], xr[SLOT][2 * e + 1], pk[e]);
#pragma unroll
            for (int q = 0; q < PARTS; ++q) bfrag[q] = u32x4{pk[0][q], pk[1][q], pk[2][q], pk[3][q]};
        }
        const u32x4* sa = s_a + (c & 1) * K::A_UNITS + half * K::ROWS + l32;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            u32x4 af[PARTS];
#pragma unroll
            for (int q = 0; q < PARTS; ++q) af[q] = sa[q * 2 * K::ROWS + m * 32];
#pragma unroll
            for (int qa = 0; qa < PARTS; ++qa)
#pragma unroll
                for (int qb = 0; qa + qb < PARTS; ++qb)
                    acc[0][m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[qa]), __builtin_bit_cast(bf16x8, bfrag[qb]), acc[0][m][0], 0, 0, 0);
        }
    };
    for (int c = 0; c < cchunks; c += 3) {
        chunk(c, S0{});
        if (c + 1 < cchunks) chunk(c + 1, S1{});
        if (c + 2 < cchunks) chunk(c + 2, S2{});
    }
    // epilogue: the shared one (bias, gain, clamp; rows staged through LDS so that a lane stores 16 bytes of one row) on the flattened
    // image: one row of h * w pixels, this workgroup's 128-pixel run
    __syncthreads();                                                         // every wave has finished reading the weight ring
    ide3d_modconv_params pf = p;
    pf.h = 1; pf.w_ = hw;
    ConvGeom g{};
    g.oh = 1; g.ow = hw; g.split_k = 1;
    modconv_finish<MODE_CONV1, 1, 1, K::BN, K::NWV, 1, MT, 1, K::ROWS, S_UNITS * 4>(pf, nullptr, g, acc, reinterpret_cast<float*>(s_a), 0, n0, 0, tile * K::BN, 0, 0, 0, wid, l32);
}


// Resident-weights form of the heads (round 4).  head_split_kernel lives for ONE 256- / 512-pixel tile: with one workgroup per CU (exclusive
// residency) the launch is `tiles / 256` rounds of load -> multiply -> store in lockstep over the whole chip, the phases of a round do not
// overlap (128 -> 192 @256: 4 rounds of ~28 us where the HBM bytes of a round take 10 and its MFMAs 8), and the barrier per K chunk keeps
// the waves of a workgroup in the same phase as well.  Here the packed weights of the image's WHOLE K (<= 128 channels: 144 KB at 192 rows,
// 24 KB at 32) are copied to LDS once per workgroup, and then the waves are independent: each takes 32-pixel tiles of its image in a
// grid-stride loop with no barrier, loads the tile's activations straight into registers (lane = pixel, 8 channels of its k half per
// chunk), multiplies against the LDS fragments and stores the accumulators directly (lane = pixel: every store writes two 128-byte runs).
// MT = 6 (MFMA-heavy): the two waves of a SIMD alternate between waiting for HBM and multiplying.  MT = 1 (HBM-bound): the loads of the
// NEXT tile are issued before the current one is multiplied (second register buffer), so a wave always has a tile in flight.
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* ptr) {
    const uint64_t v = reinterpret_cast<uint64_t>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}
template <int PARTS, int MT, int NCH>
__global__ void __launch_bounds__(512, 2)
head_resident_kernel(ide3d_modconv_params p, const u32x4* __restrict__ wp, int wgs_per_image, int tiles32) {
    using K = HeadCfg<PARTS, MT>;
    asm volatile("" ::: "v255");                                  // two waves per SIMD, both of this workgroup (section 4.2)
    constexpr int NWV = 8, NBUF = (MT == 1) ? 2 : 1;
    __shared__ __attribute__((aligned(16))) u32x4 s_a[NCH * K::A_UNITS + K::ROWS / 4 + 4];
    float* const s_bi = reinterpret_cast<float*>(s_a + NCH * K::A_UNITS);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x % p.n, slot = blockIdx.x / p.n;      // consecutive workgroups (= XCDs) take different images
    const int hw = p.h * p.w_;
    const u32x4* __restrict__ wsrc = wp + (int64_t)n0 * NCH * K::A_UNITS;
    constexpr int PIECES = NCH * K::A_UNITS / 64;
    for (int i = wid; i < PIECES; i += NWV)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + i * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(s_a + i * 64), 16, 0, 0);
    if (tid < K::ROWS) s_bi[tid] = (p.bias && tid < p.cout) ? p.bias[tid] : 0.f;      // (16-byte aligned: A_UNITS units of 16 bytes precede it)

    // addresses as (wave-uniform 64-bit base in SGPRs) + (one 32-bit lane offset): 64 - 128 loads and 16 MT stores per tile with their own
    // 64-bit lane addresses would not fit the register file
    const char* const xbase = reinterpret_cast<const char*>(p.x + (int64_t)n0 * p.cin * hw);
    char* const ybase = reinterpret_cast<char*>(p.y + (int64_t)n0 * p.cout * hw);
    const unsigned x_lane = (unsigned)((8 * half) * hw + l32) * 4u, y_lane = (unsigned)((4 * half) * hw + l32) * 4u;
    const int64_t plane = (int64_t)hw * 4;
    const int wave0 = slot * NWV + wid, nwaves = wgs_per_image * NWV;
    const float e_gain = p.gain, e_clamp = (p.clamp >= 0.f) ? p.clamp : __builtin_inff();
    float amax = 0.f;

    float xr[NBUF][NCH][8];
    auto fetch = [&](int t, auto buf_tag) {
        constexpr int B = decltype(buf_tag)::value;
        // (readfirstlane: the base stays a scalar the loop recomputes per tile; left alone the compiler turns the 64 - 128 addresses into as
        // many 64-bit induction variables in vector registers and spills)
        const char* src = uniform_ptr(xbase + (int64_t)t * 128);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xr[B][c][e] = *reinterpret_cast<const float*>(src + (c * 16 + e) * plane + x_lane);
            }
    };
    // The accumulators start from the bias (rows m * 32 + 8 g + 4 half + 0..3 of register group g: one 16-byte LDS read), so the finish is
    // gain, clamp, store: the waves of a SIMD share one VALU, and with 96 values per lane and tile every instruction there counts
    // (no loads / no stores / one product still took 44 of 106 us at 128 -> 192 @256 with six VALU instructions per value).
    typedef float f32x4a __attribute__((ext_vector_type(4)));
    const float* const bi = s_bi + 4 * half;
    f32x16 acc[MT];
    auto compute = [&](auto buf_tag) {
        constexpr int B = decltype(buf_tag)::value;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4a b4 = *reinterpret_cast<const f32x4a*>(bi + m * 32 + 8 * g);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[m][4 * g + k] = b4[k];
            }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            u32x4 bfrag[PARTS];
            {
                unsigned pk[4][PARTS];
#pragma unroll
                for (int e = 0; e < 4; ++e) split_pair<PARTS>(xr[B][c][2 * e], xr[B][c][2 * e + 1], pk[e]);
#pragma unroll
                for (int q = 0; q < PARTS; ++q) bfrag[q] = u32x4{pk[0][q], pk[1][q], pk[2][q], pk[3][q]};
            }
            const u32x4* sa = s_a + c * K::A_UNITS + half * K::ROWS + l32;
            constexpr int MG = (MT % 2 == 0) ? 2 : 1;               // row tiles multiplied together: two independent accumulator chains
#pragma unroll
            for (int m0 = 0; m0 < MT; m0 += MG) {
                u32x4 af[MG][PARTS];
#pragma unroll
                for (int g = 0; g < MG; ++g)
#pragma unroll
                    for (int q = 0; q < PARTS; ++q) af[g][q] = sa[q * 2 * K::ROWS + (m0 + g) * 32];
                constexpr int NQA = PARTS;
#pragma unroll
                for (int qa = 0; qa < NQA; ++qa)
#pragma unroll
                    for (int qb = 0; qa + qb < NQA; ++qb)
#pragma unroll
                        for (int g = 0; g < MG; ++g)
                            acc[m0 + g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[g][qa]), __builtin_bit_cast(bf16x8, bfrag[qb]), acc[m0 + g], 0, 0, 0);
            }
        }
    };
    // finish and store: register r of row tile m = row m * 32 + 8 (r / 4) + 4 half + r % 4; lane = pixel
    const bool all_rows = (p.cout == K::ROWS), unit_gain = (e_gain == 1.f), want_amax = (p.y_amax != nullptr);
    auto finish = [&](int t) {
        char* dst = uniform_ptr(ybase + (int64_t)t * 128);
        auto rows = [&](auto all_tag, auto gain_tag) {
            constexpr bool ALL = decltype(all_tag)::value, UNIT = decltype(gain_tag)::value;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row_u = m * 32 + 8 * (r >> 2) + (r & 3);               // + 4 half
                    if (ALL || row_u + 4 * half < p.cout) {
                        float v = acc[m][r];
                        if (!UNIT) v *= e_gain;
                        v = __builtin_amdgcn_fmed3f(v, -e_clamp, e_clamp);
                        *reinterpret_cast<float*>(dst + row_u * plane + y_lane) = v;
                        if (want_amax) amax_acc(amax, v);
                    }
                }
        };
        if (all_rows) { if (unit_gain) rows(std::true_type{}, std::true_type{}); else rows(std::true_type{}, std::false_type{}); }
        else          { if (unit_gain) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{}); }
    };
    using B0 = std::integral_constant<int, 0>; using B1 = std::integral_constant<int, NBUF - 1>;
    if (wave0 < tiles32) fetch(wave0, B0{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the weights (LDS-DMA: invisible to the compiler's own wait counting)
    __syncthreads();
    if constexpr (NBUF == 1) {
        // the next tile's loads go out between the last product and the finish: the register buffer is free, and the loads run ahead of the
        // 96 stores instead of queueing behind them.  (Measured without effect: starting the second wave of every SIMD 1 / 2 / 4 x 8128
        // cycles late to break a chip-wide load / multiply / store lockstep: 105.5 / 104.7 / 113.8 us against 104.8.)
        for (int t = wave0; t < tiles32; t += nwaves) {
            compute(B0{});
            if (t + nwaves < tiles32) fetch(t + nwaves, B0{});
            finish(t);
        }
    } else {
        for (int t = wave0; t < tiles32; t += 2 * nwaves) {
            if (t + nwaves < tiles32) fetch(t + nwaves, B1{});
            compute(B0{});
            finish(t);
            if (t + nwaves < tiles32) {
                if (t + 2 * nwaves < tiles32) fetch(t + 2 * nwaves, B0{});
                compute(B1{});
                finish(t + nwaves);
            }
        }
    }
    // every wave stays until the last one has finished (exclusive residency), and the image's amax is raised once per workgroup
    __syncthreads();
    if (p.y_amax != nullptr) amax_raise_block(p.y_amax, n0, amax, reinterpret_cast<float*>(s_a));
}


// The heads of the LOW-resolution blocks (4^2 .. 16^2: 512 -> 192 per image, 0.01 - 0.2 GFLOP): on the matrix loops each cost a weight-packing
// launch, a split-K main launch of a few dozen workgroups and a reduction launch — 20-22 us of latency for microseconds of work, three times
// per frame.  Here: one launch of plain fp32 FMAs (exact products).  A workgroup owns HS_RB output rows of PX pixels of one image; its 16
// waves are PX / 64 pixel groups x 1024 / PX slices of the input channels; a lane loads its pixel of every channel of its slice (coalesced) and multiplies
// it with the rows' weights, which are wave-uniform (scalar loads straight from the folded [n, cout, cin] weights: no packing); the slices
// meet in LDS, and thread (row, pixel) adds them, applies bias / gain / clamp and stores.
// PX = pixels per workgroup: 256 (4 channel slices), or 64 for maps of <= 64 pixels (16 slices: a lane's whole slice is one batch of loads).
// These launches are latency chains: 64 pixels x 16 slices with a lane's 32 loads in ONE batch: 5.3 us at 4^2 / 8^2 (8 per batch, 256 pixels: 11-12);
// 256 pixels x 4 slices, 8 loads per batch (33 registers: three workgroups per CU): 12.3 us at 16^2 (32 per batch: 13.2, and 35 instead of 27 at 32^2,
// where the matrix loop's 27.5 us stay).
constexpr int HS_RB = 4, HS_THREADS = 1024;
template <int PX, int HS_UN>
__global__ void __launch_bounds__(HS_THREADS)
head_small_kernel(ide3d_modconv_params p, int hw) {
    constexpr int KS = HS_THREADS / PX;
    __shared__ float s_red[KS][HS_RB][PX];
    __shared__ float s_am[HS_THREADS / 64];
    const int px_l = threadIdx.x % PX;
    const int ks = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / PX));          // channel slice: uniform per wave (PX is a multiple of 64)
    const int row0 = blockIdx.x * HS_RB, n = blockIdx.z;
    const int px_c = min((int)blockIdx.y * PX + px_l, hw - 1);
    const int cq = (p.cin + KS - 1) / KS, c0 = ks * cq, c1 = min(c0 + cq, p.cin);
    const float* __restrict__ xl = p.x + (int64_t)n * p.cin * hw + px_c;
    const float* __restrict__ wr[HS_RB];
#pragma unroll
    for (int r = 0; r < HS_RB; ++r) wr[r] = p.w + (int64_t)n * p.w_batch_stride + (int64_t)min(row0 + r, p.cout - 1) * p.cin;
    float acc[HS_RB] = {0.f, 0.f, 0.f, 0.f};
    int c = c0;
    for (; c + HS_UN <= c1; c += HS_UN) {
        float xv[HS_UN];
#pragma unroll
        for (int e = 0; e < HS_UN; ++e) xv[e] = xl[(int64_t)(c + e) * hw];
#pragma unroll
        for (int r = 0; r < HS_RB; ++r)
#pragma unroll
            for (int e = 0; e < HS_UN; ++e) acc[r] = fmaf(xv[e], wr[r][c + e], acc[r]);
    }
    for (; c < c1; ++c) {
        const float xv = xl[(int64_t)c * hw];
#pragma unroll
        for (int r = 0; r < HS_RB; ++r) acc[r] = fmaf(xv, wr[r][c], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < HS_RB; ++r) s_red[ks][r][px_l] = acc[r];
    __syncthreads();
    // thread (row, pixel) of the first HS_RB x PX: slices added in slice order (deterministic)
    float amax = 0.f;
    if (threadIdx.x < HS_RB * PX) {
        const int r = threadIdx.x / PX, row = row0 + r, px = (int)blockIdx.y * PX + px_l;
        float v = s_red[0][r][px_l];
#pragma unroll
        for (int q = 1; q < KS; ++q) v += s_red[q][r][px_l];
        if (row < p.cout && px < hw) {
            if (p.bias) v += p.bias[row];
            v *= p.gain;
            if (p.clamp >= 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
            p.y[((int64_t)n * p.cout + row) * hw + px] = v;
            amax_acc(amax, v);
        }
    }
    if (p.y_amax != nullptr) amax_raise_block(p.y_amax, n, amax, s_am);
}


// ------------------------------------------------------------------------------------------------
// Last output row / column of the transposed 3x3 convolution (round 4)
// ------------------------------------------------------------------------------------------------
// y = conv_transpose2d(x, w, stride 2) is (2h + 1) x (2w + 1): oy = 2 iy + ky.  The all-class kernels tile the (h + 1) x (w + 1) grid of
// positions g (outputs 2g + class), and the "+ 1" row / column is pure tile quantisation at the benchmark's power-of-two maps: 129 x 129
// positions in 8 x 16 tiles = 612 workgroups = 2.4 rounds of one workgroup per CU — three rounds — where 128 x 128 = 512 = two (same for
// 257^2 in 16 x 16: 1156 vs 1024, 65^2 in 4 x 16: 680 vs 512, 33^2: 864 vs 512).  That extra position row / column only produces output
// row 2h (from input row h - 1, ky = 2) and column 2w (input column w - 1, kx = 2): a 1-D transposed convolution, 0.3 % of the layer's
// work.  In `strip` plans the main kernel runs on the h x w grid and this kernel fills row 2h and column 2w with plain fp32 FMAs (exact
// products, one thread per output position x 8 output channels; styles / demodulation like the main kernel; `y_amax` raised).
constexpr int STRIP_CO = 8, STRIP_WAVES = 16;     // 16 waves: the input channels of a workgroup in 16 interleaved slices (8 waves: 244 us at 512 -> 256 in@64)
// strip weights of a layer, packed once per weight version: ws[ci][t][co], t = 0..2: w[co, ci, ky = 2, kx = t] (row), t = 3..5: w[co, ci, ky = t - 3, kx = 2]
// (column); co padded to a multiple of STRIP_CO
__global__ void __launch_bounds__(256)
tconv_strip_pack_kernel(const float* __restrict__ w, int cout, int cin, int cout_pad, float* __restrict__ ws) {
    const int64_t total = (int64_t)cin * 6 * cout_pad;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % cout_pad); const int64_t r = i / cout_pad;
        const int t = (int)(r % 6), ci = (int)(r / 6);
        const int tap = t < 3 ? 6 + t : (t - 3) * 3 + 2;
        ws[i] = co < cout ? w[((int64_t)co * cin + ci) * 9 + tap] : 0.f;
    }
}
// The strip's inputs — row h - 1 and column w - 1 of every input channel — gathered into xs[n][ci][w + h] first.  Read in place they sit at
// the same offset of every power-of-two sized channel plane, i.e. in ONE L2 channel, and every output-channel block of the strip kernel
// reads all of them again: 79 us per launch at 512 -> 256 in@64 against 25 us with the loads removed.  Gathered, consecutive channels are
// (w + h) * 4 bytes apart and spread over the channels; the gather itself touches each line once.
__global__ void __launch_bounds__(256)
tconv_strip_gather_kernel(const float* __restrict__ x, int rows, int h, int w_, float* __restrict__ xs) {
    const int len = w_ + h;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)rows * len; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / len; const int e = (int)(i - r * len);
        xs[i] = e < w_ ? x[r * h * w_ + (int64_t)(h - 1) * w_ + e] : x[r * h * w_ + (int64_t)(e - w_) * w_ + (w_ - 1)];
    }
}
// One workgroup = 64 strip positions of ONE kind — row / column x even / odd position, so that which taps contribute is uniform: even
// positions t = 2a take x[a] * w(tap 0) + x[a - 1] * w(tap 2), odd ones t = 2a + 1 take x[a] * w(tap 1) — x STRIP_CO output channels of one
// image; wave q sums the input channels ci = q, q + 8, ... (lane = position), the eight partial sums meet in LDS.
__global__ void __launch_bounds__(64 * STRIP_WAVES, 4)      // <= 128 registers
tconv_strip_kernel(ide3d_modconv_params p, const float* __restrict__ ws, const float* __restrict__ xs, int cout_pad, int oh, int ow, int64_t pitch,
                   int nb_re, int nb_ro, int nb_ce) {
    __shared__ float s_red[STRIP_WAVES][STRIP_CO][64];
    __shared__ float s_am[STRIP_WAVES];
    const int h = p.h, w_ = p.w_;
    const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // block kind (uniform): 0 row even, 1 row odd, 2 column even, 3 column odd
    int b = (int)blockIdx.x, kind = 0;
    if (b >= nb_re) { b -= nb_re; kind = 1; if (b >= nb_ro) { b -= nb_ro; kind = 2; if (b >= nb_ce) { b -= nb_ce; kind = 3; } } }
    const bool is_row = kind < 2, odd = kind & 1;
    const int a = b * 64 + lane;
    const int t = 2 * a + (odd ? 1 : 0);                          // ox on the row / oy on the column
    const bool live = t < (is_row ? ow : oh - 1);
    const int co0 = (int)blockIdx.y * STRIP_CO, n = (int)blockIdx.z;
    const int len = is_row ? w_ : h;
    const bool ok_a = live && a < len, ok_b = live && !odd && a >= 1;
    const int64_t xs_pitch = (int64_t)w_ + h;                    // gathered strip inputs: [ci][row h - 1 (w floats) | column w - 1 (h floats)]
    const float* __restrict__ xn = xs + (int64_t)n * p.cin * xs_pitch;
    const int64_t off_a = (is_row ? 0 : w_) + min(a, len - 1), off_b = (is_row ? 0 : w_) + min(max(a - 1, 0), len - 1);     // valid for every lane: unconditional loads
    const float* __restrict__ wt = ws + (is_row ? 0 : 3) * (int64_t)cout_pad + co0;      // + (ci * 6 + tap) * cout_pad
    // The (ci, tap, co) weights are wave-uniform, and hipcc turns uniform global loads of this pattern into per-lane dwordx4 loads (30 per
    // unrolled iteration: 133 us per launch in the first version).  Each wave therefore fetches the 8 x 24 weights of ITS input channels of a
    // chunk with three coalesced loads per lane, parks them in a wave-private LDS slice and reads them back as broadcasts: no workgroup
    // barrier inside the K loop, and every load of a chunk — samples, styles, weights — is in flight together.
    constexpr int STRIP_CH = 8 * STRIP_WAVES, PER_WAVE = STRIP_CH / STRIP_WAVES, WPC = 3 * STRIP_CO;      // 8 input channels per wave and chunk, 24 weights per input channel
    static_assert(PER_WAVE * WPC == 3 * 64, "three weight loads per lane and chunk");
    __shared__ __attribute__((aligned(16))) float s_wt[STRIP_WAVES][PER_WAVE * WPC];
    float acc[STRIP_CO];
#pragma unroll
    for (int j = 0; j < STRIP_CO; ++j) acc[j] = 0.f;
    // (a software pipeline over the chunks - the loads of chunk c + 1 in flight across the multiplications of chunk c - measured no better:
    // 245 / 162 vs 242 / 151 us at 512 -> 256 in@64 / 512 -> 512 in@32)
    for (int c0 = 0; c0 < p.cin; c0 += STRIP_CH) {
        float xa[PER_WAVE], xb[PER_WAVE], scv[PER_WAVE], wreg[3];
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ci = min(c0 + q + k * STRIP_WAVES, p.cin - 1);
            xa[k] = xn[(int64_t)ci * xs_pitch + off_a];
            xb[k] = odd ? 0.f : xn[(int64_t)ci * xs_pitch + off_b];
            scv[k] = p.styles ? p.styles[(int64_t)n * p.cin + ci] : 1.f;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = lane + 64 * r, k = e / WPC, rem = e - k * WPC, tp = rem / STRIP_CO, jj = rem - tp * STRIP_CO;
            const int ci = c0 + q + k * STRIP_WAVES;
            wreg[r] = (ci < p.cin) ? wt[((int64_t)ci * 6 + tp) * cout_pad + jj] : 0.f;          // zero beyond cin
        }
        __builtin_amdgcn_wave_barrier();                                      // (the previous chunk's reads of this slice are done: same wave, in order)
#pragma unroll
        for (int r = 0; r < 3; ++r) s_wt[q][lane + 64 * r] = wreg[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const float va = ok_a ? xa[k] * scv[k] : 0.f;
            const float4* wv = reinterpret_cast<const float4*>(&s_wt[q][k * WPC]);       // broadcast reads: every lane the same address
            if (odd) {
                const float4 b0 = wv[2], b1 = wv[3];
                const float w1[STRIP_CO] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int j = 0; j < STRIP_CO; ++j) acc[j] = fmaf(va, w1[j], acc[j]);
            } else {
                const float vb = ok_b ? xb[k] * scv[k] : 0.f;
                const float4 a0 = wv[0], a1 = wv[1], c0v = wv[4], c1v = wv[5];
                const float w0[STRIP_CO] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                const float w2[STRIP_CO] = {c0v.x, c0v.y, c0v.z, c0v.w, c1v.x, c1v.y, c1v.z, c1v.w};
#pragma unroll
                for (int j = 0; j < STRIP_CO; ++j) { acc[j] = fmaf(va, w0[j], acc[j]); acc[j] = fmaf(vb, w2[j], acc[j]); }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < STRIP_CO; ++j) s_red[q][j][lane] = acc[j];
    __syncthreads();
    float am = 0.f;
    {
        static_assert(STRIP_CO <= STRIP_WAVES, "wave q < STRIP_CO finishes output channel co0 + q");
        float v = 0.f;
        const int qc = min(q, STRIP_CO - 1);
#pragma unroll
        for (int k = 0; k < STRIP_WAVES; ++k) v += s_red[k][qc][lane];
        const int co = co0 + q;
        if (live && q < STRIP_CO && co < p.cout) {
            v *= p.dcoefs ? p.dcoefs[(int64_t)n * p.cout + co] : 1.f;
            const int oy = is_row ? oh - 1 : t, ox = is_row ? t : ow - 1;
            p.y[((int64_t)n * p.cout + co) * ((int64_t)oh * pitch) + (int64_t)oy * pitch + ox] = v;
            amax_acc(am, v);
        }
    }
    if (p.y_amax) amax_raise_block(p.y_amax, n, am, s_am);
}

// ((split, img_group, tile), m-block) with m-block fastest (blocks that share an input patch are neighbours)
struct BlockId { int mb, tile, grp, split; };
__device__ __forceinline__ BlockId decode_block(const ConvGeom& g) {
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    BlockId b;
    b.mb = bid % g.mblocks; bid /= g.mblocks;
    b.tile = bid % g.tile_base[4]; bid /= g.tile_base[4];
    b.grp = bid % g.img_groups; bid /= g.img_groups;
    b.split = bid;
    return b;
}

template <int MODE, int BIG, int TI, int PH, int PW, int NWV = 4, int KCO = 0>
// 4-wave workgroups: 128-pixel (and smaller) tiles need <= 168 VGPRs and <= 53 KB of LDS, three workgroups per CU (measured +2 %
// frames/s over two); the stride-2 mode's (2P + 1) x (2Q + 1) patches need more LDS than that; 16 accumulators per wave take
// AGPRs and one workgroup per CU.  8-wave workgroups: one per CU, two waves per SIMD.
__global__ void __launch_bounds__(64 * NWV, NWV == 8 ? 2
                                  : (McCfg<MODE, BIG, TI, PH, PW, NWV, KCO>::NCLS * McCfg<MODE, BIG, TI, PH, PW, NWV, KCO>::MTW * McCfg<MODE, BIG, TI, PH, PW, NWV, KCO>::NTW > 8) ? 1
                                  : (TI * PH * PW <= 128 && MODE != MODE_CONV3S2 && KCO == 0) ? 3 : 2)
modconv_kernel(ide3d_modconv_params p, const float* __restrict__ wp, float* __restrict__ partial, ConvGeom g) {
    using K = McCfg<MODE, BIG, TI, PH, PW, NWV, KCO>;
    __shared__ __attribute__((aligned(16))) float s_w[2 * K::LDS_W];
    __shared__ __attribute__((aligned(16))) float s_x[2 * K::LDS_X];
    const BlockId b = decode_block(g);
    int cls = 0;
    if (MODE == MODE_TCONV3) { cls = (b.tile >= g.tile_base[1]) + (b.tile >= g.tile_base[2]) + (b.tile >= g.tile_base[3]); }
    modconv_tile<MODE, BIG, TI, PH, PW, NWV, KCO>(p, wp, partial, g, s_w, s_x, b.mb, b.tile - g.tile_base[cls], b.grp, b.split, cls, g.tiles_x[cls]);
}


// EXCLUSIVE RESIDENCY (round 4).  On MI355X a packed-fp32 instruction (v_pk_fma_f32 ...) of ANOTHER kernel's wave that
// consumes registers a global load has just written returns wrong values while a wave on the SAME SIMD runs a loop of LDS reads + bf16 / fp16
// MFMAs (scripts/micro/pk_mfma_hazard2.cpp: victims on another SIMD of the same CU are never hit, 0 of 3000 launches against 2980 of 3000).
// The foreign kernel cannot be fixed from here, so every kernel of this library that contains such a loop makes sure no foreign wave can
// share its SIMDs while the loop runs: an 8-wave workgroup puts two of its own waves, 256 registers each, on every SIMD (all waves are
// allocated together and none leaves before the last barrier of the K loop); a 4-wave workgroup's wave claims all 512 registers of its SIMD.
// `make EXTRA=-DIDE3D_SP_SHARED_SIMD` drops the claims (A/B experiments only: two 4-wave workgroups per CU again).
#ifdef IDE3D_SP_SHARED_SIMD
constexpr bool kSpExclusive = false;
#else
constexpr bool kSpExclusive = true;
#endif
template <int MODE, int BIG, int PH, int PARTS, int WBUF, int NWV>
constexpr int sp_waves_per_simd() {
    using K = SpCfg<MODE, BIG, PH, PARTS, WBUF, NWV>;
    return (NWV == 8) ? 2 : (!kSpExclusive && K::LDS_BYTES <= 80 * 1024 && (WBUF == 1 || K::NCLS * K::MTW * K::NTW <= 8)) ? 2 : 1;
}
template <int MODE, int BIG, int PH, int PARTS, int WBUF = 2, int NWV = 4, int F16 = 0>
__global__ void __launch_bounds__(64 * NWV, (sp_waves_per_simd<MODE, BIG, PH, PARTS, WBUF, NWV>()))
modconv_split_kernel(ide3d_modconv_params p, const u32x4* __restrict__ wp, float* __restrict__ partial, ConvGeom g, const float* __restrict__ row_unscale) {
    using K = SpCfg<MODE, BIG, PH, PARTS, WBUF, NWV>;
    if constexpr (kSpExclusive) { if constexpr (NWV == 8) asm volatile("" ::: "v255"); else asm volatile("" ::: "v255", "a255"); }
    __shared__ __attribute__((aligned(16))) unsigned char sp_smem[K::LDS_BYTES];
    const BlockId b = decode_block(g);
    modconv_split_tile<MODE, BIG, PH, PARTS, WBUF, NWV, F16>(p, wp, partial, g, sp_smem, b.mb, b.tile, b.grp, b.split, g.tiles_x[0], row_unscale);
}

// Row-parity pairs of the all-class transposed 3x3 convolution (round 5).  out[2 i + ky] += x[i] w[ky]: the even output rows take kernel rows 0 and 2,
// the odd ones kernel row 1, so a workgroup that owns ONE row parity holds two accumulator sets (the column classes) instead of four and takes
// twice the positions into the same registers: 16 x 16 positions x 128 rows (32 x 16 x 64 rows), 64 x 64 outputs per wave and class — a stage
// (one kernel row = one 36 KB weight slab, one barrier) carries 72 MFMAs per wave like the stride-1 form, not 36.  The even-row workgroups run
// two stages per chunk, the odd-row ones one: the first half of the grid is the even rows (dispatched first), the odd rows fill in behind
// them, so the form pays where the launch is several rounds of one workgroup per CU.  Strip plans only (h x w grid; the last output row and
// column come from tconv_strip_kernel).  Every accumulator sees the same products in the same order as in the all-class form: bit-equal results.
// Measured (eager launch incl. the strip kernels, bf16x6, batch 4): 256 -> 128 in@128 243-251 -> 215-228 us; f16x3 152 -> 152.  The 64-row form
// (32 x 16 positions) is SLOWER than the all-class 16 x 16 form (128 -> 64 in@256: 253 -> 291 us) and is only reachable with IDE3D_MODCONV_PAIR=2;
// two launches (even rows, then odd rows) instead of one: no difference (the scalar spills of the two-path kernel do not matter).
template <int BIG, int PH, int PARTS, int F16 = 0>
__global__ void __launch_bounds__(512, 2)
modconv_split_pair_kernel(ide3d_modconv_params p, const u32x4* __restrict__ wp, float* __restrict__ partial, ConvGeom g, const float* __restrict__ row_unscale) {
    using K = SpCfg<MODE_TCONV3A, BIG, PH, PARTS, 2, 8, 1>;
    if constexpr (kSpExclusive) asm volatile("" ::: "v255");
    __shared__ __attribute__((aligned(16))) unsigned char sp_smem[K::LDS_BYTES];
    const int half_grid = (int)(gridDim.x >> 1);
    const int par = __builtin_amdgcn_readfirstlane((int)blockIdx.x >= half_grid ? 1 : 0);
    int bid = xcd_remap((int)blockIdx.x - par * half_grid, half_grid);
    BlockId b;
    b.mb = bid % g.mblocks; bid /= g.mblocks;
    b.tile = bid % g.tile_base[4]; bid /= g.tile_base[4];
    b.grp = bid % g.img_groups; bid /= g.img_groups;
    b.split = bid;
    if (par == 0) modconv_split_tile<MODE_TCONV3A, BIG, PH, PARTS, 2, 8, F16, 1, 0>(p, wp, partial, g, sp_smem, b.mb, b.tile, b.grp, b.split, g.tiles_x[0], row_unscale);
    else          modconv_split_tile<MODE_TCONV3A, BIG, PH, PARTS, 2, 8, F16, 1, 1>(p, wp, partial, g, sp_smem, b.mb, b.tile, b.grp, b.split, g.tiles_x[0], row_unscale);
}

// Two 4-wave teams per workgroup (exclusive residency without giving up the second wave per SIMD for the forms whose tile is too small
// for eight waves: the 4 x 16-position transposed tiles of the 4^2 .. 32^2 layers).  Team t takes output-channel block 2 k + t of the same
// (tile, image, K split): same geometry, same number of stages, so the workgroup barriers inside the tile code line up; each team has its
// own LDS and its own thread numbering.  A team alone on a CU has nothing to overlap its staging with (round 4: 247 vs 183 us at 512 ->
// 512 in@32 when two workgroups shared a CU); two teams in lock-step overlap less than two free-running workgroups did, but more than none.
template <int MODE, int BIG, int PH, int PARTS, int WBUF, int F16>
__global__ void __launch_bounds__(512, 2)
modconv_split_teams_kernel(ide3d_modconv_params p, const u32x4* __restrict__ wp, float* __restrict__ partial, ConvGeom g, const float* __restrict__ row_unscale) {
    using K = SpCfg<MODE, BIG, PH, PARTS, WBUF, 4>;
    static_assert(2 * K::LDS_BYTES <= 160 * 1024, "both teams' LDS must fit");
    asm volatile("" ::: "v255");                       // 8 waves x 256 registers: the workgroup owns its CU's register files
    __shared__ __attribute__((aligned(16))) unsigned char sp_smem[2 * K::LDS_BYTES];
    const int team = (int)(threadIdx.x >> 8);
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mpairs = g.mblocks >> 1;
    BlockId b;
    b.mb = 2 * (bid % mpairs) + team; bid /= mpairs;
    b.tile = bid % g.tile_base[4]; bid /= g.tile_base[4];
    b.grp = bid % g.img_groups; bid /= g.img_groups;
    b.split = bid;
    modconv_split_tile<MODE, BIG, PH, PARTS, WBUF, 4, F16, 2>(p, wp, partial, g, sp_smem + team * K::LDS_BYTES, b.mb, b.tile, b.grp, b.split, g.tiles_x[0], row_unscale,
                                                              reinterpret_cast<float*>(sp_smem));
}

// reduce split-K partials + epilogue
__global__ void __launch_bounds__(256)
modconv_epilogue_kernel(ide3d_modconv_params p, const float* __restrict__ partial, int split_k, int oh, int ow) {
    const int64_t per = (int64_t)p.n * p.cout * oh * ow;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int am_n = -1; float am = 0.f;
    __shared__ unsigned s_am[64];
    const bool am_lds = p.y_amax && p.n <= 64;                  // per-workgroup maxima in LDS, one global update per image and workgroup
    if (p.y_amax) { if (threadIdx.x < 64) s_am[threadIdx.x] = 0u; __syncthreads(); }
    // dense rows of a multiple of four pixels per plane: four consecutive outputs (same image, same channel) per thread, 16-byte loads
    const int hw = oh * ow;
    if (p.y_pitch <= 0 && (hw & 3) == 0) {
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        const int64_t per4 = per >> 2;
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < per4; q += stride) {
            const int64_t i = q << 2;
            f32x4v v = {0.f, 0.f, 0.f, 0.f};
            int s = 0;
            for (; s + 4 <= split_k; s += 4) {
                f32x4v t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = *reinterpret_cast<const f32x4v*>(partial + (int64_t)(s + k) * per + i);
#pragma unroll
                for (int k = 0; k < 4; ++k) v += t[k];
            }
            for (; s < split_k; ++s) v += *reinterpret_cast<const f32x4v*>(partial + (int64_t)s * per + i);
            const int pix = (int)(i % hw);
            const int co = (int)((i / hw) % p.cout);
            const int n = (int)(i / ((int64_t)hw * p.cout));
            const float d = p.dcoefs ? p.dcoefs[(int64_t)n * p.cout + co] : 1.f, bb = p.bias ? p.bias[co] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = v[e];
                if (p.dcoefs) x *= d;
                if (p.noise) x += p.noise[pix + e] * p.noise_strength;
                if (p.bias) x += bb;
                if (p.act == 3) x = (x > 0.f) ? x : x * p.alpha;
                x *= p.gain;
                if (p.clamp >= 0.f) x = fminf(fmaxf(x, -p.clamp), p.clamp);
                v[e] = x;
            }
            *reinterpret_cast<f32x4v*>(p.y + i) = v;
            if (p.y_amax) {
                if (n != am_n) { if (am_n >= 0) { if (am_lds) amax_lds_flush(s_am, am_n, am); else amax_commit(p.y_amax, am_n, am, false); } am_n = n; am = 0.f; }
                amax_acc(am, v[0]); amax_acc(am, v[1]); amax_acc(am, v[2]); amax_acc(am, v[3]);
            }
        }
    } else
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += stride) {
        // partial sums added in split order (deterministic); the loads of up to eight splits are in flight together — one by one
        // every addition waits a whole L2 round trip and the twelve reductions of a pass cost 0.13 ms
        float v = 0.f;
        int s = 0;
        for (; s + 8 <= split_k; s += 8) {
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = partial[(int64_t)(s + k) * per + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) v += t[k];
        }
        if (s + 4 <= split_k) {
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = partial[(int64_t)(s + k) * per + i];
#pragma unroll
            for (int k = 0; k < 4; ++k) v += t[k];
            s += 4;
        }
        for (; s < split_k; ++s) v += partial[(int64_t)s * per + i];
        const int pix = (int)(i % ((int64_t)oh * ow));
        const int co = (int)((i / ((int64_t)oh * ow)) % p.cout);
        const int n = (int)(i / ((int64_t)oh * ow * p.cout));
        if (p.dcoefs) v *= p.dcoefs[(int64_t)n * p.cout + co];
        if (p.noise) v += p.noise[pix] * p.noise_strength;
        if (p.bias) v += p.bias[co];
        if (p.act == 3) v = (v > 0.f) ? v : v * p.alpha;
        v *= p.gain;
        if (p.clamp >= 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
        if (p.y_pitch > 0) p.y[(i / ow) * p.y_pitch + (i % ow)] = v; else p.y[i] = v;        // rows of y may be padded (y_pitch)
        if (p.y_amax) {                                          // small layers only (split-K): per-thread running maximum per image
            if (n != am_n) { if (am_n >= 0) { if (am_lds) amax_lds_flush(s_am, am_n, am); else amax_commit(p.y_amax, am_n, am, false); } am_n = n; am = 0.f; }
            amax_acc(am, v);
        }
    }
    if (p.y_amax) {
        if (am_n >= 0) { if (am_lds) amax_lds_flush(s_am, am_n, am); else amax_commit(p.y_amax, am_n, am, false); }
        if (am_lds) amax_lds_commit(p.y_amax, s_am, p.n);
    }
}


// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct ConvPlan {
    int mode, big, tile;             // tile: 0 = 1x8x16, 1 = 2x8x8, 2 = 8x4x4
    int bm, kc, taps, mblocks, cchunks, oh, ow;
    int parts;                       // 0: fp32 MFMA loop; 2 / 3: split loop with that many pieces per operand
    int f16;                         // split loop on fp16 pieces (f16x3: parts == 2) instead of bf16
    int pair;                        // all-class transposed conv, split loop: one output-row parity per workgroup (modconv_split_pair_kernel), grid doubled
    int strip;                       // all-class transposed conv on the h x w class grid; output row 2h and column 2w by tconv_strip_kernel
    int64_t strip_off, strip_floats; // packed strip weights [cin][6][cout_pad] inside the aux region (every all-class plan reserves them)
    int64_t packed_floats, aux_floats, partial_floats;      // aux: per-row scale + unscale of the f16x3 weights
    ConvGeom g;
};

// Developer knobs (read once): IDE3D_MODCONV_NO_FLAT / _NO_TCONV3A switch the flattened 1x1 tiles / the all-class
// transposed kernel off, IDE3D_MODCONV_TILE forces a pixel tile (0..3), IDE3D_MODCONV_DEBUG = 1 drops the staging after
// the first chunk (timing experiments; wrong results).
struct McEnv { bool no_flat, no_allcls; int tile, debug, ta_rows; };
static const McEnv& mc_env() {          // (knobs.h: read once)
    static const McEnv e = {knobs().mc_no_flat, knobs().mc_no_allcls, knobs().mc_tile, knobs().mc_debug, knobs().mc_ta_rows};
    return e;
}

// Experiment switches of plan_conv (A/B runs of scripts/micro/: each is the "before" of a plan rule and is documented where the rule is),
// read once.  IDE3D_MODCONV_NO_STRIP alone is read per call: tests/test_gpu_conv_arith.py flips it inside one process.
struct PlanKnobs { bool ta_bm64, head_bm128, ta_kc8, no_smallmap, ta_old, sp_oldplan, no_ph32, no_w8split, no_one_round; int sp_modes; };
static const PlanKnobs& plan_knobs() {          // (knobs.h: read once)
    const Knobs& e = knobs();
    static const PlanKnobs k = {e.ta_bm64, e.head_bm128, e.ta_kc8, e.no_smallmap, e.ta_old, e.sp_oldplan, e.no_ph32, e.no_w8split, e.no_one_round, e.sp_modes};
    return k;
}

// A 1x1 convolution does not see the image shape: h x w is treated as one row, tiled in runs of 128 pixels whose patch
// rows are contiguous in memory (16-byte staging, 128-byte output runs).  Needs 16-byte aligned rows.
static ide3d_modconv_params flatten_pointwise(const ide3d_modconv_params& in) {
    ide3d_modconv_params p = in;
    const int64_t hw = (int64_t)p.h * p.w_;
    if (p.k == 1 && p.mode == 0 && hw % 4 == 0 && hw >= 128 && ((uintptr_t)p.x % 16) == 0 && !mc_env().no_flat) { p.h = 1; p.w_ = (int)hw; }
    return p;
}

// Arithmetic of the big 3x3 layers: 1 = fp32 MFMA (exact fp32 products), 6 = three bf16 pieces per operand / 6 products
// (fp32-grade: per-product error <= 2^-23 relative, measured as accurate as the fp32 MFMA against float64), 3 = two pieces / 3 products
// (~2^-17 relative per product), 16 = two fp16 pieces / 3 products with exact power-of-two range scales (~2^-21; needs x_amax).
// Process default: IDE3D_CONV_ARITH, else bf16x6 (round 4).  Rounds 2-3 kept fp32 as the default because a foreign kernel's packed-fp32
// wave beside an LDS-fed bf16 / fp16 MFMA loop returns wrong results on MI355X; every such loop of this library now keeps foreign waves
// off its SIMDs (exclusive residency, `kSpExclusive` above / DESIGN.md section 4.2), so the fp32-grade split arithmetic is the default.
static int g_conv_arith = 0;
static int conv_arith_default() {
    if (g_conv_arith) return g_conv_arith;
    static const int env = [] {
        return knobs().conv_arith;          // IDE3D_CONV_ARITH (knobs.h)
    }();
    return env;
}
static int resolve_arith(int a) { return (a == 1 || a == 3 || a == 6 || a == 16) ? a : conv_arith_default(); }

// tile index (ConvPlan::tile) -> images per tile, tile height, tile width (pixels; transposed all-class form: grid positions)
static const int TIv[13] = {1, 2, 8, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, PHv[13] = {8, 8, 4, 16, 4, 1, 8, 16, 8, 16, 1, 8, 32}, PWv[13] = {16, 8, 4, 16, 16, 128, 16, 16, 16, 16, 256, 16, 16};

static void plan_conv(const ide3d_modconv_params& p, ConvPlan& pl, int arith) {
    pl.mode = (p.mode == 2) ? MODE_TCONV3 : (p.mode == 1) ? MODE_CONV3S2 : (p.k == 1 ? MODE_CONV1 : MODE_CONV3);
    // all-class form from 4 x 4 maps on (round 3: the per-class form took 65 / 78 us for the 0.3-GFLOP layers at 4^2 / 8^2; 962 -> 974-981
    // frames/s; IDE3D_MODCONV_ALLCLS_MIN=12 restores the old threshold)
    const int allcls_min = knobs().allcls_min;
    const bool allcls = (pl.mode == MODE_TCONV3) && mc_bm(p.cout) >= 64 && p.h >= allcls_min && p.w_ >= allcls_min && !mc_env().no_allcls;
    if (allcls) pl.mode = MODE_TCONV3A;
    pl.bm = mc_bm(p.cout);
    if (allcls && pl.bm == 128) {
        // all-class transposed conv with too few 128-row blocks to fill the 768 resident slots (3 per CU): 64-row blocks double
        // the block count (512 -> 512 in@32: 432 -> 864 blocks, measured +5 %)
        const int64_t blocks128 = (int64_t)cdiv(p.cout, 128) * cdiv(p.h + 1, 4) * cdiv(p.w_ + 1, 16) * p.n;
        const bool old_plan = knobs().ta_old;
        if ((blocks128 < 3 * kNumCU * 3 / 4 && !old_plan) || mc_env().ta_rows == 64 || plan_knobs().ta_bm64) pl.bm = 64;
    }
    // 1x1 heads with cout = 192 (96 + 96 tri-plane channels): three 64-row blocks instead of 128 + 64 rows padded to 128
    if (p.k == 1 && pl.bm == 128 && p.cout % 128 != 0 && p.cout % 64 == 0 && !plan_knobs().head_bm128) pl.bm = 64;
    // experiment (IDE3D_MODCONV_TA_KC8): all-class transposed conv as 64-row blocks x 8 x 16 positions x 8 input channels per chunk:
    // 72 MFMAs per wave and barrier instead of 36, 18 KB of weights streamed per 72 MFMAs instead of per 36
    bool ta_kc8 = false;
    if (allcls && p.cin % 8 == 0 && plan_knobs().ta_kc8) { pl.bm = 64; ta_kc8 = true; }
    pl.big = pl.bm == 128 ? 1 : (pl.bm == 64 ? 2 : 0);
    pl.kc = ta_kc8 ? 8 : mc_kc(p.k); pl.taps = p.k * p.k;
    pl.mblocks = cdiv(p.cout, pl.bm); pl.cchunks = cdiv(p.cin, pl.kc);
    const bool transposed = (pl.mode == MODE_TCONV3 || pl.mode == MODE_TCONV3A);
    pl.oh = transposed ? 2 * p.h + 1 : (pl.mode == MODE_CONV3S2) ? (p.h - 3) / 2 + 1 : p.h;
    pl.ow = transposed ? 2 * p.w_ + 1 : (pl.mode == MODE_CONV3S2) ? (p.w_ - 3) / 2 + 1 : p.w_;
    pl.packed_floats = (int64_t)pl.mblocks * pl.cchunks * pl.taps * pl.kc * pl.bm * (p.w_batch_stride ? p.n : 1);
    pl.parts = 0; pl.f16 = 0; pl.aux_floats = 0; pl.strip_off = 0; pl.strip_floats = 0; pl.pair = 0;
    int want_split = 0;                                         // split-K chosen together with the form (0 = by block count below)
    bool one_round = false;                                     // transposed 8-wave form chosen for whole rounds on the strip plan's grid: no split-K
    const int split_min = knobs().split_min;      // fewer workgroups than this: split-K
    // class grids
    int gh[4], gw[4];
    const int ncls = (pl.mode == MODE_TCONV3) ? 4 : 1;
    for (int c = 0; c < 4; ++c) {
        if (pl.mode == MODE_TCONV3) { gh[c] = (c >> 1) ? p.h : p.h + 1; gw[c] = (c & 1) ? p.w_ : p.w_ + 1; }
        else if (pl.mode == MODE_TCONV3A) { gh[c] = p.h + 1; gw[c] = p.w_ + 1; }
        else if (pl.mode == MODE_CONV3S2) { gh[c] = pl.oh; gw[c] = pl.ow; }
        else { gh[c] = p.h; gw[c] = p.w_; }
    }
    const int mind = (pl.mode == MODE_CONV3S2) ? ((pl.oh < pl.ow) ? pl.oh : pl.ow) : ((p.h < p.w_) ? p.h : p.w_);
    pl.tile = (mind >= 12 || p.w_batch_stride) ? 0 : (mind >= 6 ? 1 : 2);     // per-image weights need one image per tile
    // small maps (4^2 .. 8^2) of wide 3x3 layers in a split arithmetic: the 8 x 16-pixel split-bf16 tile (one image per tile, most of it
    // outside the map) with split-K 16 instead of the fp32 loop's several-images tiles: 35.5 -> 23.1 us at 512 -> 512 @8, 33.4 -> 20.2 @4
    // (the launch is weight streaming + latency: 256 workgroups of two 16-channel chunks each)
    if (arith != 1 && pl.mode == MODE_CONV3 && pl.tile != 0 && !p.w_batch_stride && pl.big != 0 && p.cin >= 256 && mc_env().tile < 0 &&
        !plan_knobs().no_smallmap) {
        pl.tile = 0;
        const int c16 = cdiv(p.cin, 16);
        want_split = c16 / 2 < 1 ? 1 : (c16 / 2 > 16 ? 16 : c16 / 2);
    }
    // 256-pixel tiles (8 accumulators per wave) for big-cout 3x3 layers with enough work to fill the chip twice over
    if (pl.tile == 0 && pl.big == 1 && pl.mode != MODE_CONV1 && pl.mode != MODE_CONV3S2 && !p.w_batch_stride) {
        const int64_t blocks256 = (int64_t)pl.mblocks * cdiv(gh[0], 16) * cdiv(gw[0], 16) * p.n * ((pl.mode == MODE_TCONV3) ? 4 : 1);
        if (blocks256 >= 2 * kNumCU) pl.tile = 3;
    }
    if (mc_env().tile >= 0) { const int t = mc_env().tile; if (t >= 0 && t <= 3 && (t == 0 || t == 3 || !p.w_batch_stride)) pl.tile = (t == 3 && (pl.big != 1 || pl.mode == MODE_CONV1 || pl.mode == MODE_CONV3S2)) ? 0 : t; }
    if (pl.mode == MODE_TCONV3A) {
        // grid positions per block x 4 classes: 4 x 16 (8 accumulators per wave at BM = 128, two or three workgroups per CU) or —
        // the weight slab of a K chunk (18 KB at BM = 128) is streamed L2 -> LDS once per chunk and per block, and that stream
        // (~11 B/clk/CU) is what bounds the small tile — 8 x 16 / 16 x 16 positions with 16 accumulators per wave (AGPRs, one
        // workgroup per CU): twice the MFMAs per streamed weight byte
        pl.tile = 4;
        // 64-row blocks have registers for 8 x 16 positions (8 accumulators per wave): half the weight bytes streamed per MFMA
        // (128 -> 64 in@256: measured +3.5 %) as long as enough blocks remain
        if (pl.big == 2 && (int64_t)pl.mblocks * cdiv(p.h + 1, 8) * cdiv(p.w_ + 1, 16) * p.n >= 4 * kNumCU && !plan_knobs().ta_old) pl.tile = 6;
        const int rows = mc_env().ta_rows ? mc_env().ta_rows : 0;
        if (rows == 4) pl.tile = 4;
        if (ta_kc8) pl.tile = 11;
        if (rows == 8 && pl.big == 1) pl.tile = 6;
        if (rows == 16 && pl.big == 2) pl.tile = 7;
        if (rows == 8 && pl.big == 2) pl.tile = 6;
        // 512-thread workgroups: 8 x 16 positions at BM = 128, 16 x 16 at BM = 64 (8 accumulators per wave either way)
        if (rows == 108) pl.tile = (pl.big == 1) ? 8 : 9;
    }
    if (pl.mode == MODE_CONV1 && p.h == 1 && p.w_ % 4 == 0 && p.w_ >= 128) pl.tile = 5;   // flattened by flatten_pointwise()
    // split-bf16 loop: shared-weight 3x3 / all-class transposed layers on 16-pixel-wide tiles, 64- or 128-row blocks
    // (3x3 layers with fewer than 3 K chunks stay on the fp32 loop: prologue + epilogue dominate)
    // (transposed layers from 32 input channels = 2 K chunks on: 32 -> 128 in@128 78.6 -> 60.5 us on the 8-wave strip-plan form; round 3's 81 vs 76
    // the other way round was the 4-wave (h + 1) x (w + 1) form)
    const int sp_min_cin = knobs().sp_min_cin;
    const int min_cin = sp_min_cin ? sp_min_cin : (pl.mode == MODE_TCONV3A ? 32 : 33);
    if (arith != 1 && (pl.mode == MODE_CONV3 || pl.mode == MODE_TCONV3A) && !p.w_batch_stride && pl.big != 0 && p.cin >= min_cin &&
        (pl.tile == 0 || pl.tile == 3 || pl.tile == 4 || pl.tile == 6 || pl.tile == 7) && !ta_kc8 &&
        (plan_knobs().sp_modes & (pl.mode == MODE_CONV3 ? 1 : 2))) {
        pl.parts = (arith == 3 || arith == 16) ? 2 : 3;
        pl.f16 = (arith == 16) ? 1 : 0;
        const int sp_rows = knobs().sp_rows;
        // Exclusive residency (round 4): one workgroup per CU, so the forms that put TWO of their own waves on a SIMD (8 waves) are chosen
        // wherever the launch still has >= 2 workgroups per CU to run through; measured per layer, bf16x6 / f16x3 us at batch 4:
        //   64 -> 64 @512: 16 x 16 px x 64 rows, 8 waves 431 / 343 (8 x 16, 4 waves with the whole SIMD claimed: 537 / 457)
        //   transposed 256 -> 128 in@128: 8 x 16 positions x 128 rows, 8 waves 250 / 175 (4 x 16, 4 waves: 269 / 207)
        //   transposed 128 -> 64 in@256: 16 x 16 positions x 64 rows, 8 waves 270 / 211 (8 x 16, 4 waves: 318 / 263)
        //   transposed 512 -> 256 in@64 (360 workgroups of 8 x 16): stays on 4 x 16, 4 waves 292 / 215 (8 waves: 401 / 325)
        if (kSpExclusive && !plan_knobs().sp_oldplan) {
            if (pl.mode == MODE_CONV3) {
                const int64_t b256 = (int64_t)pl.mblocks * cdiv(p.h, 16) * cdiv(p.w_, 16) * p.n;
                if (pl.big == 2 && b256 >= 2 * kNumCU) pl.tile = 3;
                // 64 rows: 32 x 16 pixels (64 x 64 outputs per wave, half the tiles) while >= 2 workgroups per CU remain
                if (pl.big == 2 && (int64_t)pl.mblocks * cdiv(p.h, 32) * cdiv(p.w_, 16) * p.n >= 2 * kNumCU && !plan_knobs().no_ph32) pl.tile = 12;
                // a quarter .. one workgroup per CU on 16 x 16 pixels (512 -> 512 @32 at batch 4: 64): 8 waves and split-K up to ONE workgroup
                // per CU with >= 8 chunks each, instead of 8 x 16 pixels / 4 waves / split 6 = 768 workgroups of 2 - 6 chunks: 119 -> 93 us
                if (pl.big == 1 && b256 < 2 * kNumCU && b256 * 4 >= kNumCU && cdiv(p.cin, 16) * b256 >= 8 * kNumCU && !plan_knobs().no_w8split) {
                    pl.tile = 3; want_split = (int)(kNumCU / b256);
                }
            } else {
                const int64_t b8 = (int64_t)pl.mblocks * cdiv(p.h + 1, 8) * cdiv(p.w_ + 1, 16) * p.n, b16 = (int64_t)pl.mblocks * cdiv(p.h + 1, 16) * cdiv(p.w_ + 1, 16) * p.n;
                // on the h x w grid of a strip plan (below) the 8-wave form may come out at whole rounds of ONE workgroup per CU where the (h + 1) x
                // (w + 1) grid did not: 512 -> 256 in@64 = 256 workgroups of 8 x 16 positions x 128 rows, no split-K: 241 -> 211 us (f16x3 181 -> 149)
                // against 512 four-wave workgroups of 4 x 16; 512 -> 512 in@32 (64 rows): 149 -> 141 against 256 two-team workgroups
                const bool strip_ok = !p.w_batch_stride && !p.noise && !p.bias && p.act == 1 && p.gain == 1.f && p.clamp < 0.f && !knob_live("IDE3D_MODCONV_NO_STRIP") && !plan_knobs().no_one_round;
                const int64_t b8s = (int64_t)pl.mblocks * cdiv(p.h, 8) * cdiv(p.w_, 16) * p.n, b4s = (int64_t)pl.mblocks * cdiv(p.h, 4) * cdiv(p.w_, 16) * p.n;
                const int64_t b16s = (int64_t)pl.mblocks * cdiv(p.h, 16) * cdiv(p.w_, 16) * p.n;
                auto rounds = [](int64_t blocks) { return (blocks + kNumCU - 1) / kNumCU; };
                if (pl.big == 1) {
                    pl.tile = (b8 >= 2 * kNumCU) ? 6 : 4;
                    // (a 4-wave workgroup of 4 x 16 positions is half the work at ~1.15 x the time per unit)
                    if (pl.tile == 4 && strip_ok && b8s >= kNumCU && rounds(b8s) * 200 <= rounds(b4s) * 115) { pl.tile = 6; one_round = true; }
                }
                else if (b16 >= 2 * kNumCU) pl.tile = 7;
                // whole rounds of ONE 8-wave workgroup of 16 x 16 positions per CU on the strip grid (128 -> 64 in@256 at batch 1: 256 workgroups) against
                // four-wave workgroups of 4 x 16 (a quarter of the work at ~1.15 x the time per unit; no team pairs with an odd block count)
                else if (strip_ok && (pl.mblocks & 1) && b16s >= kNumCU && rounds(b16s) * 400 <= rounds(b4s) * 115 && !knob_live("IDE3D_MODCONV_NO_R16")) { pl.tile = 7; one_round = true; }
                else if (strip_ok && (pl.mblocks & 1) == 0 && b8s >= kNumCU && rounds(b8s) * 100 <= rounds(b4s / 2) * 105) { pl.tile = 6; one_round = true; }
                else if (p.h <= 16 && p.w_ <= 16 && (pl.mblocks & 1)) pl.tile = 6;       // 4^2 .. 16^2 maps with an odd block count (no team pairs): 8 x 16 positions, 8 waves: 99 / 45 / 24 us at in@16 / 8 / 4 (4 waves alone on a CU: 115 / 47 / 31; two 4-wave teams: 92 / 45 / 24)
            }
        }
        if (pl.mode == MODE_CONV3) {
            if (sp_rows == 8) pl.tile = 0;
            if (sp_rows == 16) pl.tile = 3;
            if (sp_rows == 32 && pl.big == 2) pl.tile = 12;
        } else {
            if (pl.big == 1 && pl.tile == 7) pl.tile = 6;            // 128-row blocks: at most 8 x 16 positions (16 accumulators)
            if (sp_rows == 4) pl.tile = 4;
            if (sp_rows == 8) pl.tile = 6;
            if (sp_rows == 16 && pl.big == 2) pl.tile = 7;
        }
        const int sp_maxlds = knobs().sp_maxlds;
        const int ph = (pl.tile == 4) ? 4 : (pl.tile == 0 || pl.tile == 6) ? 8 : (pl.tile == 12) ? 32 : 16;
        const int lds = 2 * 16 * (3 * pl.parts * 2 * pl.bm + pl.parts * 2 * (ph + 2) * 18);
        if (sp_maxlds && lds > sp_maxlds) pl.parts = 0;
        // Row-parity pairs (modconv_split_pair_kernel): twice the positions per workgroup, 72 MFMAs per stage.  The even-row workgroups carry two
        // stages per chunk, the odd-row ones one, so the form needs at least one whole round of even-row workgroups (one per CU) for the odd rows
        // to fill in behind: 256 -> 128 in@128 at batch 4; 512 -> 256 in@64 (128 per parity: 215 -> 237 us) keeps the 8 x 16 form; 128-row blocks
        // only (see the kernel).  IDE3D_MODCONV_PAIR (read per call: the tests flip it): 0 = never, 2 = wherever the form exists.
        if (pl.parts && pl.mode == MODE_TCONV3A && kSpExclusive && !sp_rows && !sp_maxlds) {
            const char* pe = knob_live_str("IDE3D_MODCONV_PAIR");
            const int pair_knob = pe ? atoi(pe) : 1;
            const bool plain = !p.w_batch_stride && !p.noise && !p.bias && p.act == 1 && p.gain == 1.f && p.clamp < 0.f && !knob_live("IDE3D_MODCONV_NO_STRIP");
            const int64_t per_parity = (int64_t)pl.mblocks * cdiv(p.h, pl.big == 1 ? 16 : 32) * cdiv(p.w_, 16) * p.n;
            if (plain && pair_knob && (pair_knob == 2 || (pl.big == 1 && per_parity >= kNumCU && p.cin >= 64))) {
                pl.pair = 1; pl.tile = (pl.big == 1) ? 7 : 12; one_round = true;
            }
        }
    }
    if (!pl.parts) pl.f16 = 0;
    if (pl.parts) {
        pl.kc = 16; pl.cchunks = cdiv(p.cin, 16);
        pl.packed_floats = sp_packed_units(pl.mblocks, pl.cchunks, pl.bm, pl.parts) * 4;
        if (pl.f16) pl.aux_floats = 2 * (int64_t)pl.mblocks * pl.bm;
    }
    if (pl.mode == MODE_TCONV3A) {                              // reserved whether or not this call's epilogue allows the strip plan (workspace sizing does not know)
        pl.strip_off = pl.aux_floats;
        pl.strip_floats = (int64_t)p.cin * 6 * (cdiv(p.cout, STRIP_CO) * STRIP_CO);
        pl.aux_floats += pl.strip_floats + (int64_t)p.n * p.cin * (p.h + p.w_);          // + the gathered strip inputs xs[n][cin][w + h]
    }
    // Strip plan (tconv_strip_kernel): the all-class transposed convolution on the h x w grid when that saves tiles, the launch needs no split-K
    // either way (the reduction kernel would finish the strip's unwritten partials) and the epilogue is the plain one of the up-sampling
    // layers (demodulation only: noise / bias / activation follow the FIR).  IDE3D_MODCONV_NO_STRIP = the (h + 1) x (w + 1) grid everywhere.
    // Shared weights only: the strip's pack / compute kernels read ONE weight tensor (per-image weights stay on the (h + 1) x (w + 1) grid).
    pl.strip = 0;
    if (pl.mode == MODE_TCONV3A && !p.w_batch_stride && !p.noise && !p.bias && p.act == 1 && p.gain == 1.f && p.clamp < 0.f && !knob_live("IDE3D_MODCONV_NO_STRIP")) {
        const int ph = PHv[pl.tile], pw = PWv[pl.tile];
        const int64_t t_full = (int64_t)cdiv(p.h + 1, ph) * cdiv(p.w_ + 1, pw), t_main = (int64_t)cdiv(p.h, ph) * cdiv(p.w_, pw);
        const int64_t groups = cdiv(p.n, TIv[pl.tile]);
        auto splits = [&](int64_t tiles) { return (int64_t)pl.mblocks * tiles * groups < split_min && pl.cchunks >= 8; };
        if (pl.pair || (t_main < t_full && (one_round || (!splits(t_main) && !splits(t_full))))) {
            pl.strip = 1;
            for (int c = 0; c < 4; ++c) { gh[c] = p.h; gw[c] = p.w_; }
        }
    }
    ConvGeom& g = pl.g;
    g.tile_base[0] = 0;
    for (int c = 0; c < 4; ++c) {
        g.tiles_x[c] = cdiv(gw[c], PWv[pl.tile]); g.tiles_y[c] = cdiv(gh[c], PHv[pl.tile]);
        g.tile_base[c + 1] = g.tile_base[c] + (c < ncls ? g.tiles_x[c] * g.tiles_y[c] : 0);
    }
    g.img_groups = cdiv(p.n, TIv[pl.tile]);
    g.mblocks = pl.mblocks; g.cchunks = pl.cchunks; g.oh = pl.oh; g.ow = pl.ow;
    g.debug = mc_env().debug;
    const int64_t base_blocks = (int64_t)g.mblocks * g.tile_base[4] * g.img_groups;
    int split = 1;
    if (base_blocks < split_min && pl.cchunks >= 8) {
        split = (int)cdiv64(768, base_blocks);
        if (split > pl.cchunks / 4) split = pl.cchunks / 4;
        if (split > 16) split = 16;
        if (split < 1) split = 1;
    }
    if (one_round) want_split = 1;
    if (want_split > 0) split = want_split < pl.cchunks ? want_split : pl.cchunks;
    const int force_split = knobs().force_split;   // experiments
    if (force_split > 0 && !pl.strip) split = force_split < pl.cchunks ? force_split : pl.cchunks;
    g.chunks_per_split = cdiv(pl.cchunks, split);
    g.split_k = cdiv(pl.cchunks, g.chunks_per_split);
    pl.partial_floats = (g.split_k > 1) ? (int64_t)g.split_k * p.n * p.cout * pl.oh * pl.ow : 0;
}

template <int MODE, int BIG>
static void launch_tiles(const ide3d_modconv_params& p, const ConvPlan& pl, const float* wp, float* partial, hipStream_t st) {
    const ConvGeom& g = pl.g;
    const unsigned nblocks = (unsigned)((int64_t)g.mblocks * g.tile_base[4] * g.img_groups * g.split_k);
    if constexpr (MODE == MODE_TCONV3A) {
        if constexpr (BIG != 0) {
            if (pl.tile == 11) { if constexpr (BIG == 2) hipLaunchKernelGGL((modconv_kernel<MODE, BIG, 1, 8, 16, 4, 8>), dim3(nblocks), dim3(256), 0, st, p, wp, partial, g); }
            else if (pl.tile == 8) { if constexpr (BIG == 1) hipLaunchKernelGGL((modconv_kernel<MODE, BIG, 1, 8, 16, 8>), dim3(nblocks), dim3(512), 0, st, p, wp, partial, g); }
            else if (pl.tile == 9) { if constexpr (BIG == 2) hipLaunchKernelGGL((modconv_kernel<MODE, BIG, 1, 16, 16, 8>), dim3(nblocks), dim3(512), 0, st, p, wp, partial, g); }
            else if (pl.tile == 6) hipLaunchKernelGGL((modconv_kernel<MODE, BIG, 1, 8, 16>), dim3(nblocks), dim3(256), 0, st, p, wp, partial, g);
            else if (pl.tile == 7) { if constexpr (BIG == 2) hipLaunchKernelGGL((modconv_kernel<MODE, BIG, 1, 16, 16>), dim3(nblocks), dim3(256), 0, st, p, wp, partial, g); }
            else hipLaunchKernelGGL((modconv_kernel<MODE, BIG, 1, 4, 16>), dim3(nblocks), dim3(256), 0, st, p, wp, partial, g);
        }
    } else
    if (pl.tile == 5) {
        if constexpr (MODE == MODE_CONV1)
            hipLaunchKernelGGL((modconv_kernel<MODE, BIG, 1, 1, 128>), dim3(nblocks), dim3(256), 0, st, p, wp, partial, g);
    }
    else if (pl.tile == 0) hipLaunchKernelGGL((modconv_kernel<MODE, BIG, 1, 8, 16>), dim3(nblocks), dim3(256), 0, st, p, wp, partial, g);
    else if (pl.tile == 1) hipLaunchKernelGGL((modconv_kernel<MODE, BIG, 2, 8, 8>), dim3(nblocks), dim3(256), 0, st, p, wp, partial, g);
    else if (pl.tile == 2) hipLaunchKernelGGL((modconv_kernel<MODE, BIG, 8, 4, 4>), dim3(nblocks), dim3(256), 0, st, p, wp, partial, g);
    else if constexpr (BIG == 1 && MODE != MODE_CONV1 && MODE != MODE_CONV3S2)
        hipLaunchKernelGGL((modconv_kernel<MODE, BIG, 1, 16, 16>), dim3(nblocks), dim3(256), 0, st, p, wp, partial, g);
}


constexpr int kSpW8Default = kSpExclusive ? 2 : 0;
// Waves per workgroup (per team) of the split-bf16 launch of a plan, and whether two 4-wave teams share a workgroup: the ONE place that decides
// it (launch_split launches what this says; ide3d_modconv_plan reports it).
// 8-wave forms (two of this workgroup's waves per SIMD: exclusive residency without giving up the second wave): IDE3D_SP_W8 bit 0 =
// 3x3 on 8 x 16 pixels, bit 1 = all-class transposed 3x3 (8 x 16 positions at 128 rows, 16 x 16 at 64 rows); IDE3D_MODCONV_SP_W4 = 4 waves
// on the 16 x 16 tiles as well; IDE3D_SP_NO_TEAMS = no two-team workgroups.
struct SpForm { int waves; bool teams; };
static SpForm sp_form(const ConvPlan& pl) {
    const int w8 = knobs().sp_w8 >= 0 ? knobs().sp_w8 : kSpW8Default;
    const bool no_teams = knobs().sp_no_teams, no8 = knobs().sp_w4;
    const bool conv3 = (pl.mode == MODE_CONV3);
    if (pl.pair) return {8, false};
    // 64-row blocks, even block count: two teams per workgroup (128-row blocks: LDS does not fit twice in bf16x6, and in f16x3 the teams measured 233 vs 217 us at 512 -> 256 in@64)
    if (pl.tile == 4) return {4, !conv3 && pl.big == 2 && kSpExclusive && !no_teams && (pl.mblocks & 1) == 0};
    if (pl.tile == 0 || pl.tile == 6) return {(w8 & (conv3 ? 1 : 2)) ? 8 : 4, false};
    if (pl.tile == 12) return {8, false};
    // 16 x 16 pixels: 8 waves with time-shifted roles (3x3: 333 vs 352 us at 128 -> 128 @256, 321 vs 335 at 256 -> 256 @128, bf16x6)
    return {(!no8 && (conv3 || (w8 & 2))) ? 8 : 4, false};
}
template <int MODE, int BIG, int PARTS, int F16 = 0>
static void launch_split(const ide3d_modconv_params& p, const ConvPlan& pl, const float* wp, float* partial, hipStream_t st) {
    const ConvGeom& g = pl.g;
    const unsigned nblocks = (unsigned)((int64_t)g.mblocks * g.tile_base[4] * g.img_groups * g.split_k);
    const u32x4* wu = reinterpret_cast<const u32x4*>(wp);
    const float* ru = F16 ? wp + pl.packed_floats + (int64_t)pl.mblocks * pl.bm : nullptr;       // [row scale | row unscale] behind the packed weights
    const SpForm form = sp_form(pl);
    if (pl.pair) {
        if constexpr (MODE == MODE_TCONV3A) IDE3D_EXCL_LAUNCH((modconv_split_pair_kernel<BIG, (BIG == 1 ? 16 : 32), PARTS, F16>), dim3(2 * nblocks), 512, 0, st, p, wu, partial, g, ru);
        return;
    }
    if (pl.tile == 4) {
        if constexpr (MODE == MODE_TCONV3A) {
            if constexpr (BIG == 2) {
                if (form.teams) {
                    IDE3D_EXCL_LAUNCH((modconv_split_teams_kernel<MODE, BIG, 4, PARTS, 2, F16>), dim3(nblocks / 2), 512, 0, st, p, wu, partial, g, ru);
                    return;
                }
            }
            IDE3D_EXCL_LAUNCH((modconv_split_kernel<MODE, BIG, 4, PARTS, 2, 4, F16>), dim3(nblocks), 256, 0, st, p, wu, partial, g, ru);
        }
    }
    else if (pl.tile == 0 || pl.tile == 6) {
        // 3x3: one weight buffer, two workgroups per CU (measured 338 vs 355 us at 512 -> 512 @64, bf16x6); IDE3D_MODCONV_SP_WBUF2 = old form
        const bool one_wbuf = !knobs().sp_wbuf2 && !kSpExclusive;
        if (form.waves == 8) IDE3D_EXCL_LAUNCH((modconv_split_kernel<MODE, BIG, 8, PARTS, 2, 8, F16>), dim3(nblocks), 512, 0, st, p, wu, partial, g, ru);
        else if (one_wbuf && MODE == MODE_CONV3) IDE3D_EXCL_LAUNCH((modconv_split_kernel<MODE_CONV3, BIG, 8, PARTS, 1, 4, F16>), dim3(nblocks), 256, 0, st, p, wu, partial, g, ru);
        else IDE3D_EXCL_LAUNCH((modconv_split_kernel<MODE, BIG, 8, PARTS, 2, 4, F16>), dim3(nblocks), 256, 0, st, p, wu, partial, g, ru);
    }
    else if (pl.tile == 12) {
        if constexpr (MODE == MODE_CONV3 && BIG == 2)
            IDE3D_EXCL_LAUNCH((modconv_split_kernel<MODE, BIG, 32, PARTS, 2, 8, F16>), dim3(nblocks), 512, 0, st, p, wu, partial, g, ru);
    }
    else if constexpr (MODE == MODE_CONV3 || BIG == 2) {
        if (form.waves == 8) IDE3D_EXCL_LAUNCH((modconv_split_kernel<MODE, BIG, 16, PARTS, 2, 8, F16>), dim3(nblocks), 512, 0, st, p, wu, partial, g, ru);
        else IDE3D_EXCL_LAUNCH((modconv_split_kernel<MODE, BIG, 16, PARTS, 2, 4, F16>), dim3(nblocks), 256, 0, st, p, wu, partial, g, ru);
    }
}

}  // namespace ide3d

// per-image 1x1 convolution without modulation / noise, linear, <= 32 or 161..192 outputs: the dual heads
static bool head_split_applies(const ide3d_modconv_params& p, int arith) {
    const bool off = ide3d::knobs().head_fp32;
    return !off && arith != 1 && p.k == 1 && p.mode == 0 && p.w_batch_stride > 0 && !p.styles && !p.dcoefs && !p.noise && p.act == 1 &&
           (p.cout <= 32 || (p.cout > 160 && p.cout <= 192)) && p.cin >= 32 &&
           (int64_t)p.n * ide3d::cdiv64((int64_t)p.h * p.w_, 32 * ide3d::head_waves(p.cout <= 32 ? 1 : 6)) >= ide3d::kNumCU;     // fewer workgroups: the serial K loop of a workgroup is exposed
                                                                              // (512 channels @32^2 / @64^2: 54 us against 16 / 40 us on the fp32 loop)
}

// per-image linear 1x1 convolution on a small map: one launch of fp32 FMAs (head_small_kernel) in every arithmetic
static bool head_small_applies(const ide3d_modconv_params& p) {
    const bool no_small = ide3d::knobs().head_no_small;
    return !no_small && p.k == 1 && p.mode == 0 && p.w_batch_stride > 0 && !p.styles && !p.dcoefs && !p.noise && p.act == 1 && (int64_t)p.h * p.w_ <= 256 &&
           p.y_pitch == 0 && p.n <= 65535;
}
// resident-weights form of the split heads: whole K in LDS (K = 64 / 128 at <= 32 rows, K = 128 at 192 rows), >= 2 tiles of 32 pixels per wave
static bool head_resident_applies(const ide3d_modconv_params& p) {
    const bool no_resident = ide3d::knobs().head_no_resident;
    const int mt = p.cout <= 32 ? 1 : 6, cchunks = ide3d::cdiv(p.cin, 16);
    const int hw = p.h * p.w_, wgs_img = ide3d::kNumCU / p.n, tiles32 = hw / 32;
    return !no_resident && ide3d::kSpExclusive && p.cin % 16 == 0 && hw % 32 == 0 && wgs_img >= 1 && p.y_pitch == 0 &&
           ((mt == 1 && (cchunks == 4 || cchunks == 8)) || (mt == 6 && cchunks == 8)) && tiles32 >= 2 * wgs_img * 8;
}

static int check_modconv(const ide3d_modconv_params& p) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(p.n > 0 && p.cin > 0 && p.cout > 0 && p.h > 0 && p.w_ > 0, "modconv2d: bad shape");
    IDE3D_CHECK_ARG(p.k == 1 || p.k == 3, "modconv2d: kernel size must be 1 or 3 (got %d)", p.k);
    IDE3D_CHECK_ARG(p.mode == 0 || ((p.mode == 1 || p.mode == 2) && p.k == 3),
                    "modconv2d: mode must be 0 (conv), 1 (3x3 stride-2 conv, no padding) or 2 (3x3 transposed, stride 2)");
    IDE3D_CHECK_ARG(p.mode != 1 || (p.h >= 3 && p.w_ >= 3), "modconv2d: stride-2 convolution needs an input of at least 3x3");
    IDE3D_CHECK_ARG((int64_t)p.n * p.cin * p.h * p.w_ < 0x7fffffffLL, "modconv2d: input too large for 32-bit indexing");
    return IDE3D_OK;
}

extern "C" int64_t ide3d_modconv_workspace_bytes(int32_t n, int32_t cin, int32_t cout, int32_t h, int32_t w, int32_t k, int32_t mode,
                                                 int32_t per_image_weights) {
    using namespace ide3d;
    ide3d_modconv_params p{};
    p.n = n; p.cin = cin; p.cout = cout; p.h = h; p.w_ = w; p.k = k; p.mode = mode; p.w_batch_stride = per_image_weights ? 1 : 0;
    if (check_modconv(p) != IDE3D_OK) return -1;
    p.x = nullptr;                               // alignment of the real tensor is unknown here: size for the larger plan
    // sized for every arithmetic (the packed copy of the split-bf16 loops is the larger one) and for the flattened 1x1 plan
    // ... and for both epilogue classes: the plan depends on the epilogue (the strip / one-round forms of the transposed layers need the plain
    // one: demodulation only), so the sizing plans with the plain epilogue AND with the convolution layers' (noise, bias, activation) and takes
    // the larger of each part — a launch can then never pick a tile / split-K / strip the workspace was not sized for.
    int64_t packed = 0, part = 0;
    static const float dummy = 0.f;
    for (int epi = 0; epi < 2; ++epi) {
        ide3d_modconv_params q0 = p;
        q0.act = 1; q0.gain = 1.f; q0.clamp = -1.f;
        if (epi == 1) { q0.noise = &dummy; q0.bias = &dummy; q0.noise_strength = 1.f; q0.act = 3; q0.alpha = 0.2f; q0.gain = 1.41421356f; }      // never dereferenced: plan_conv only tests them
        for (int arith : {1, 3, 6, 16}) {
            ConvPlan pl; plan_conv(q0, pl, arith);
            ConvPlan pf; plan_conv(flatten_pointwise(q0), pf, arith);
            for (const ConvPlan* q : {&pl, &pf}) {
                if (q->packed_floats + q->aux_floats > packed) packed = q->packed_floats + q->aux_floats;
                if (q->partial_floats > part) part = q->partial_floats;
            }
        }
    }
    int64_t bytes = (packed + part) * (int64_t)sizeof(float);
    if (per_image_weights && k == 1) {                      // packed bf16x6 head weights (head_split_kernel)
        const int64_t hb = head_packed_units(n, cdiv(cin, 16), 3, cout <= 32 ? 1 : 6) * 16;
        if (hb > bytes) bytes = hb;
    }
    return bytes;
}

extern "C" int ide3d_modconv2d(const ide3d_modconv_params* pp, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(pp != nullptr, "modconv2d: null params");
    const ide3d_modconv_params p = flatten_pointwise(*pp);
    IDE3D_CHECK_ARG(p.x && p.w && p.y && p.workspace, "modconv2d: null tensor / workspace pointer");
    int rc = check_modconv(p);
    if (rc) return rc;
    IDE3D_CHECK_ARG(p.act == 1 || p.act == 3, "modconv2d: act must be linear (1) or lrelu (3)");
    IDE3D_CHECK_ARG(p.y_pitch == 0 || (p.mode == 2 && p.y_pitch >= 2 * p.w_ + 1), "modconv2d: y_pitch is the row pitch of a transposed convolution's output (>= 2 w + 1), or 0");
    hipStream_t st_head = (hipStream_t)stream;
    {
        const int64_t hw = (int64_t)p.h * p.w_;
        if (head_small_applies(p)) {
            if (hw <= 64) hipLaunchKernelGGL((head_small_kernel<64, 32>), dim3(cdiv(p.cout, HS_RB), 1, p.n), dim3(HS_THREADS), 0, st_head, p, (int)hw);
            else          hipLaunchKernelGGL((head_small_kernel<256, 8>), dim3(cdiv(p.cout, HS_RB), cdiv((int)hw, 256), p.n), dim3(HS_THREADS), 0, st_head, p, (int)hw);
            IDE3D_CHECK_LAUNCH("modconv2d (small-map heads)");
            return IDE3D_OK;
        }
    }
    if (head_split_applies(p, resolve_arith(p.arith))) {
        const int parts = resolve_arith(p.arith) == 3 ? 2 : 3, mt = p.cout <= 32 ? 1 : 6, cchunks = cdiv(p.cin, 16);     // f16x3: the heads stay on bf16x6
        IDE3D_CHECK_ARG(p.workspace_bytes >= head_packed_units(p.n, cchunks, parts, mt) * 16, "modconv2d: workspace too small for the packed head weights");
        u32x4* wu = reinterpret_cast<u32x4*>(p.workspace);
        const int64_t items = (int64_t)p.n * cchunks * 2 * mt * 32;
        const int tiles = cdiv(p.h * p.w_, 32 * head_waves(mt));
        const int hw_h = p.h * p.w_, wgs_img = kNumCU / p.n, tiles32 = hw_h / 32;
        const bool resident = head_resident_applies(p);
#define IDE3D_HEAD_RES(P, M, C) IDE3D_EXCL_LAUNCH((head_resident_kernel<P, M, C>), dim3(p.n * wgs_img), 512, 0, st_head, p, wu, wgs_img, tiles32)
#define IDE3D_HEAD(P, M) do { \
            hipLaunchKernelGGL(head_pack_split_kernel<P>, dim3(stream_grid(items, 256)), dim3(256), 0, st_head, p.w, p.w_batch_stride, p.n, p.cout, p.cin, M * 32, cchunks, wu); \
            if (resident) { if (M == 6) IDE3D_HEAD_RES(P, 6, 8); else if (cchunks == 4) IDE3D_HEAD_RES(P, 1, 4); else IDE3D_HEAD_RES(P, 1, 8); } \
            else IDE3D_EXCL_LAUNCH((head_split_kernel<P, M>), dim3(p.n * tiles), 64 * head_waves(M), 0, st_head, p, wu, cchunks, tiles); } while (0)
        if (parts == 2) { if (mt == 1) IDE3D_HEAD(2, 1); else IDE3D_HEAD(2, 6); }
        else            { if (mt == 1) IDE3D_HEAD(3, 1); else IDE3D_HEAD(3, 6); }
#undef IDE3D_HEAD
#undef IDE3D_HEAD_RES
        IDE3D_CHECK_LAUNCH("modconv2d (split-bf16 heads)");
        return IDE3D_OK;
    }
    // f16x3 needs the caller's bound on |x| (x_amax); without it the launch runs in bf16x6 (same interface, no range to manage)
    int arith_eff = resolve_arith(p.arith);
    if (arith_eff == 16 && !p.x_amax) arith_eff = 6;
    ConvPlan pl; plan_conv(p, pl, arith_eff);
    IDE3D_CHECK_ARG(p.workspace_bytes >= (pl.packed_floats + pl.aux_floats + pl.partial_floats) * (int64_t)sizeof(float),
                    "modconv2d: workspace too small (need %lld bytes)", (long long)((pl.packed_floats + pl.aux_floats + pl.partial_floats) * sizeof(float)));
    hipStream_t st = (hipStream_t)stream;
    float* wp = p.workspace;
    float* partial = p.workspace + pl.packed_floats + pl.aux_floats;
    if (pl.parts) {
        if (!p.weights_packed) {
            const int64_t items = (int64_t)pl.mblocks * pl.cchunks * 9 * 2 * pl.bm;
            if (pl.f16) {
                float* const rs = wp + pl.packed_floats;                  // [rows] scale, then [rows] unscale
                const int rows = pl.mblocks * pl.bm;
                hipLaunchKernelGGL(modconv_row_scale_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, p.w, p.cout, p.cin * 9, rows, rs, rs + rows);
                hipLaunchKernelGGL((modconv_pack_split_kernel<2, 1>), dim3(stream_grid(items, 256)), dim3(256), 0, st, p.w, p.cout, p.cin, pl.bm, pl.mblocks, pl.cchunks, reinterpret_cast<u32x4*>(wp), rs);
            }
            else if (pl.parts == 2) hipLaunchKernelGGL(modconv_pack_split_kernel<2>, dim3(stream_grid(items, 256)), dim3(256), 0, st, p.w, p.cout, p.cin, pl.bm, pl.mblocks, pl.cchunks, reinterpret_cast<u32x4*>(wp), nullptr);
            else               hipLaunchKernelGGL(modconv_pack_split_kernel<3>, dim3(stream_grid(items, 256)), dim3(256), 0, st, p.w, p.cout, p.cin, pl.bm, pl.mblocks, pl.cchunks, reinterpret_cast<u32x4*>(wp), nullptr);
        }
#define IDE3D_SP_DISPATCH(M) \
        do { if (pl.big == 1) { if (pl.f16) launch_split<M, 1, 2, 1>(p, pl, wp, partial, st); else if (pl.parts == 2) launch_split<M, 1, 2>(p, pl, wp, partial, st); else launch_split<M, 1, 3>(p, pl, wp, partial, st); } \
             else             { if (pl.f16) launch_split<M, 2, 2, 1>(p, pl, wp, partial, st); else if (pl.parts == 2) launch_split<M, 2, 2>(p, pl, wp, partial, st); else launch_split<M, 2, 3>(p, pl, wp, partial, st); } } while (0)
        if (pl.mode == MODE_CONV3) IDE3D_SP_DISPATCH(MODE_CONV3); else IDE3D_SP_DISPATCH(MODE_TCONV3A);
#undef IDE3D_SP_DISPATCH
    } else {
    if (!p.weights_packed) {
        hipLaunchKernelGGL(modconv_pack_kernel, dim3(stream_grid(pl.packed_floats, 256)), dim3(256), 0, st,
                           p.w, p.w_batch_stride, p.w_batch_stride ? p.n : 1, p.cout, p.cin, pl.taps, pl.bm, pl.kc, pl.mblocks, pl.cchunks, wp);
    }
#define IDE3D_MC_DISPATCH(M) \
    do { if (pl.big == 1) launch_tiles<M, 1>(p, pl, wp, partial, st); else if (pl.big == 2) launch_tiles<M, 2>(p, pl, wp, partial, st); \
         else launch_tiles<M, 0>(p, pl, wp, partial, st); } while (0)
    if (pl.mode == MODE_CONV3)       IDE3D_MC_DISPATCH(MODE_CONV3);
    else if (pl.mode == MODE_CONV1)  IDE3D_MC_DISPATCH(MODE_CONV1);
    else if (pl.mode == MODE_CONV3S2) IDE3D_MC_DISPATCH(MODE_CONV3S2);
    else if (pl.mode == MODE_TCONV3A) IDE3D_MC_DISPATCH(MODE_TCONV3A);
    else                             IDE3D_MC_DISPATCH(MODE_TCONV3);
#undef IDE3D_MC_DISPATCH
    }
    if (pl.g.split_k > 1) {
        const int64_t per = (int64_t)p.n * p.cout * pl.oh * pl.ow;
        hipLaunchKernelGGL(modconv_epilogue_kernel, dim3(stream_grid((p.y_pitch <= 0 && ((pl.oh * pl.ow) & 3) == 0) ? per / 4 : per, 256)), dim3(256), 0, st, p, partial, pl.g.split_k, pl.oh, pl.ow);
    }
    if (pl.mode == MODE_TCONV3A && !p.weights_packed) {
        // the strip weights are packed with the main ones, whether or not THIS call's epilogue allows the strip plan: `weights_packed` of a later
        // call vouches for the whole workspace
        hipLaunchKernelGGL(tconv_strip_pack_kernel, dim3(stream_grid(pl.strip_floats, 256)), dim3(256), 0, st, p.w, p.cout, p.cin, cdiv(p.cout, STRIP_CO) * STRIP_CO,
                           wp + pl.packed_floats + pl.strip_off);
    }
    if (pl.strip) {
        const int cout_pad = cdiv(p.cout, STRIP_CO) * STRIP_CO;
        float* const wstrip = wp + pl.packed_floats + pl.strip_off;
        const int nb_re = cdiv(p.w_ + 1, 64), nb_ro = cdiv(p.w_, 64), nb_ce = cdiv(p.h, 64), nb_co = cdiv(p.h, 64);
        float* const xs = wstrip + pl.strip_floats;
        hipLaunchKernelGGL(tconv_strip_gather_kernel, dim3(stream_grid((int64_t)p.n * p.cin * (p.h + p.w_), 256)), dim3(256), 0, st, p.x, p.n * p.cin, p.h, p.w_, xs);
        hipLaunchKernelGGL(tconv_strip_kernel, dim3(nb_re + nb_ro + nb_ce + nb_co, cout_pad / STRIP_CO, p.n), dim3(64 * STRIP_WAVES), 0, st, p, wstrip, xs, cout_pad,
                           pl.oh, pl.ow, (int64_t)(p.y_pitch > 0 ? p.y_pitch : pl.ow), nb_re, nb_ro, nb_ce);
    }
    IDE3D_CHECK_LAUNCH("modconv2d");
    return IDE3D_OK;
}

// Host-only: which kernel family, tile and grid ide3d_modconv2d would launch for these parameters (pointers are not dereferenced; `x` only
// for its alignment).  Same routing functions as the launch itself.
extern "C" int ide3d_modconv_plan(const ide3d_modconv_params* pp, ide3d_modconv_plan_info* out) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(pp != nullptr && out != nullptr, "modconv_plan: null argument");
    const ide3d_modconv_params p = flatten_pointwise(*pp);
    int rc = check_modconv(p);
    if (rc) return rc;
    *out = ide3d_modconv_plan_info{};
    const int arith = resolve_arith(p.arith);
    if (head_small_applies(p)) {
        const int hw = p.h * p.w_;
        out->kind = IDE3D_PLAN_HEAD_SMALL; out->tile_h = 1; out->tile_w = hw <= 64 ? 64 : 256; out->rows = HS_RB; out->waves = HS_THREADS / 64; out->split_k = 1;
        out->workgroups = (int64_t)cdiv(p.cout, HS_RB) * (hw <= 64 ? 1 : cdiv(hw, 256)) * p.n;
        return IDE3D_OK;
    }
    if (head_split_applies(p, arith)) {
        const int mt = p.cout <= 32 ? 1 : 6;
        out->rows = mt * 32; out->parts = arith == 3 ? 2 : 3; out->split_k = 1; out->tile_h = 1;
        if (head_resident_applies(p)) { out->kind = IDE3D_PLAN_HEAD_RESIDENT; out->tile_w = 32; out->waves = 8; out->workgroups = (int64_t)p.n * (kNumCU / p.n); }
        else { out->kind = IDE3D_PLAN_HEAD_SPLIT; out->waves = head_waves(mt); out->tile_w = 32 * out->waves; out->workgroups = (int64_t)p.n * cdiv(p.h * p.w_, out->tile_w); }
        return IDE3D_OK;
    }
    int arith_eff = arith;
    if (arith_eff == 16 && !p.x_amax) arith_eff = 6;
    ConvPlan pl; plan_conv(p, pl, arith_eff);
    const int64_t blocks = (int64_t)pl.g.mblocks * pl.g.tile_base[4] * pl.g.img_groups * pl.g.split_k * (pl.pair ? 2 : 1);
    out->tile_h = PHv[pl.tile]; out->tile_w = PWv[pl.tile]; out->images_per_tile = TIv[pl.tile];
    out->rows = pl.bm; out->parts = pl.parts; out->f16 = pl.f16; out->split_k = pl.g.split_k; out->strip = pl.strip;
    out->transposed_all_class = (pl.mode == MODE_TCONV3A);
    if (pl.parts) {
        const SpForm f = sp_form(pl);
        out->kind = f.teams ? IDE3D_PLAN_SPLIT_TEAMS : IDE3D_PLAN_SPLIT; out->waves = f.teams ? 8 : f.waves; out->workgroups = f.teams ? blocks / 2 : blocks;
    } else {
        out->kind = IDE3D_PLAN_FP32; out->waves = (pl.tile == 8 || pl.tile == 9) ? 8 : 4; out->workgroups = blocks;
    }
    return IDE3D_OK;
}

extern "C" int ide3d_set_conv_arithmetic(int32_t arith) {
    IDE3D_CHECK_ARG(arith == 0 || arith == 1 || arith == 3 || arith == 6 || arith == 16,
                    "set_conv_arithmetic: 0 (environment default), 1 (fp32), 3 (bf16x3), 6 (bf16x6) or 16 (f16x3)");
    ide3d::g_conv_arith = arith;
    return IDE3D_OK;
}
extern "C" int32_t ide3d_get_conv_arithmetic(void) { return ide3d::conv_arith_default(); }

#ifdef IDE3D_MC_TRACE
extern "C" int ide3d_debug_mc(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ide3d::g_mc_dbg), sizeof(unsigned long long) * 256);
}
#endif

namespace ide3d {
const char* modconv_build_flags() {
    return ""
#ifdef IDE3D_SP_SHARED_SIMD
        "!IDE3D_SP_SHARED_SIMD "
#endif
#ifdef IDE3D_MC_TRACE
        "IDE3D_MC_TRACE "
#endif
        ;
}
}  // namespace ide3d
