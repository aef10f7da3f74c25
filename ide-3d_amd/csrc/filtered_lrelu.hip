// filtered_lrelu.hip — fused bias -> up-FIR -> gain*lrelu -> clamp -> down-FIR (StyleGAN3 op) for gfx950.
//
// Semantics follow torch_utils/ops/filtered_lrelu.py:56-153 and the sign-tensor contract of
// filtered_lrelu.cu:494-519 / :562-572 (2-bit codes, 4 per byte: bit0 = negative, bit1 = clamped;
// on read: code&1 -> *slope, code&2 -> 0, outside the sign tensor -> value only scaled by gain).
//
// CDNA4 design (differs from the reference's 31 compile-time variants + __constant__ filter buffer):
// one runtime-parameterised kernel per dtype/sign-mode whose working set lives entirely in the
// CU's 160 KiB LDS.  A 256-thread workgroup owns one (n, c) plane tile and runs four LDS-resident
// passes: input(+bias) -> horizontal up-FIR -> vertical up-FIR + activation (+ sign pack / unpack)
// -> horizontal down-FIR -> vertical down-FIR -> global store.  Filters are staged in LDS from the
// kernel arguments, so concurrent streams are safe (the reference warns it is not:
// filtered_lrelu.py:215-216).  The intermediate (up^2 larger) never touches HBM: traffic = in + out
// (+ signs).  Non-separable filters run the same passes with 2-D tap loops.
#include "common.h"
#include "knobs.h"
#include <stdlib.h>

namespace ide3d {

struct FlrGeom {
    int tow, toh;          // output tile
    int zw, zh;            // activated intermediate tile (multiple of 4 wide)
    int iw, ih;            // input tile
    int off_fu, off_fd, off_in, off_t, off_z, off_d, total;   // LDS carve (floats)
};

__host__ __device__ inline int flr_in_extent(int z, int f, int up) { return (z + f - 1) / up + 2; }

static bool flr_geometry(const ide3d_filtered_lrelu_params& p, int tow, int toh, FlrGeom& g) {
    const int fuw = p.fu_w, fuh = p.fu_h ? p.fu_h : p.fu_w;
    const int fdw = p.fd_w, fdh = p.fd_h ? p.fd_h : p.fd_w;
    g.tow = tow; g.toh = toh;
    g.zw = (((tow - 1) * p.down + fdw) + 3) & ~3;
    g.zh = (toh - 1) * p.down + fdh;
    g.iw = flr_in_extent(g.zw, fuw, p.up);
    g.ih = flr_in_extent(g.zh, fuh, p.up);
    int o = 0;
    g.off_fu = o; o += (p.fu_h ? p.fu_w * p.fu_h : 2 * p.fu_w);
    g.off_fd = o; o += (p.fd_h ? p.fd_w * p.fd_h : 2 * p.fd_w);
    o = (o + 3) & ~3;
    g.off_in = o; o += g.iw * g.ih;                  o = (o + 3) & ~3;
    g.off_t = o;  o += p.fu_h ? 0 : g.ih * g.zw;     o = (o + 3) & ~3;   // horizontal up pass (separable only)
    g.off_z = o;  o += g.zh * g.zw;                  o = (o + 3) & ~3;
    g.off_d = o;  o += p.fd_h ? 0 : g.zh * g.tow;    // horizontal down pass (separable only)
    g.total = o;
    return (size_t)o * sizeof(float) <= 150 * 1024;
}

template <class T, int SIGN>   // SIGN: 0 none, 1 write, 2 read
__global__ void __launch_bounds__(256)
filtered_lrelu_kernel(ide3d_filtered_lrelu_params p, FlrGeom g, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_fu = lds + g.off_fu;
    float* s_fd = lds + g.off_fd;
    float* s_in = lds + g.off_in;
    float* s_t  = lds + g.off_t;
    float* s_z  = lds + g.off_z;
    float* s_d  = lds + g.off_d;
    const int tid = threadIdx.x;
    const bool fu_sep = (p.fu_h == 0), fd_sep = (p.fd_h == 0);
    const int fuw = p.fu_w, fuh = fu_sep ? p.fu_w : p.fu_h;
    const int fdw = p.fd_w, fdh = fd_sep ? p.fd_w : p.fd_h;

    // ---- filters -> LDS, already oriented for correlation (flipped unless p.flip) ----
    if (fu_sep) {
        for (int i = tid; i < fuw; i += 256) {
            const float v = p.fu[(p.flip ? i : fuw - 1 - i) * p.fu_stride[1]];
            s_fu[i] = v; s_fu[fuw + i] = v;
        }
    } else {
        for (int i = tid; i < fuw * fuh; i += 256) {
            const int ky = i / fuw, kx = i - ky * fuw;
            s_fu[i] = p.fu[(p.flip ? ky : fuh - 1 - ky) * p.fu_stride[0] + (p.flip ? kx : fuw - 1 - kx) * p.fu_stride[1]];
        }
    }
    if (fd_sep) {
        for (int i = tid; i < fdw; i += 256) {
            const float v = p.fd[(p.flip ? i : fdw - 1 - i) * p.fd_stride[1]];
            s_fd[i] = v; s_fd[fdw + i] = v;
        }
    } else {
        for (int i = tid; i < fdw * fdh; i += 256) {
            const int ky = i / fdw, kx = i - ky * fdw;
            s_fd[i] = p.fd[(p.flip ? ky : fdh - 1 - ky) * p.fd_stride[0] + (p.flip ? kx : fdw - 1 - kx) * p.fd_stride[1]];
        }
    }

    // ---- tile decomposition ----
    int bid = blockIdx.x;
    const int tix = bid % tiles_x; bid /= tiles_x;
    const int tiy = bid % tiles_y; bid /= tiles_y;
    const int plane = bid, n = plane / p.c, c = plane % p.c;
    const int ox0 = tix * g.tow, oy0 = tiy * g.toh;
    const int zx0 = ox0 * p.down, zy0 = oy0 * p.down;          // intermediate-space origin of the tile
    // input element (0,0) of s_in: smallest input index any tap of z (zx0, zy0) can touch
    const int ix0 = floordiv(zx0 - p.pad_x0, p.up);
    const int iy0 = floordiv(zy0 - p.pad_y0, p.up);

    // ---- pass 0: input + bias -> LDS (zero outside the image) ----
    {
        const T* xp = (const T*)p.x + n * p.x_stride[0] + c * p.x_stride[1];
        const float bias = Elem<T>::ld((const T*)p.b + c);
        for (int i = tid; i < g.iw * g.ih; i += 256) {
            const int ly = i / g.iw, lx = i - ly * g.iw;
            const int iy = iy0 + ly, ix = ix0 + lx;
            float v = 0.f;
            if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w)
                v = Elem<T>::ld(xp + iy * p.x_stride[2] + ix * p.x_stride[3]) + bias;
            s_in[i] = v;
        }
    }
    __syncthreads();

    const float zgain = (float)p.up * (float)p.up * p.gain;

    // ---- pass 1 (separable fu): horizontal up-FIR: s_t[ly, zx] ----
    if (fu_sep) {
        for (int i = tid; i < g.ih * g.zw; i += 256) {
            const int ly = i / g.zw, lzx = i - ly * g.zw;
            const int X0 = zx0 + lzx - p.pad_x0;                // upsampled coordinate of tap kx = 0
            int kx = ((-X0) % p.up + p.up) % p.up;
            int lx = (X0 + kx) / p.up - ix0;                    // exact division
            float acc = 0.f;
            for (; kx < fuw; kx += p.up, ++lx) acc += s_in[ly * g.iw + lx] * s_fu[kx];
            s_t[i] = acc;
        }
        __syncthreads();
    }

    // ---- pass 2: vertical up-FIR (or full 2-D), gain, lrelu, clamp, signs -> s_z ----
    {
        const int64_t s_plane = (int64_t)plane * p.s_h * p.s_w_bytes;
        // 4 consecutive z per thread so that sign bytes are produced / consumed whole.
        const int zw4 = g.zw / 4;
        for (int i = tid; i < g.zh * zw4; i += 256) {
            const int lzy = i / zw4, lzx4 = (i - lzy * zw4) * 4;
            const int Y0 = zy0 + lzy - p.pad_y0;
            const int ky0 = ((-Y0) % p.up + p.up) % p.up;
            const int ly0 = (Y0 + ky0) / p.up - iy0;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lzx = lzx4 + j;
                float acc = 0.f;
                if (fu_sep) {
                    int ly = ly0;
                    for (int ky = ky0; ky < fuh; ky += p.up, ++ly) acc += s_t[ly * g.zw + lzx] * s_fu[fuw + ky];
                } else {
                    const int X0 = zx0 + lzx - p.pad_x0;
                    const int kx0 = ((-X0) % p.up + p.up) % p.up;
                    const int lx0 = (X0 + kx0) / p.up - ix0;
                    int ly = ly0;
                    for (int ky = ky0; ky < fuh; ky += p.up, ++ly) {
                        int lx = lx0;
                        for (int kx = kx0; kx < fuw; kx += p.up, ++lx) acc += s_in[ly * g.iw + lx] * s_fu[ky * fuw + kx];
                    }
                }
                v[j] = acc * zgain;
            }
            const int signX = zx0 + lzx4 + p.s_ofs_x, signY = zy0 + lzy + p.s_ofs_y;
            if (SIGN == 1) {
                unsigned packed = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned code = __float_as_uint(v[j]) >> 31;
                    if (code) v[j] *= p.slope;
                    if (fabsf(v[j]) > p.clamp) { code = 2; v[j] = copysignf(p.clamp, v[j]); }
                    packed |= code << (2 * j);
                }
                // signX is a multiple of 4 by construction (tile origin and s_ofs_x are).
                const int sb = signX >> 2;
                if (signX >= 0 && sb < p.sw_limit && signY >= 0 && signY < p.s_h)
                    p.s[s_plane + (int64_t)signY * p.s_w_bytes + sb] = (uint8_t)packed;
            } else if (SIGN == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sx = signX + j;
                    if (sx >= 0 && (sx >> 2) < p.sw_limit && (sx >> 2) < p.s_w_bytes && signY >= 0 && signY < p.s_h) {
                        const unsigned code = (p.s[s_plane + (int64_t)signY * p.s_w_bytes + (sx >> 2)] >> ((sx & 3) << 1)) & 3u;
                        if (code & 1u) v[j] *= p.slope;
                        if (code & 2u) v[j] = 0.f;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (v[j] < 0.f) v[j] *= p.slope;
                    v[j] = fminf(fmaxf(v[j], -p.clamp), p.clamp);
                }
            }
            *reinterpret_cast<float4*>(&s_z[lzy * g.zw + lzx4]) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    __syncthreads();

    // ---- pass 3 (separable fd): horizontal down-FIR: s_d[lzy, lox] ----
    if (fd_sep) {
        for (int i = tid; i < g.zh * g.tow; i += 256) {
            const int lzy = i / g.tow, lox = i - lzy * g.tow;
            const float* zr = s_z + lzy * g.zw + lox * p.down;
            float acc = 0.f;
            for (int kx = 0; kx < fdw; ++kx) acc += zr[kx] * s_fd[kx];
            s_d[i] = acc;
        }
        __syncthreads();
    }

    // ---- pass 4: vertical down-FIR (or full 2-D) -> global ----
    {
        T* yp = (T*)p.y + n * p.y_stride[0] + c * p.y_stride[1];
        for (int i = tid; i < g.toh * g.tow; i += 256) {
            const int loy = i / g.tow, lox = i - loy * g.tow;
            const int oy = oy0 + loy, ox = ox0 + lox;
            if (oy >= p.out_h || ox >= p.out_w) continue;
            float acc = 0.f;
            if (fd_sep) {
                for (int ky = 0; ky < fdh; ++ky) acc += s_d[(loy * p.down + ky) * g.tow + lox] * s_fd[fdw + ky];
            } else {
                for (int ky = 0; ky < fdh; ++ky) {
                    const float* zr = s_z + (loy * p.down + ky) * g.zw + lox * p.down;
                    for (int kx = 0; kx < fdw; ++kx) acc += zr[kx] * s_fd[ky * fdw + kx];
                }
            }
            Elem<T>::st(yp + oy * p.y_stride[2] + ox * p.y_stride[3], acc);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Specialised kernel: separable filters, compile-time (up, down, taps) — the StyleGAN3 layer shapes
// (filtered_lrelu.cu:1248-1278 instantiates 31 such variants; inversion/networks.py:576-597 picks 6 taps x up / down).
// ------------------------------------------------------------------------------------------------
// Same four LDS-resident passes as the generic kernel, but every inner loop is unrolled over compile-time taps / phases and
// each thread owns a register micro-tile that shares its input window:
//   P1 horizontal up-FIR   4 consecutive z columns of one input row   (window of 4 / UP + taps / UP inputs)
//   P2 vertical up-FIR     one polyphase cell = UP z rows x 4 columns, + gain / lrelu / clamp / sign codes (16-byte LDS rows)
//   P3 horizontal down-FIR 4 consecutive output columns of one z row  (window of 3 DOWN + taps values, 16-byte reads)
//   P4 vertical down-FIR   2 output rows x 4 columns                   (window of DOWN + taps rows), 16-byte stores
// 32 x 32 output tile: 44 KB of LDS at (up 2, down 2, 12 + 12 taps) -> three workgroups per CU instead of the generic kernel's
// one, no integer division / modulo and no filter reads in the inner loops.  The op is VALU-bound, not HBM-bound:
// ~92 multiply-adds per output element (6 + 6 up taps on a 4x larger intermediate, 12 + 12 down taps).
template <int UP, int DOWN, int FUT, int FDT>
struct FlsCfg {
    static_assert(4 % UP == 0 && FUT % UP == 0, "up must divide 4 and the up filter length");
    static constexpr int TOW = 32, TOH = 32;
    static constexpr int NTU = FUT / UP;                                       // up taps per phase
    static constexpr int ZW = (((TOW - 1) * DOWN + FDT) + 3) & ~3;             // activated intermediate tile
    static constexpr int ZH = (TOH - 1) * DOWN + FDT;
    static constexpr int IW = (ZW + 2 * UP - 2) / UP + NTU + 1;                // input tile (covers every phase offset)
    static constexpr int IH = (ZH + 2 * UP - 2) / UP + NTU + 1;
    // round 6: row pitches.  s_in rows are padded so that the horizontal up pass reads its window as 16-byte (up 2) / 8-byte (up 4) vectors, s_t rows
    // to whole 8-column groups so that it writes two 16-byte vectors per item without a bounds check
    static constexpr int P1_VEC = (UP == 2) ? 4 : 2;                           // floats per window read of the aligned horizontal pass
    static constexpr int P1_STEP = 8 / UP;                                     // input columns per 8 z columns
    static constexpr int P1_WIN = P1_STEP + NTU + 1;                           // window floats of an 8-column item
    static constexpr int P1_READS = (P1_WIN + P1_VEC - 1) / P1_VEC;
    static constexpr int ZWP = (ZW + 7) & ~7;                                  // s_t row pitch
    static constexpr int P1_GROUPS = ZWP / 8;
    static constexpr int IWP_MIN = (P1_GROUPS - 1) * P1_STEP + P1_READS * P1_VEC;
    static constexpr int IWP = (((IW > IWP_MIN ? IW : IWP_MIN) + 3) & ~3);     // s_in row pitch
    // s_d row pitch: the vertical down pass reads rows DOWN apart with 8 lanes per row (8 x 16 B = 128 B): a pitch of 32 floats puts every row of a
    // 16-lane group on the same banks (2-way conflict on all 12 window reads per item); pitch = 16 (mod 32) floats alternates the bank halves
    static constexpr int N_IN = IWP * IH, N_T = IH * ZWP, N_Z = ZH * ZW;
    // LDS layout (round 6): [ s_t | s_z ] with s_in ALIASED onto the start of s_z (the input tile is dead once the horizontal up pass has run, s_z is
    // first written by the pass after it) and s_d aliased onto s_t (dead once s_z is complete): 36.9 KB instead of 45.5 at (2, 2, 12, 12), i.e. FOUR
    // workgroups per CU instead of three — the kernel issues vector instructions 61 % of the time at three waves per SIMD (kernel_pmc.json)
    static constexpr int OFF_T = 0;
    static constexpr int OFF_Z = ((N_T > ZH * TOW ? N_T : ZH * TOW) + 3) & ~3;          // (s_d may be the larger one: up 4)
    static constexpr int OFF_IN = OFF_Z;
    static constexpr int DP = (DOWN == 2 && ZH * (TOW + 16) <= OFF_Z) ? TOW + 16 : TOW;          // (where the padded rows still fit the aliased space)
    static constexpr int N_D = ZH * DP;
    static constexpr int TOTAL = OFF_Z + (N_Z > N_IN ? N_Z : N_IN);
    static_assert(N_D <= OFF_Z, "the down-pass buffer must fit the space of the up-pass buffer");
};

template <class T, int UP, int DOWN, int FUT, int FDT, int SIGN>
__global__ void __launch_bounds__(256)
filtered_lrelu_sep_kernel(ide3d_filtered_lrelu_params p, int tiles_x, int tiles_y) {
    using K = FlsCfg<UP, DOWN, FUT, FDT>;
    __shared__ __attribute__((aligned(16))) float lds[K::TOTAL];
    float* const s_in = lds + K::OFF_IN;
    float* const s_t = lds + K::OFF_T;
    float* const s_z = lds + K::OFF_Z;
    float* const s_d = lds;
    const int tid = threadIdx.x;

    // filters oriented for correlation (flipped unless p.flip), in registers (uniform loads)
    // (round 6: one pointer + one signed step per filter instead of a select and a 64-bit multiply per tap: the scalar unit is shared by the CU's 16 waves and
    // the prologue was ~400 scalar instructions per wave)
    float fu[FUT], fd[FDT];
    {
        const int64_t su = p.flip ? p.fu_stride[1] : -p.fu_stride[1], sd = p.flip ? p.fd_stride[1] : -p.fd_stride[1];
        const float* bu = p.fu + (p.flip ? 0 : (FUT - 1) * p.fu_stride[1]);
        const float* bd = p.fd + (p.flip ? 0 : (FDT - 1) * p.fd_stride[1]);
#pragma unroll
        for (int i = 0; i < FUT; ++i) { fu[i] = *bu; bu += su; }
#pragma unroll
        for (int i = 0; i < FDT; ++i) { fd[i] = *bd; bd += sd; }
    }

    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tix = bid % tiles_x; bid /= tiles_x;
    const int tiy = bid % tiles_y; bid /= tiles_y;
    const int plane = bid, n = plane / p.c, c = plane % p.c;
    const int ox0 = tix * K::TOW, oy0 = tiy * K::TOH;
    const int zx0 = ox0 * DOWN, zy0 = oy0 * DOWN;
    const int ix0 = floordiv(zx0 - p.pad_x0, UP), iy0 = floordiv(zy0 - p.pad_y0, UP);
    const int rx0 = (zx0 - p.pad_x0) - UP * ix0, ry0 = (zy0 - p.pad_y0) - UP * iy0;     // phase of local z index 0, in [0, UP)

    // ---- P0: input + bias -> LDS (zero outside the image); all loads issued before the first LDS write.
    // Round 6: lane = column, wave = row (+ 4 k): the row index and its validity are wave-uniform (scalar unit), the column's are computed once,
    // the LDS address is one base + compile-time offsets — ~5 vector instructions per element instead of 27 (the flat index needed a division by
    // the row length, four clamps and 64-bit address arithmetic per element: 245 of the kernel's ~1100 vector instructions per wave).
    {
        const T* xp = (const T*)p.x + n * p.x_stride[0] + c * p.x_stride[1];
        const float bias = Elem<T>::ld((const T*)p.b + c);
        if constexpr (K::IWP <= 64) {
            const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // (scalar: the row arithmetic below then runs on the scalar unit)
            constexpr int NR = (K::IH + 3) / 4;
            const int ix = ix0 + lane;
            const bool col_ok = (unsigned)ix < (unsigned)p.in_w && lane < K::IW;
            const unsigned col_off = (unsigned)((int64_t)min(max(ix, 0), p.in_w - 1) * p.x_stride[3]);          // elements; a plane is < 2^32 elements
            float v[NR];
            const T* rowp = xp + (int64_t)(iy0 + wv) * p.x_stride[2];          // scalar base (rows outside the image are replaced by row 0 before the load) + 32-bit lane offset
            const int64_t row_step = 4 * p.x_stride[2];
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int iy = iy0 + wv + 4 * k;                                 // wave-uniform
                const bool row_ok = (unsigned)iy < (unsigned)p.in_h;
                const float t = Elem<T>::ld((row_ok ? rowp : xp) + col_off);          // (a scalar select: every load is issued, none leaves the plane)
                v[k] = (row_ok && col_ok) ? t + bias : 0.f;
                rowp += row_step;
            }
            if (lane < K::IWP) {
#pragma unroll
                for (int k = 0; k < NR; ++k) { const int ly = wv + 4 * k; if (ly < K::IH) s_in[ly * K::IWP + lane] = v[k]; }
            }
        } else {
            constexpr int NL = (K::N_IN + 255) / 256;
            float v[NL];
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int i = tid + k * 256;
                const int ly = i / K::IWP, lx = i - ly * K::IWP;
                const int iy = iy0 + ly, ix = ix0 + lx;
                const int cy = min(max(iy, 0), p.in_h - 1), cx = min(max(ix, 0), p.in_w - 1);
                const float t = Elem<T>::ld(xp + cy * p.x_stride[2] + cx * p.x_stride[3]);
                v[k] = (iy == cy && ix == cx && lx < K::IW) ? t + bias : 0.f;
            }
#pragma unroll
            for (int k = 0; k < NL; ++k) { const int i = tid + k * 256; if (i < K::N_IN) s_in[i] = v[k]; }
        }
    }
    __syncthreads();

    // ---- P1: horizontal up-FIR.  Thread item = (input row ly, group g): z columns j = 4 g - rx0 + e, e = 0..3, i.e. groups are
    // aligned to the polyphase grid (s = rx0 + j = 4 g + e), so phase and tap set of e are compile-time:
    //   first tap kx0 = (UP - e % UP) % UP, input column lx = ceil(s / UP) + t = 4 g / UP + ceil(e / UP) + t, taps kx0 + t UP.
    // Round 6, when the tile starts on the polyphase grid (rx0 == 0: pad_x0 a multiple of UP — every StyleGAN3 layer): 8 columns per item, the
    // window read as 16- / 8-byte vectors, two 16-byte stores, no bounds checks: 2 rounds of 48 multiply-adds + ~16 other instructions instead
    // of 4 rounds of 24 + 42.
    if (UP > 1 && rx0 == 0) {
        for (int i = tid; i < K::IH * K::P1_GROUPS; i += 256) {
            const int ly = i / K::P1_GROUPS, g = i - ly * K::P1_GROUPS;
            const float* src = s_in + ly * K::IWP + g * K::P1_STEP;
            float w[K::P1_READS * K::P1_VEC];
#pragma unroll
            for (int k = 0; k < K::P1_READS; ++k) {
                if constexpr (K::P1_VEC == 4) { const float4 q = *reinterpret_cast<const float4*>(src + 4 * k); w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w; }
                else { const float2 q = *reinterpret_cast<const float2*>(src + 2 * k); w[2 * k] = q.x; w[2 * k + 1] = q.y; }
            }
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kx0 = (UP - e % UP) % UP, base = (e + UP - 1) / UP;
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < K::NTU; ++t) acc += w[base + t] * fu[kx0 + t * UP];
                o[e] = acc;
            }
            float* dst = s_t + ly * K::ZWP + 8 * g;
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
    } else {
        constexpr int NG = (K::ZW + UP - 1 + 3) / 4 + 1;            // groups covering j = -rx0 .. ZW - 1
        constexpr int WIN = 4 / UP + K::NTU;                        // distinct input columns a group touches (+1 when UP > 1)
        for (int i = tid; i < K::IH * NG; i += 256) {
            const int ly = i / NG, g = i - ly * NG;
            const float* src = s_in + ly * K::IWP + g * (4 / UP);
            float w[WIN + 1];
#pragma unroll
            for (int k = 0; k < WIN + 1; ++k) w[k] = (g * (4 / UP) + k < K::IW) ? src[k] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kx0 = (UP - e % UP) % UP, base = (e + UP - 1) / UP;
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < K::NTU; ++t) acc += w[base + t] * fu[kx0 + t * UP];
                const int j = 4 * g - rx0 + e;
                if (j >= 0 && j < K::ZW) s_t[ly * K::ZWP + j] = acc;
            }
        }
    }
    __syncthreads();

    // ---- P2: vertical up-FIR + gain, lrelu, clamp (+ sign codes).  Thread item = (polyphase cell cy, column group x4): z rows
    // jy = UP cy - ry0 + r, r = 0..UP-1, input rows ly = cy + ceil(r / UP) + t.
    {
        constexpr int NC = (K::ZH + UP - 1 + UP - 1) / UP + 1;       // cells covering jy = -ry0 .. ZH - 1
        constexpr int ZW4 = K::ZW / 4;
        // round 6: the gain rides in the taps of this pass (one multiply per tap and thread at the start instead of one per z value), and the plain
        // (no sign tensor) activation is 3 instructions per value: lrelu as max(v, v * slope) when 0 <= slope <= 1, the clamp as one v_med3
        const float zgain = (float)(UP * UP) * p.gain;
        float fuz[FUT];
#pragma unroll
        for (int i = 0; i < FUT; ++i) fuz[i] = fu[i] * zgain;
        const bool lrelu_max = p.slope >= 0.f && p.slope <= 1.f;
        const int64_t s_plane = (int64_t)plane * p.s_h * p.s_w_bytes;
        for (int i = tid; i < NC * ZW4; i += 256) {
            const int cy = i / ZW4, x4 = (i - cy * ZW4) * 4;
            float4 w[K::NTU + 1];
            static_assert(NC - 1 + K::NTU < K::IH, "every window row of every cell lies inside the up-pass buffer");          // (no per-row guard: 4 selects per row)
#pragma unroll
            for (int k = 0; k < K::NTU + 1; ++k) w[k] = *reinterpret_cast<const float4*>(s_t + (cy + k) * K::ZWP + x4);
#pragma unroll
            for (int r = 0; r < UP; ++r) {
                const int jy = UP * cy - ry0 + r;
                const bool row_ok = (unsigned)jy < (unsigned)K::ZH;          // (only the stores are guarded: no branch between the rows of a cell)
                const int ky0 = (UP - r % UP) % UP, base = (r + UP - 1) / UP;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < K::NTU; ++t) {
                    const float f = fuz[ky0 + t * UP];
                    v[0] += w[base + t].x * f; v[1] += w[base + t].y * f; v[2] += w[base + t].z * f; v[3] += w[base + t].w * f;
                }
                const int signX = zx0 + x4 + p.s_ofs_x, signY = zy0 + jy + p.s_ofs_y;
                if (SIGN == 1) {
                    unsigned packed = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned code = __float_as_uint(v[j]) >> 31;
                        if (code) v[j] *= p.slope;
                        if (fabsf(v[j]) > p.clamp) { code = 2; v[j] = copysignf(p.clamp, v[j]); }
                        packed |= code << (2 * j);
                    }
                    const int sb = signX >> 2;          // signX is a multiple of 4 (tile origin and s_ofs_x are)
                    if (row_ok && signX >= 0 && sb < p.sw_limit && signY >= 0 && signY < p.s_h)
                        p.s[s_plane + (int64_t)signY * p.s_w_bytes + sb] = (uint8_t)packed;
                } else if (SIGN == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int sx = signX + j;
                        if (row_ok && sx >= 0 && (sx >> 2) < p.sw_limit && (sx >> 2) < p.s_w_bytes && signY >= 0 && signY < p.s_h) {
                            const unsigned code = (p.s[s_plane + (int64_t)signY * p.s_w_bytes + (sx >> 2)] >> ((sx & 3) << 1)) & 3u;
                            if (code & 1u) v[j] *= p.slope;
                            if (code & 2u) v[j] = 0.f;
                        }
                    }
                } else {
                    if (lrelu_max) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)          // max(v, v * slope) as a median with +inf: fmaxf costs an extra canonicalising v_max per value
                            v[j] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(v[j], v[j] * p.slope, __builtin_inff()), -p.clamp, p.clamp);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (v[j] < 0.f) v[j] *= p.slope;
                            v[j] = fminf(fmaxf(v[j], -p.clamp), p.clamp);
                        }
                    }
                }
                if (row_ok) *reinterpret_cast<float4*>(s_z + jy * K::ZW + x4) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    __syncthreads();

    // ---- P3: horizontal down-FIR.  Thread item = (z row, 4 consecutive output columns): window of 3 DOWN + FDT values. ----
    {
        constexpr int WIN = 3 * DOWN + FDT, W4 = (WIN + 3) / 4;
        for (int i = tid; i < K::ZH * (K::TOW / 4); i += 256) {
            const int zy = i / (K::TOW / 4), o4 = (i - zy * (K::TOW / 4)) * 4;
            const float* zr = s_z + zy * K::ZW + o4 * DOWN;
            float w[W4 * 4];
#pragma unroll
            for (int k = 0; k < W4; ++k) {
                const float4 q = (o4 * DOWN + 4 * k + 3 < K::ZW) ? *reinterpret_cast<const float4*>(zr + 4 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
                w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w;
            }
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < FDT; ++k) acc[e] += w[e * DOWN + k] * fd[k];
            *reinterpret_cast<float4*>(s_d + zy * K::DP + o4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
    __syncthreads();

    // ---- P4: vertical down-FIR -> global.  Thread item = (1 output row, 4 columns): window of FDT rows.  (Rounds 2-5: 2 rows per item
    // = 128 items for 256 threads — half the lanes idle through a quarter of the tile's multiply-adds.) ----
    {
        T* yp = (T*)p.y + n * p.y_stride[0] + c * p.y_stride[1];
        static_assert(K::TOH * (K::TOW / 4) == 256, "one item per thread");
        const int o1 = tid / (K::TOW / 4), o4 = (tid % (K::TOW / 4)) * 4;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < FDT; ++k) {
            const float4 w = *reinterpret_cast<const float4*>(s_d + (o1 * DOWN + k) * K::DP + o4);          // (o1 * DOWN + k <= (TOH - 1) DOWN + FDT - 1 = ZH - 1)
            a[0] += w.x * fd[k]; a[1] += w.y * fd[k]; a[2] += w.z * fd[k]; a[3] += w.w * fd[k];
        }
        const int oy = oy0 + o1, ox = ox0 + o4;
        if (oy < p.out_h) {
            T* yr = yp + oy * p.y_stride[2] + ox * p.y_stride[3];
            bool done = false;
            if constexpr (sizeof(T) == 4) {
                if (ox + 3 < p.out_w && p.y_stride[3] == 1 && ((reinterpret_cast<uintptr_t>(yr) & 15) == 0)) {
                    *reinterpret_cast<float4*>(yr) = make_float4(a[0], a[1], a[2], a[3]);
                    done = true;
                }
            }
            if (!done) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (ox + j < p.out_w) Elem<T>::st(yr + j * p.y_stride[3], a[j]);
            }
        }
    }
}

template <class T, int UP, int DOWN, int FUT, int FDT>
static int launch_flr_sep(const ide3d_filtered_lrelu_params& p, hipStream_t st) {
    using K = FlsCfg<UP, DOWN, FUT, FDT>;
    const int tiles_x = cdiv(p.out_w, K::TOW), tiles_y = cdiv(p.out_h, K::TOH);
    const int64_t nblocks = (int64_t)tiles_x * tiles_y * p.n * p.c;
    if (nblocks > 0x7fffffff) { set_error("filtered_lrelu: grid too large"); return IDE3D_EINVAL; }
    if (p.sign_mode == 1)      hipLaunchKernelGGL((filtered_lrelu_sep_kernel<T, UP, DOWN, FUT, FDT, 1>), dim3((unsigned)nblocks), dim3(256), 0, st, p, tiles_x, tiles_y);
    else if (p.sign_mode == 2) hipLaunchKernelGGL((filtered_lrelu_sep_kernel<T, UP, DOWN, FUT, FDT, 2>), dim3((unsigned)nblocks), dim3(256), 0, st, p, tiles_x, tiles_y);
    else                       hipLaunchKernelGGL((filtered_lrelu_sep_kernel<T, UP, DOWN, FUT, FDT, 0>), dim3((unsigned)nblocks), dim3(256), 0, st, p, tiles_x, tiles_y);
    IDE3D_CHECK_LAUNCH("filtered_lrelu");
    return IDE3D_OK;
}

// Returns 1 when a specialised instance ran, 0 when none matches (the generic kernel then takes the call), < 0 on error.
template <class T>
static int try_flr_sep(const ide3d_filtered_lrelu_params& p, hipStream_t st) {
    const bool off = knobs().flr_generic;
    if (off || p.fu_h != 0 || p.fd_h != 0) return 0;                // separable filters only
#define IDE3D_FLS(U, D, FU, FD) \
    if (p.up == U && p.down == D && p.fu_w == FU && p.fd_w == FD) { const int rc = launch_flr_sep<T, U, D, FU, FD>(p, st); return rc ? rc : 1; }
    IDE3D_FLS(2, 2, 12, 12)
    IDE3D_FLS(4, 2, 12, 12)
    IDE3D_FLS(4, 2, 24, 12)
    IDE3D_FLS(2, 2, 8, 8)
    IDE3D_FLS(2, 1, 12, 1)
    IDE3D_FLS(1, 2, 1, 12)
#undef IDE3D_FLS
    return 0;
}

template <class T>
static int launch_flr(const ide3d_filtered_lrelu_params& p, hipStream_t st) {
    FlrGeom g;
    // Largest tile that fits the LDS budget; (tow * down) % 4 == 0 keeps sign bytes tile-aligned.
    static const int cand[][2] = {{64, 32}, {32, 32}, {32, 16}, {16, 16}, {16, 8}, {8, 8}, {4, 4}};
    bool ok = false;
    for (auto& cd : cand) {
        if ((cd[0] * p.down) % 4 != 0) continue;
        if (flr_geometry(p, cd[0], cd[1], g)) { ok = true; break; }
    }
    if (!ok) { set_error("filtered_lrelu: tile does not fit LDS for this filter configuration"); return IDE3D_ENOKERNEL; }
    const int tiles_x = cdiv(p.out_w, g.tow), tiles_y = cdiv(p.out_h, g.toh);
    const int64_t nblocks = (int64_t)tiles_x * tiles_y * p.n * p.c;
    if (nblocks > 0x7fffffff) { set_error("filtered_lrelu: grid too large"); return IDE3D_EINVAL; }
    const size_t lds_bytes = (size_t)g.total * sizeof(float);
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), lds_bytes, st, p, g, tiles_x, tiles_y);
    };
    if (p.sign_mode == 1)      go(filtered_lrelu_kernel<T, 1>);
    else if (p.sign_mode == 2) go(filtered_lrelu_kernel<T, 2>);
    else                       go(filtered_lrelu_kernel<T, 0>);
    IDE3D_CHECK_LAUNCH("filtered_lrelu");
    return IDE3D_OK;
}

// ------------------------------------------------------------------------------------------------
// Stand-alone activation + sign kernel (generic fallback path of the reference, filtered_lrelu.cu:1105)
// ------------------------------------------------------------------------------------------------

template <class T, int SIGN>
__global__ void __launch_bounds__(256)
filtered_lrelu_act_kernel(T* __restrict__ x, uint8_t* __restrict__ s, int n, int c, int h, int w,
                          int64_t sn, int64_t sc, int64_t sh, int64_t sw_,
                          int s_w, int s_h, int s_ofs_x, int s_ofs_y, float gain, float slope, float clamp) {
    // One thread handles 4 horizontally consecutive elements = one sign byte.
    const int gw = SIGN == 1 ? s_w : w;                 // logical launch width
    const int gh = SIGN == 1 ? s_h : h;
    const int gw4 = (gw + 3) / 4;
    const int64_t total = (int64_t)n * c * gh * gw4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        int64_t r = idx;
        const int x4 = (int)(r % gw4) * 4; r /= gw4;
        const int yy = (int)(r % gh); r /= gh;
        const int64_t q = r;                             // n * c + c
        const int cc = (int)(q % c), nn = (int)(q / c);
        T* row = x + nn * sn + cc * sc + yy * sh;
        unsigned packed = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = x4 + j;
            if (xx < w && yy < h) {
                float v = (float)Elem<T>::ld(row + xx * sw_) * gain;
                if (SIGN == 1) {
                    unsigned code = 0;
                    if (v < 0.f) { v *= slope; code = 1; }
                    if (fabsf(v) > clamp) { v = copysignf(clamp, v); code = 2; }
                    packed |= code << (2 * j);
                } else if (SIGN == 2) {
                    const int sx = xx + s_ofs_x, sy = yy + s_ofs_y;
                    if (sx >= 0 && sx < s_w && sy >= 0 && sy < s_h) {
                        const unsigned code = (s[(q * s_h + sy) * (int64_t)(s_w >> 2) + (sx >> 2)] >> ((sx & 3) << 1)) & 3u;
                        if (code & 1u) v *= slope;
                        if (code & 2u) v = 0.f;
                    }
                } else {
                    if (v < 0.f) v *= slope;
                    if (fabsf(v) > clamp) v = copysignf(clamp, v);
                }
                Elem<T>::st(row + xx * sw_, (typename Elem<T>::math_t)v);
            }
        }
        if (SIGN == 1 && x4 < s_w)
            s[(q * s_h + yy) * (int64_t)(s_w >> 2) + (x4 >> 2)] = (uint8_t)packed;
    }
}

template <class T>
static int launch_act(void* x, uint8_t* s, int n, int c, int h, int w, const int64_t* xs,
                      int s_w, int s_h, int sox, int soy, float gain, float slope, float clamp, int sign_mode, hipStream_t st) {
    const int gw = sign_mode == 1 ? s_w : w, gh = sign_mode == 1 ? s_h : h;
    const int64_t total = (int64_t)n * c * gh * ((gw + 3) / 4);
    const int grid = stream_grid(total, 256);
#define IDE3D_ACT_LAUNCH(S) hipLaunchKernelGGL((filtered_lrelu_act_kernel<T, S>), dim3(grid), dim3(256), 0, st, \
        (T*)x, s, n, c, h, w, xs[0], xs[1], xs[2], xs[3], s_w, s_h, sox, soy, gain, slope, clamp)
    if (sign_mode == 1) IDE3D_ACT_LAUNCH(1); else if (sign_mode == 2) IDE3D_ACT_LAUNCH(2); else IDE3D_ACT_LAUNCH(0);
#undef IDE3D_ACT_LAUNCH
    IDE3D_CHECK_LAUNCH("filtered_lrelu_act");
    return IDE3D_OK;
}

}  // namespace ide3d

extern "C" int ide3d_filtered_lrelu(const ide3d_filtered_lrelu_params* pp, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(pp != nullptr, "filtered_lrelu: null params");
    const ide3d_filtered_lrelu_params& p = *pp;
    IDE3D_CHECK_ARG(p.x && p.y && p.b && p.fu && p.fd, "filtered_lrelu: null tensor pointer");
    IDE3D_CHECK_ARG(p.n > 0 && p.c > 0 && p.in_h > 0 && p.in_w > 0, "filtered_lrelu: x is empty");
    IDE3D_CHECK_ARG(p.up >= 1 && p.down >= 1, "filtered_lrelu: up and down must be at least 1");
    IDE3D_CHECK_ARG(p.fu_w >= 1 && p.fd_w >= 1 && p.fu_h >= 0 && p.fd_h >= 0, "filtered_lrelu: bad filter shape");
    IDE3D_CHECK_ARG(p.out_h >= 1 && p.out_w >= 1, "filtered_lrelu: output must be at least 1x1");
    IDE3D_CHECK_ARG(p.sign_mode >= 0 && p.sign_mode <= 2, "filtered_lrelu: bad sign_mode");
    IDE3D_CHECK_ARG(p.sign_mode == 0 || p.s != nullptr, "filtered_lrelu: sign tensor missing");
    if (p.sign_mode == 1 && (p.s_ofs_x % 4) != 0) {
        set_error("filtered_lrelu: sign write needs s_ofs_x %% 4 == 0");
        return IDE3D_ENOKERNEL;
    }
    const int64_t taps = (int64_t)(p.fu_h ? p.fu_w * p.fu_h : p.fu_w) + (p.fd_h ? p.fd_w * p.fd_h : p.fd_w);
    if (taps > 8192) { set_error("filtered_lrelu: filters too large for the fused kernel"); return IDE3D_ENOKERNEL; }
    hipStream_t st = (hipStream_t)stream;
    switch (p.dtype) {
    case IDE3D_F32:  { const int r = try_flr_sep<float>(p, st); if (r) return r < 0 ? r : IDE3D_OK; return launch_flr<float>(p, st); }
    case IDE3D_F16:  { const int r = try_flr_sep<__half>(p, st); if (r) return r < 0 ? r : IDE3D_OK; return launch_flr<__half>(p, st); }
    case IDE3D_BF16: return launch_flr<__hip_bfloat16>(p, st);
    }
    set_error("filtered_lrelu: no kernel for dtype code %d", p.dtype);
    return IDE3D_ENOKERNEL;
}

extern "C" int ide3d_filtered_lrelu_act(void* x, uint8_t* s, int dtype,
                                        int32_t n, int32_t c, int32_t h, int32_t w,
                                        const int64_t x_stride[4],
                                        int32_t s_w, int32_t s_h, int32_t s_ofs_x, int32_t s_ofs_y,
                                        float gain, float slope, float clamp, int sign_mode, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(x && x_stride, "filtered_lrelu_act: null pointer");
    IDE3D_CHECK_ARG(n > 0 && c > 0 && h > 0 && w > 0, "filtered_lrelu_act: x is empty");
    IDE3D_CHECK_ARG(sign_mode >= 0 && sign_mode <= 2, "filtered_lrelu_act: bad sign_mode");
    IDE3D_CHECK_ARG(sign_mode == 0 || (s != nullptr && (s_w % 4) == 0), "filtered_lrelu_act: bad sign tensor");
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case IDE3D_F32:  return launch_act<float>(x, s, n, c, h, w, x_stride, s_w, s_h, s_ofs_x, s_ofs_y, gain, slope, clamp, sign_mode, st);
    case IDE3D_F16:  return launch_act<__half>(x, s, n, c, h, w, x_stride, s_w, s_h, s_ofs_x, s_ofs_y, gain, slope, clamp, sign_mode, st);
    case IDE3D_BF16: return launch_act<__hip_bfloat16>(x, s, n, c, h, w, x_stride, s_w, s_h, s_ofs_x, s_ofs_y, gain, slope, clamp, sign_mode, st);
    case IDE3D_F64:  return launch_act<double>(x, s, n, c, h, w, x_stride, s_w, s_h, s_ofs_x, s_ofs_y, gain, slope, clamp, sign_mode, st);
    }
    set_error("filtered_lrelu_act: unsupported dtype code %d", dtype);
    return IDE3D_EINVAL;
}
