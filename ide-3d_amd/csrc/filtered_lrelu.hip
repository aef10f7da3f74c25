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
    case IDE3D_F32:  return launch_flr<float>(p, st);
    case IDE3D_F16:  return launch_flr<__half>(p, st);
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
