// resample.hip — the two small resampling steps that sit between the big kernels of G.synthesis.
//
// 1. ide3d_skip_upsample_add_cl: the LAST skip accumulation of the tri-plane backbone,
//        img = upsample2d(img_lo, [1,3,3,1]) + torgb(x)        (inversion/networks.py:1100-1111, upfirdn2d.py:313-349)
//    written straight into a channels-last tensor.  The ray-marcher gathers 32-channel texels, so the tri-planes must be
//    channels-last ([N, H, W, 3C]: a bilinear tap = one 128-byte line); producing them NCHW and transposing afterwards cost
//    two 100 MB read + write passes per tri-plane (239 us per synthesis pass of ATen `direct_copy`).  Here the NCHW inputs
//    are read coalesced along x, a 8 x 32-pixel x 32-channel tile is transposed through LDS, and every pixel's 128 bytes
//    leave as one line.  FIR: up = 2, 4-tap [1,3,3,1] / 8 per axis with gain 2 per axis — polyphase weights (1,3)/4 and (3,1)/4,
//    zero padding outside the image (pad (2,1,2,1) of upsample2d).
// 2. ide3d_bilinear_up2_split: `torch.nn.functional.interpolate(mode='bilinear', align_corners=False)` by exactly 2x of the
//    composited feature image, split into the three tensors the super-resolution blocks take (x: colour features, img: raw
//    RGB = first channels, seg: semantic logits) — one launch instead of three ATen launches (79 us per pass).
//    ATen semantics (UpSample.h `area_pixel_compute_source_index`): src = 0.5 * (dst + 0.5) - 0.5 clamped at 0,
//    i0 = (int)src, i1 = i0 + (i0 < size - 1), lambda1 = src - i0, lambda0 = 1 - lambda1,
//    out = l0y * (l0x * v00 + l1x * v01) + l1y * (l0x * v10 + l1x * v11).
#include "common.h"

namespace ide3d {
namespace {

constexpr int SK_ROWS = 8, SK_COLS = 32, SK_CH = 32;            // output tile: 8 x 32 pixels x 32 channels
constexpr int SK_LR = SK_ROWS / 2 + 2, SK_LC = SK_COLS / 2 + 2; // low-resolution patch 6 x 18
constexpr int SK_TP = SK_CH + 1;                                // transposed tile pitch (floats)

struct SkipArgs {
    const float* lo; const float* add; float* out;
    int64_t lo_s[4], add_s[4];           // element strides [n, c, y, x]
    int n, c, h, w;                      // low-resolution size; output 2h x 2w
    int tiles_x, tiles_y, cgroups;
};

__global__ void __launch_bounds__(256)
skip_upsample_add_cl_kernel(const SkipArgs p) {
    __shared__ float s_lo[SK_CH][SK_LR][SK_LC + 1];
    __shared__ float s_t[SK_ROWS * SK_COLS][SK_TP];
    const int tid = threadIdx.x;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = b % p.cgroups; b /= p.cgroups;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y; b /= p.tiles_y;
    const int n = b;
    const int c0 = cg * SK_CH, Y0 = ty * SK_ROWS, X0 = tx * SK_COLS;
    const int H = 2 * p.h, W = 2 * p.w;
    // ---- low-resolution patch (zero outside the image) and this thread's 32 `add` values: every global load of the tile is
    // issued before the first one is used (one HBM round trip per workgroup instead of one per channel) ----
    const int m0 = Y0 / 2 - 1, q0 = X0 / 2 - 1;
    constexpr int NLO = (SK_CH * SK_LR * SK_LC + 255) / 256;
    float lo_v[NLO];
#pragma unroll
    for (int i = 0; i < NLO; ++i) {
        const int e = tid + i * 256;
        const int c = e / (SK_LR * SK_LC), r = e % (SK_LR * SK_LC), ry = r / SK_LC, rx = r % SK_LC;
        const int yy = m0 + ry, xx = q0 + rx;
        const bool ok = e < SK_CH * SK_LR * SK_LC && c0 + c < p.c && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
        const int64_t off = n * p.lo_s[0] + (int64_t)min(c0 + c, p.c - 1) * p.lo_s[1] + (int64_t)min(max(yy, 0), p.h - 1) * p.lo_s[2]
                          + (int64_t)min(max(xx, 0), p.w - 1) * p.lo_s[3];
        const float v = p.lo[off];                       // unconditional load from a clamped address, masked afterwards
        lo_v[i] = ok ? v : 0.f;
    }
    const int row = tid >> 5, x = tid & 31;
    const int Y = Y0 + row, X = X0 + x;
    const bool ok = Y < H && X < W;
    float add_v[SK_CH];
    {
        const float* ap = p.add + n * p.add_s[0] + (int64_t)min(Y, H - 1) * p.add_s[2] + (int64_t)min(X, W - 1) * p.add_s[3];
#pragma unroll
        for (int c = 0; c < SK_CH; ++c) add_v[c] = ap[(int64_t)min(c0 + c, p.c - 1) * p.add_s[1]];
    }
#pragma unroll
    for (int i = 0; i < NLO; ++i) {
        const int e = tid + i * 256;
        if (e < SK_CH * SK_LR * SK_LC) {
            const int c = e / (SK_LR * SK_LC), r = e % (SK_LR * SK_LC);
            s_lo[c][r / SK_LC][r % SK_LC] = lo_v[i];
        }
    }
    __syncthreads();
    // ---- FIR + add in the NCHW thread layout (x fastest: 128-byte runs of `add`), result transposed into s_t ----
    {
        // even output index 2m: (1 * in[m-1] + 3 * in[m]) / 4; odd 2m+1: (3 * in[m] + 1 * in[m+1]) / 4
        const int ly = (Y >> 1) - m0 - 1 + (Y & 1), lx = (X >> 1) - q0 - 1 + (X & 1);     // first of the two taps in the patch
        const float wy0 = (Y & 1) ? 0.75f : 0.25f, wy1 = 1.0f - wy0;
        const float wx0 = (X & 1) ? 0.75f : 0.25f, wx1 = 1.0f - wx0;
#pragma unroll
        for (int c = 0; c < SK_CH; ++c) {
            const float a = s_lo[c][ly][lx], bq = s_lo[c][ly][lx + 1], cq = s_lo[c][ly + 1][lx], d = s_lo[c][ly + 1][lx + 1];
            float v = wy0 * (wx0 * a + wx1 * bq) + wy1 * (wx0 * cq + wx1 * d);
            if (ok && c0 + c < p.c) v += add_v[c];
            s_t[tid][c] = v;
        }
    }
    __syncthreads();
    // ---- channels-last store: lane = (pixel, 4-channel slice); 8 pixels x 128 bytes per wave instruction ----
    {
        const int j = tid & 7;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int px = (tid >> 3) + 32 * k, row = px >> 5, x = px & 31;
            const int Y = Y0 + row, X = X0 + x;
            if (Y >= H || X >= W) continue;
            float* o = p.out + (((int64_t)n * H + Y) * W + X) * p.c + c0 + 4 * j;
            const float* t = &s_t[px][4 * j];
            if (c0 + 4 * j + 3 < p.c) {
                typedef float f32x4_t __attribute__((ext_vector_type(4)));
                const f32x4_t v = {t[0], t[1], t[2], t[3]};
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(o));
            } else {
                for (int e = 0; e < 4; ++e) if (c0 + 4 * j + e < p.c) o[e] = t[e];
            }
        }
    }
}

struct BilinArgs {
    const float* x; int n, c, h, w;
    float* dst[3]; int c_begin[3], c_count[3];      // output k takes input channels [c_begin, c_begin + c_count)
    int64_t bs[3];                                  // floats between images of output k
};

__global__ void __launch_bounds__(256)
bilinear_up2_split_kernel(const BilinArgs p) {
    // a thread produces 4 consecutive output pixels of one row (w % 2 == 0 checked on the host, so 2w % 4 == 0): they need input
    // columns x0 - 1 .. x0 + 2 of two input rows; one 16-byte store
    const int H = 2 * p.h, W = 2 * p.w, W4 = W / 4;
    const int ctot = p.c_count[0] + p.c_count[1] + p.c_count[2];
    const int64_t total = (int64_t)p.n * ctot * H * W4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int64_t r = i;
        const int X = (int)(r % W4) * 4; r /= W4;
        const int Y = (int)(r % H); r /= H;
        int cc = (int)(r % ctot);
        const int n = (int)(r / ctot);
        int k = 0;
        if (cc >= p.c_count[0]) { cc -= p.c_count[0]; k = 1; if (cc >= p.c_count[1]) { cc -= p.c_count[1]; k = 2; } }
        const int ci = p.c_begin[k] + cc;
        const float sy = fmaxf(0.5f * ((float)Y + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)sy, y1 = y0 + (y0 < p.h - 1);
        const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
        const float* r0 = p.x + ((int64_t)n * p.c + ci) * ((int64_t)p.h * p.w) + (int64_t)y0 * p.w;
        const float* r1 = r0 + (int64_t)(y1 - y0) * p.w;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sx = fmaxf(0.5f * ((float)(X + e) + 0.5f) - 0.5f, 0.f);
            const int x0 = (int)sx, x1 = x0 + (x0 < p.w - 1);
            const float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
            o[e] = ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]);
        }
        *reinterpret_cast<float4*>(p.dst[k] + (int64_t)n * p.bs[k] + ((int64_t)cc * H + Y) * W + X) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace
}  // namespace ide3d

extern "C" int ide3d_skip_upsample_add_cl(const float* lo, const int64_t lo_stride[4], const float* add, const int64_t add_stride[4],
                                          int32_t n, int32_t c, int32_t h, int32_t w, float* out, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(lo && add && out && lo_stride && add_stride, "skip_upsample_add_cl: null pointer");
    IDE3D_CHECK_ARG(n > 0 && c > 0 && h > 0 && w > 0, "skip_upsample_add_cl: bad shape");
    IDE3D_CHECK_ARG(c % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "skip_upsample_add_cl: channels must be a multiple of 4 and out 16-byte aligned");
    SkipArgs a;
    a.lo = lo; a.add = add; a.out = out;
    for (int i = 0; i < 4; ++i) { a.lo_s[i] = lo_stride[i]; a.add_s[i] = add_stride[i]; }
    a.n = n; a.c = c; a.h = h; a.w = w;
    a.tiles_x = cdiv(2 * w, SK_COLS); a.tiles_y = cdiv(2 * h, SK_ROWS); a.cgroups = cdiv(c, SK_CH);
    const int64_t blocks = (int64_t)n * a.tiles_x * a.tiles_y * a.cgroups;
    IDE3D_CHECK_ARG(blocks < 0x7fffffffLL, "skip_upsample_add_cl: too many tiles");
    hipLaunchKernelGGL(skip_upsample_add_cl_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    IDE3D_CHECK_LAUNCH("skip_upsample_add_cl");
    return IDE3D_OK;
}

extern "C" int ide3d_bilinear_up2_split(const float* x, int32_t n, int32_t c, int32_t h, int32_t w,
                                        float* const dst[3], const int32_t c_begin[3], const int32_t c_count[3], const int64_t* dst_batch_floats, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(x && dst && c_begin && c_count, "bilinear_up2_split: null pointer");
    IDE3D_CHECK_ARG(n > 0 && c > 0 && h > 0 && w > 0, "bilinear_up2_split: bad shape");
    IDE3D_CHECK_ARG(w % 2 == 0, "bilinear_up2_split: input width must be even (16-byte output stores)");
    BilinArgs a;
    a.x = x; a.n = n; a.c = c; a.h = h; a.w = w;
    int64_t total = 0;
    for (int k = 0; k < 3; ++k) {
        a.dst[k] = dst[k]; a.c_begin[k] = c_begin[k]; a.c_count[k] = c_count[k];
        const int64_t dense = (int64_t)c_count[k] * 4 * h * w;
        a.bs[k] = (dst_batch_floats && dst_batch_floats[k]) ? dst_batch_floats[k] : dense;
        IDE3D_CHECK_ARG(a.bs[k] >= dense && a.bs[k] % 4 == 0, "bilinear_up2_split: batch stride of output %d is smaller than an image or not a multiple of 4 floats", k);
        IDE3D_CHECK_ARG(c_count[k] >= 0 && c_begin[k] >= 0 && c_begin[k] + c_count[k] <= c && (c_count[k] == 0 || dst[k]),
                        "bilinear_up2_split: channel range %d outside the input", k);
        IDE3D_CHECK_ARG(c_count[k] == 0 || (reinterpret_cast<uintptr_t>(dst[k]) & 15) == 0, "bilinear_up2_split: output %d is not 16-byte aligned", k);
        total += (int64_t)n * c_count[k] * 2 * h * (w / 2);
    }
    if (total == 0) return IDE3D_OK;
    hipLaunchKernelGGL(bilinear_up2_split_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
    IDE3D_CHECK_LAUNCH("bilinear_up2_split");
    return IDE3D_OK;
}
