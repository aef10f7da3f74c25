// frame.hip — RGB + coloured segmentation -> one uint8 frame.
//
// Fuses `mask2color` (dnnlib/seg_tools.py:75-81: argmax over the class channel, first maximum wins,
// then a palette look-up), the [-1,1] -> uint8 conversion of `layout_grid`
// (dnnlib/util.py:637: (x * 127.5 + 128).clamp(0, 255).to(uint8), truncating) and the side-by-side
// `image_seg` concatenation of gen_videos.py:133-135 into one pass: the fp32 seg logits
// (classes x 4 B per pixel) are read once and never written back; 6 bytes per pixel leave the chip.
// Each lane converts 4 horizontally adjacent pixels: 16-byte loads per channel, three 4-byte stores.
#include "common.h"

namespace ide3d {

__device__ __forceinline__ unsigned to_u8(float v) {
    float t = v * 127.5f + 128.0f;
    t = fminf(fmaxf(t, 0.0f), 255.0f);
    return (unsigned)t;     // truncation, like .to(torch.uint8)
}

__global__ void __launch_bounds__(256)
frame_u8_kernel(const float* __restrict__ img, const float* __restrict__ seg, const uint8_t* __restrict__ palette,
                int n, int classes, int H, int W, uint8_t* __restrict__ out) {
    const int W4 = W / 4;
    const int64_t total = (int64_t)n * H * W4;
    const int64_t hw = (int64_t)H * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        int64_t r = idx;
        const int x4 = (int)(r % W4) * 4; r /= W4;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        const int64_t pix = (int64_t)y * W + x4;
        // RGB
        unsigned rgb[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(img + ((int64_t)b * 3 + c) * hw + pix);
            rgb[0][c] = to_u8(v.x); rgb[1][c] = to_u8(v.y); rgb[2][c] = to_u8(v.z); rgb[3][c] = to_u8(v.w);
        }
        // argmax over classes (first maximum wins; NaN counts as maximal like torch.argmax)
        float best[4]; int arg[4];
        {
            const float4 v = *reinterpret_cast<const float4*>(seg + ((int64_t)b * classes) * hw + pix);
            best[0] = v.x; best[1] = v.y; best[2] = v.z; best[3] = v.w;
            arg[0] = arg[1] = arg[2] = arg[3] = 0;
        }
        for (int c = 1; c < classes; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(seg + ((int64_t)b * classes + c) * hw + pix);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool take = (vv[j] > best[j]) || (vv[j] != vv[j] && best[j] == best[j]);
                if (take) { best[j] = vv[j]; arg[j] = c; }
            }
        }
        unsigned sc[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) sc[j][c] = palette[arg[j] * 3 + c];
        // pack 4 pixels x 3 bytes = 3 dwords, for both halves of the frame row
        uint8_t* orow = out + (((int64_t)b * H + y) * (2 * W)) * 3;
        auto pack = [](const unsigned (*px)[3], unsigned* w) {
            w[0] = px[0][0] | (px[0][1] << 8) | (px[0][2] << 16) | (px[1][0] << 24);
            w[1] = px[1][1] | (px[1][2] << 8) | (px[2][0] << 16) | (px[2][1] << 24);
            w[2] = px[2][2] | (px[3][0] << 8) | (px[3][1] << 16) | (px[3][2] << 24);
        };
        unsigned w0[3], w1[3];
        pack(rgb, w0); pack(sc, w1);
        unsigned* o0 = reinterpret_cast<unsigned*>(orow + (int64_t)x4 * 3);
        unsigned* o1 = reinterpret_cast<unsigned*>(orow + ((int64_t)W + x4) * 3);
        o0[0] = w0[0]; o0[1] = w0[1]; o0[2] = w0[2];
        o1[0] = w1[0]; o1[1] = w1[1]; o1[2] = w1[2];
    }
}

}  // namespace ide3d

extern "C" int ide3d_frame_u8(const float* img, const float* seg, const uint8_t* palette,
                              int32_t n, int32_t classes, int32_t H, int32_t W, uint8_t* out, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(img && seg && palette && out, "frame_u8: null pointer");
    IDE3D_CHECK_ARG(n > 0 && classes > 0 && H > 0 && W > 0, "frame_u8: bad shape");
    IDE3D_CHECK_ARG(W % 4 == 0, "frame_u8: W must be a multiple of 4");
    IDE3D_CHECK_ARG(((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(seg)) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 3) == 0, "frame_u8: misaligned tensor");
    const int64_t total = (int64_t)n * H * (W / 4);
    hipLaunchKernelGGL(frame_u8_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       img, seg, palette, n, classes, H, W, out);
    IDE3D_CHECK_LAUNCH("frame_u8");
    return IDE3D_OK;
}
