/*
 * ide3d_hip.h — C ABI of libide3d_hip.so, the MI355X (gfx950) implementation of the
 * IDE-3D rendering hot path.
 *
 * Every entry point is `extern "C"`, takes plain device pointers / sizes / strides and a
 * `void* stream` (a hipStream_t; NULL = the default stream), launches asynchronously on that
 * stream and returns 0 on success or a negative IDE3D_E* code.  After a failure
 * `ide3d_last_error()` returns a human readable, thread-local message.  No entry point
 * allocates, frees or synchronises; the caller owns every buffer.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the
 * IDE-3D repository, MrTornado24/IDE-3D).
 */
#ifndef IDE3D_HIP_H_
#define IDE3D_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- common --------------------------------------------------------------------------- */

#define IDE3D_OK            0
#define IDE3D_EINVAL       (-1)   /* bad argument (the reference raises via TORCH_CHECK)      */
#define IDE3D_ENOKERNEL    (-2)   /* no specialised kernel: caller must use its generic path  */
#define IDE3D_ELAUNCH      (-3)   /* hipLaunchKernel / runtime error                          */

/* element types of activation tensors */
#define IDE3D_F32  0
#define IDE3D_F16  1
#define IDE3D_BF16 2
#define IDE3D_F64  3

const char* ide3d_last_error(void);
/* ABI version (bumped on any signature change) and the gfx arch string the library was built for. */
int         ide3d_abi_version(void);
const char* ide3d_build_arch(void);
/* ABI 6.  Space-separated list of the non-default compile-time knobs this binary was built with ("" for a release build; `make EXTRA=-D...`,
 * scripts/micro/build_variant.sh).  A knob that changes RESULTS or drops a safety property — the timing-only experiments IDE3D_SP_DBG /
 * IDE3D_HEAD_DBG / IDE3D_F16_BF16MFMA (wrong values by design) and IDE3D_SP_SHARED_SIMD (no exclusive residency) — is listed
 * with a leading '!'; the host side (`hip_plugin.load()`) refuses such a library unless IDE3D_ALLOW_EXPERIMENT_BUILD=1 is set. */
const char* ide3d_build_flags(void);
/* ABI 6.  Kernels that contain an LDS-fed bf16 / fp16 matrix loop keep other waves off their SIMDs (one workgroup per CU: DESIGN.md 4.2).  The
 * library asks the HIP runtime, once per kernel and device, whether that holds where it runs (hipOccupancyMaxActiveBlocksPerMultiprocessor == 1);
 * a kernel for which it does not is never launched (the call returns IDE3D_ELAUNCH) and counted here.  0 on every supported configuration. */
int         ide3d_exclusive_violations(void);
const char* ide3d_exclusive_violation_text(void);

/* ---- bias_act ------------------------------------------------------------------------- */
/*
 * Replaces `_plugin.bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)`
 * (torch_utils/ops/bias_act.cpp:32, registered :94-97; kernel bias_act.cu:23).
 *   grad = 0: y = clamp(act(x + b) * gain)
 *   grad = 1: y = d/dx of the above, applied to incoming gradient x (=dy), using xref/yref
 *   grad = 2: second-order term (needs dy = the first-order incoming gradient)
 * x/xref/yref/dy/y are dense with identical memory layout, `size_x` elements of `dtype`.
 * b (may be NULL) has `size_b` elements of `dtype`; element i uses b[(i / step_b) % size_b]
 * (step_b = x.stride(dim), bias_act.cpp:80-82).  act is the reference's cuda_idx 1..9
 * (bias_act.py:22-30).  clamp < 0 disables clamping.
 */
int ide3d_bias_act(const void* x, const void* b, const void* xref, const void* yref,
                   const void* dy, void* y, int dtype, int grad, int act,
                   float alpha, float gain, float clamp,
                   int64_t size_x, int64_t size_b, int64_t step_b, void* stream);

/* ---- upfirdn2d ------------------------------------------------------------------------ */
/*
 * Replaces `_plugin.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1,
 * flip, gain)` (torch_utils/ops/upfirdn2d.cpp:16, registered :102-105; kernels
 * upfirdn2d.cu:29,97).  x is [n, c, in_h, in_w] with arbitrary element strides (NCHW or
 * channels_last), f is a float32 [f_h, f_w] tap array with element strides, y is
 * [n, c, out_h, out_w] with its own strides; out = (in*up + pad0 + pad1 - f + down) / down
 * (upfirdn2d.cpp:35-36) must match what the caller allocated.
 */
typedef struct ide3d_upfirdn2d_params {
    const void*  x;            /* input activations                                   */
    const float* f;            /* filter taps (float32, device)                       */
    void*        y;            /* output activations                                  */
    int32_t dtype;             /* IDE3D_F32 / F16 / BF16 / F64                         */
    int32_t n, c, in_h, in_w;  /* input shape                                         */
    int32_t out_h, out_w;      /* output spatial shape                                */
    int64_t x_stride[4];       /* element strides of x: n, c, h, w                    */
    int64_t y_stride[4];       /* element strides of y: n, c, h, w                    */
    int32_t f_h, f_w;          /* filter shape                                        */
    int64_t f_stride[2];       /* element strides of f: h, w                          */
    int32_t up_x, up_y, down_x, down_y;
    int32_t pad_x0, pad_y0;    /* leading pads (trailing pads are implied by out size) */
    int32_t flip;              /* 1 = correlation (flip_filter=True)                  */
    float   gain;
    int32_t x_row_floats;      /* ABI 5.  The caller's promise: this many elements may be READ from the start of EVERY row of x (padded row
                                  storage, e.g. the `y_pitch` output of ide3d_modconv2d mode 2).  0 = no promise beyond in_w.  The tile kernels
                                  stage rows with 16-byte loads only when round_up(in_w, 4) <= max(in_w, x_row_floats), and never touch an
                                  element at or beyond round_up(in_w, 4) of a row (values past in_w are discarded). */
} ide3d_upfirdn2d_params;

int ide3d_upfirdn2d(const ide3d_upfirdn2d_params* p, void* stream);

/*
 * upfirdn2d with a fused epilogue (MI355X extension; saves the HBM round trips of the element-wise ops that always
 * follow the FIR on the render path):
 *   y = bias_act( FIR(x) + add + noise * noise_strength )
 * `add` (same dtype as x, element strides) fuses the skip accumulation `img = upsample2d(img) + torgb(x)`
 * (inversion/networks.py:1100-1111); `noise` [out_h, out_w] float32 + `bias` [c] + act fuse the tail of an
 * up-sampling SynthesisLayer, `x.add_(noise)` and `bias_act(x, b, act='lrelu', gain, clamp)` (networks.py:507-512).
 * Any pointer may be NULL; fused_act = 0 skips the bias/activation stage.  act: 1 linear, 3 lrelu.
 */
#define IDE3D_AMAX_SLOTS  32       /* slots per image in the `amax` side outputs: workgroup b raises slot b % 32 */
#define IDE3D_AMAX_STRIDE 64       /* floats between two slots (256 bytes: every slot in its own cache line / L2 channel) */
#define IDE3D_AMAX_FLOATS (IDE3D_AMAX_SLOTS * IDE3D_AMAX_STRIDE)     /* floats per image; the value of slot k is element k * STRIDE */

typedef struct ide3d_upfirdn2d_epilogue {
    const void*  add;
    int64_t      add_stride[4];
    const float* noise;
    float        noise_strength;
    const void*  bias;
    int32_t      fused_act;
    int32_t      act;
    float        alpha, act_gain, clamp;
    float*       y_amax;      /* NULL, or [n][IDE3D_AMAX_FLOATS] float32 the caller zeroed: max over a row = max |y[n, :, :, :]| of the
                                 finished output, NaNs ignored (every workgroup raises slot `its index % IDE3D_AMAX_SLOTS` ONCE, after a read that
                                 usually makes the atomic unnecessary — 32 cache lines per image instead of one hot word;
                                 non-negative floats compare like unsigned integers).
                                 Feeds `x_amax` of the convolution that consumes y in the f16x3 arithmetic (ide3d_modconv_params). */
} ide3d_upfirdn2d_epilogue;

int ide3d_upfirdn2d_ex(const ide3d_upfirdn2d_params* p, const ide3d_upfirdn2d_epilogue* ep, void* stream);

/* ---- filtered_lrelu ------------------------------------------------------------------- */
/*
 * Replaces `_plugin.filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy,
 * gain, slope, clamp, flip_filters, writeSigns)` (torch_utils/ops/filtered_lrelu.cpp:16,
 * parameter block filtered_lrelu.h:14-53; kernel filtered_lrelu.cu:139).  One launch does
 * bias -> up-FIR(gain up^2) -> gain*lrelu(slope) -> clamp -> down-FIR.  Filters travel in the
 * kernel argument block / LDS (no process-global constant buffer: stream-safe, unlike
 * filtered_lrelu.cu:77-78).  fu/fd are float32; shape [h, w] with f*_h = 0 meaning a separable
 * 1-D filter of f*_w taps (filtered_lrelu.cpp:49-50).
 * Sign tensor s (uint8, 4 x 2-bit codes per byte; bit0 = negative, bit1 = clamped;
 * filtered_lrelu.cu:494-519) is [n, c, s_h, s_w_bytes] contiguous.  sign_mode: 0 none,
 * 1 write, 2 read.  Returns IDE3D_ENOKERNEL when the tile does not fit the LDS budget or the
 * filter exceeds the supported tap count (the reference's `return_code = -1`,
 * filtered_lrelu.cpp:52-56): the caller then runs the generic path with
 * ide3d_filtered_lrelu_act.
 */
typedef struct ide3d_filtered_lrelu_params {
    const void*  x;
    void*        y;
    const void*  b;            /* per-channel bias, dtype of x, never NULL            */
    uint8_t*     s;            /* sign tensor or NULL                                 */
    const float* fu;
    const float* fd;
    int32_t dtype;             /* IDE3D_F32 / F16 / BF16                               */
    int32_t n, c, in_h, in_w;
    int32_t out_h, out_w;
    int64_t x_stride[4];       /* element strides n, c, h, w                          */
    int64_t y_stride[4];
    int32_t fu_w, fu_h;        /* fu_h == 0 -> separable                              */
    int32_t fd_w, fd_h;
    int64_t fu_stride[2];      /* element strides: h, w (h unused when separable)     */
    int64_t fd_stride[2];
    int32_t up, down;
    int32_t pad_x0, pad_y0;
    int32_t s_w_bytes, s_h;    /* sign tensor shape (width in bytes)                  */
    int32_t s_ofs_x, s_ofs_y;  /* sign offset (elements), sx/sy of the reference      */
    int32_t sw_limit;          /* active sign width in bytes, filtered_lrelu.cpp:125  */
    int32_t sign_mode;         /* 0 none, 1 write, 2 read                             */
    int32_t flip;
    float   gain, slope, clamp;
} ide3d_filtered_lrelu_params;

int ide3d_filtered_lrelu(const ide3d_filtered_lrelu_params* p, void* stream);

/*
 * Replaces `_plugin.filtered_lrelu_act_(x, si, sx, sy, gain, slope, clamp, writeSigns)`
 * (torch_utils/ops/filtered_lrelu.cpp:213; kernel filtered_lrelu.cu:1105).  In-place
 * gain*lrelu*clamp on x [n, c, h, w] (element strides) with sign write / read.  s_w is the
 * sign width in *elements* (a multiple of 16 when writing), s is contiguous.
 */
int ide3d_filtered_lrelu_act(void* x, uint8_t* s, int dtype,
                             int32_t n, int32_t c, int32_t h, int32_t w,
                             const int64_t x_stride[4],
                             int32_t s_w, int32_t s_h, int32_t s_ofs_x, int32_t s_ofs_y,
                             float gain, float slope, float clamp, int sign_mode, void* stream);

/* ---- tri-plane feature gather ----------------------------------------------------------- */
/*
 * Replaces `dnnlib.util.sample_from_triplane(coordinates, grid)` (dnnlib/util.py:580-617:
 * three `grid_sample(bilinear, zeros, align_corners=False)` calls — torch_utils/ops/
 * grid_sample_gradfix.py:26-29 — on the xy, yz and xz planes, summed).
 * planes: float32 [n, 3*C, H, W] addressed through element strides (NCHW or channels_last;
 *         channels_last is the fast path: each bilinear tap is one contiguous C*4-byte read).
 * coords: float32 [n, m, 3] contiguous, world coordinates in grid_sample's [-1, 1] convention.
 * out:    float32 [n*m, C] contiguous, row = n*m_total + m.
 * Tap index math is bit-exact w.r.t. ATen grid_sampler_2d: u = ((c + 1) * size - 1) / 2
 * evaluated add -> mul -> sub -> mul(0.5) in fp32 with no FMA contraction.
 */
int ide3d_triplane_sample(const float* planes, const int64_t plane_stride[4],
                          int32_t n, int32_t C, int32_t H, int32_t W,
                          const float* coords, int64_t m, float* out, void* stream);

/*
 * Same operation and same results, for samples that come from a ray grid (what the ray-marcher
 * of volumetric_rendering.py:160-191 produces): coords is [n, rays_h, rays_w, steps, 3], i.e.
 * m == rays_h * rays_w * steps with the depth step innermost.  The shape is a grouping hint
 * only: 8x8 ray tiles x 4 depth steps stage the plane texels they share in LDS once instead of
 * fetching them per sample.  Any coordinates are legal; shapes / layouts the tiled kernel does
 * not cover run `ide3d_triplane_sample`.
 */
int ide3d_triplane_sample_rays(const float* planes, const int64_t plane_stride[4],
                               int32_t n, int32_t C, int32_t H, int32_t W,
                               const float* coords, int64_t m, float* out,
                               int32_t rays_h, int32_t rays_w, int32_t steps, void* stream);

/*
 * Debug / parity hook: writes, per sample and plane, the integer tap origin (floor(u),
 * floor(v)) and the in-bounds mask of the four taps, so tests can assert bit-exact index
 * math against the oracle.  taps: int32 [n*m, 3, 3] = (ix0, iy0, mask4).
 */
int ide3d_triplane_taps(int32_t H, int32_t W, const float* coords, int64_t n_times_m,
                        int32_t* taps, void* stream);

/*
 * Backward of the gather w.r.t. the planes (atomic scatter-add into grad_planes, which the
 * caller zero-fills) and optionally w.r.t. the coordinates (grad_coords may be NULL).
 * Replaces aten::grid_sampler_2d_backward as used by grid_sample_gradfix.py:55-60.
 */
int ide3d_triplane_sample_backward(const float* grad_out, const float* planes,
                                   const int64_t plane_stride[4],
                                   int32_t n, int32_t C, int32_t H, int32_t W,
                                   const float* coords, int64_t m,
                                   float* grad_planes, const int64_t grad_plane_stride[4],
                                   float* grad_coords, void* stream);

/* ---- volume compositing ----------------------------------------------------------------- */
/*
 * Replaces `training.volumetric_rendering.fancy_integration` (volumetric_rendering.py:34-74).
 * rgb_sigma: float32 [rays, steps, ch+1] contiguous (sigma last), z_vals: [rays, steps],
 * dir_norm: [rays] (= ||rays_d_cam||), noise: [rays, steps] or NULL (already scaled by
 * noise_std).  Outputs: rgb [rays, ch], depth [rays], weights [rays, steps] (NULL = skip).
 * clamp_mode: 0 softplus, 1 relu.  fill_mode: 0 none, 1 'debug', 2 'weight'.
 * One wavefront integrates one ray: lane-local alpha, wave-wide exclusive prefix product of
 * (1 - alpha + 1e-10) by shuffles, then a shuffle reduction per channel.
 */
int ide3d_composite(const float* rgb_sigma, const float* z_vals, const float* dir_norm,
                    const float* noise, int64_t rays, int32_t steps, int32_t ch,
                    int clamp_mode, int last_back, int white_back, float max_depth,
                    int fill_mode, float* rgb, float* depth, float* weights, void* stream);

/* ---- importance resampling ------------------------------------------------------------- */
/*
 * Replaces `training.volumetric_rendering.sample_pdf` (volumetric_rendering.py:224-265), the
 * inverse-CDF draw of the optional hierarchical pass.  bins: float32 [rays, k+1], weights:
 * [rays, k] (both contiguous), u: the draws in [0, 1] — row `r` starts at u + r * u_ray_stride
 * (0 = one row shared by all rays, the `det=True` linspace), samples: [rays, n_importance].
 * Row sums / prefix sums are accumulated in double and rounded per element like ATen's CPU
 * kernels; everything after the cdf is the reference's float expression without contraction.
 */
int ide3d_sample_pdf(const float* bins, const float* weights, const float* u, int64_t u_ray_stride,
                     int64_t rays, int32_t k, int32_t n_importance, float eps, float* samples,
                     void* stream);

/* ---- fused ray-marcher ------------------------------------------------------------------ */
/*
 * One launch for steps 3-7 of G.synthesis: camera-space sample points
 * (get_initial_rays_trig, volumetric_rendering.py:77-97) -> optional stratified jitter
 * (perturb_points :99-105) -> cam2world transform (transform_sampled_points :108-136) ->
 * two tri-plane gathers (dnnlib/util.py:580) -> decoder MLPs (renderer.sample_voxel, source
 * absent: spec in DESIGN.md) -> fancy_integration (:34-74).
 *
 * rays_d_cam [rays_per_img, 3] and z_lin [steps] are produced on the host with torch.linspace
 * exactly as the reference does, so the kernel never re-derives them.
 * cam2world: [n, 16] row-major 4x4.  jitter: [n, rays_per_img, steps] uniform [0,1) or NULL.
 * tex_planes / geo_planes: [n, 3*C, H, W] through element strides.
 * MLP weights (float32, row-major [out, in], already multiplied by their runtime gains):
 *   geo: w0 [hidden, C], b0 [hidden], w1 [1 + seg_ch, hidden], b1 [1 + seg_ch]
 *   tex: w0 [hidden, C], b0 [hidden], w1 [feat_ch, hidden],    b1 [feat_ch]
 * out_feat: [n, feat_ch + seg_ch, rays_per_img] (channel-major image planes),
 * out_depth: [n, rays_per_img], out_wsum: [n, rays_per_img] (either may be NULL).
 */
typedef struct ide3d_render_params {
    const float* rays_d_cam;
    const float* z_lin;
    const float* cam2world;
    const float* jitter;
    const float* sigma_noise;   /* [n, rays, steps] pre-scaled density noise or NULL */
    const float* tex_planes;
    const float* geo_planes;
    int64_t tex_stride[4];
    int64_t geo_stride[4];
    const float* geo_w0; const float* geo_b0; const float* geo_w1; const float* geo_b1;
    const float* tex_w0; const float* tex_b0; const float* tex_w1; const float* tex_b1;
    int32_t n, rays_per_img, steps;
    int32_t C, H, W;
    int32_t hidden, feat_ch, seg_ch;
    int32_t clamp_mode, last_back, white_back;
    float   max_depth;
    float*  out_feat;
    float*  out_depth;
    float*  out_wsum;
} ide3d_render_params;

int ide3d_render_rays(const ide3d_render_params* p, void* stream);

/*
 * `renderer.sample_voxel(img_v, seg_v, pts)` (call site extract_shapes.py:146): the same two
 * gathers + MLPs for arbitrary points, no compositing.  out: [n*m, feat_ch + seg_ch + 1]
 * (sigma last).  If sigma_only != 0 only out_sigma [n*m] is written (the 256^3 density-cube
 * query) and the texture branch is skipped.
 */
int ide3d_sample_voxel(const ide3d_render_params* p, const float* pts, int64_t m,
                       float* out, float* out_sigma, int sigma_only, void* stream);

/*
 * The point lattice of `extract_shapes.create_samples` (extract_shapes.py:74-96) with the
 * 0.9 scale of extract_shapes.py:103 folded in: point i (0 <= i < n^3) is
 *   s2 = i % n,  s1 = (float(i) / n) % n,  s0 = ((float(i) / n) / n) % n     (fp32, NOT floored: reference quirk)
 *   (x, y, z) = ((s0 * voxel_size + corner[2]) * scale, (s1 * voxel_size + corner[1]) * scale,
 *                (s2 * voxel_size + corner[0]) * scale)
 * with every operation rounded to fp32 separately, i.e. bit-equal to the reference's host code.
 * corner = voxel_origin - cube_length / 2, voxel_size = cube_length / (n - 1), both rounded to fp32.
 */
typedef struct ide3d_lattice {
    int32_t n;
    float   voxel_size;
    float   corner[3];
    float   scale;
} ide3d_lattice;

/* Writes points [first, first + count) of the lattice to pts [count, 3] (device). */
int ide3d_lattice_points(const ide3d_lattice* lat, int64_t first, int64_t count, float* pts, void* stream);

/*
 * The chunk loop body of extract_shapes.py:144-148 without materialising coordinates:
 * `renderer.sample_voxel(img_v, seg_v, samples[:, first:first+count])[..., -1]` for every image
 * of p (p->n), lattice points generated in registers.  out_sigma: [p->n * count].
 * (p's ray / compositing fields are ignored, as in ide3d_sample_voxel.)
 */
int ide3d_density_lattice(const ide3d_render_params* p, const ide3d_lattice* lat, int64_t first, int64_t count,
                          float* out_sigma, void* stream);

/* ---- modulated 3x3 / 1x1 convolution (MFMA implicit GEMM) --------------------------------- */
/*
 * Replaces the ATen conv behind `conv2d_gradfix.conv2d` for the StyleGAN2 modulated
 * convolution (inversion/networks.py:55-130 `modulated_conv2d`, stride 1, `flip_weight=True`)
 * with its epilogue (`+ noise`, `bias_act(lrelu)`, networks.py:457-512) fused:
 *   y[n,o,:,:] = act( d[n,o] * sum_{i,ky,kx} w[o,i,ky,kx] * s[n,i] * x[n,i,y+ky-p,x+kx-p]
 *                     + noise_strength * noise[:, :] + b[o] ) * act_gain, clamped.
 * x [n, cin, h, w], y [n, cout, h, w] NCHW contiguous float32; w [cout, cin, k, k];
 * styles s [n, cin]; dcoefs d [n, cout] or NULL (no demodulation); noise [h, w] or NULL;
 * bias [cout] or NULL.  act: 1 linear, 3 lrelu (bias_act cuda_idx).  k in {1, 3}.
 * fp32 in / fp32 out / fp32 accumulate.  Shared-weight 1x1, stride-2, narrow (< 64 output channels, <= 32 input channels) and small
 * (< 12 pixels) layers run on v_mfma_f32_32x32x2_f32 (exact fp32 products); the shared-weight 3x3 and transposed 3x3 layers run in the arithmetic
 * selected by `arith` / ide3d_set_conv_arithmetic() below.
 */
typedef struct ide3d_modconv_params {
    const float* x; const float* w; const float* styles; const float* dcoefs;
    const float* noise; const float* bias; float* y;
    int32_t n, cin, cout, h, w_, k;
    float noise_strength;
    int32_t act; float alpha, gain, clamp;
    int32_t mode;             /* 0: stride-1 k x k "same" correlation (y is h x w);
                                 1: 3x3 correlation, stride 2, no padding (y is ((h-3)/2+1) x ((w-3)/2+1)) — the conv that
                                    follows the low-pass filter of a down-sampling Conv2dLayer (conv2d_resample.py:100-103;
                                    styles / dcoefs may be NULL: plain convolution, inversion/networks.py:169-226);
                                 2: 3x3 transposed convolution, stride 2, pad 0 (y is (2h+1) x (2w+1)) —
                                    `conv_transpose2d(x, w.transpose(0,1), stride=2)` of conv2d_resample.py:114-125 */
    int32_t weights_packed;   /* 1: `workspace` already holds the packed form of `w` from an earlier call */
    float*  workspace;        /* scratch of ide3d_modconv_workspace_bytes(): packed weights, then split-K partials */
    int64_t workspace_bytes;
    int64_t w_batch_stride;   /* 0: one weight tensor for the batch (modulation via `styles` on the input);
                                 > 0: per-image weights w + n * w_batch_stride (styles already folded in by the caller:
                                 lets heads with different styles — toRGB + toSeg, networks.py:1109,1130 — share one launch) */
    int32_t arith;            /* arithmetic of the shared-weight 3x3 layers: 0 = process default (ide3d_set_conv_arithmetic),
                                 1 = fp32 MFMA, 3 = bf16x3, 6 = bf16x6, 16 = f16x3 (see ide3d_set_conv_arithmetic) */
    const float* x_amax;      /* NULL, or [n][IDE3D_AMAX_FLOATS]: the row maximum is an upper bound of max |x[n, :, :, :]| (what `y_amax` of the
                                 producing launch holds).  The f16x3 arithmetic needs it to place x * styles inside the fp16 range;
                                 without it a launch that asked for f16x3 runs in bf16x6. */
    float*       y_amax;      /* NULL, or [n][IDE3D_AMAX_FLOATS] float32 zeroed by the caller: row maximum = max |y[n, :, :, :]| (NaNs ignored; an inf makes
                                 the consumer compute that image without range scaling) */
    int32_t      y_pitch;     /* mode 2 only: 0 = dense y [n, cout, 2h+1, 2w+1]; else the row pitch in floats (>= 2w+1) of a y whose rows are padded —
                                 [n, cout, 2h+1, y_pitch] storage, columns >= 2w+1 never written — so that rows start 16-byte aligned for the FIR
                                 that consumes it (ide3d_upfirdn2d_ex stages aligned rows with 16-byte loads) */
} ide3d_modconv_params;

int64_t ide3d_modconv_workspace_bytes(int32_t n, int32_t cin, int32_t cout, int32_t h, int32_t w, int32_t k, int32_t mode,
                                      int32_t per_image_weights);
int ide3d_modconv2d(const ide3d_modconv_params* p, void* stream);

/* Host-only planning query (no launch, no device access): the kernel family, tile and grid that ide3d_modconv2d would use for `p` in the
 * arithmetic `p->arith` resolves to.  Pointers are not dereferenced (set `x_amax` non-null to plan the f16x3 launch; `x` counts for its
 * alignment only).  For tests of the planner and for tooling; not part of the reference's interface. */
enum { IDE3D_PLAN_FP32 = 0,          /* fp32-MFMA matrix loop (modconv_kernel) */
       IDE3D_PLAN_SPLIT = 1,         /* split-bf16 / fp16 matrix loop (modconv_split_kernel) */
       IDE3D_PLAN_SPLIT_TEAMS = 2,   /* the same, two 4-wave teams per workgroup */
       IDE3D_PLAN_HEAD_SPLIT = 3,    /* per-image 1x1 heads, one tile per workgroup (head_split_kernel) */
       IDE3D_PLAN_HEAD_RESIDENT = 4, /* per-image 1x1 heads, packed weights resident in LDS (head_resident_kernel) */
       IDE3D_PLAN_HEAD_SMALL = 5 };  /* per-image 1x1 heads on maps of <= 256 pixels, fp32 FMAs (head_small_kernel) */
typedef struct ide3d_modconv_plan_info {
    int32_t kind;                    /* IDE3D_PLAN_* */
    int32_t tile_h, tile_w;          /* pixels (transposed: grid positions) per workgroup / team */
    int32_t images_per_tile;
    int32_t rows;                    /* output channels per workgroup / team */
    int32_t waves;                   /* per workgroup */
    int32_t parts, f16;              /* pieces per operand (0: fp32 products), fp16 pieces */
    int32_t split_k;                 /* > 1: partial sums + reduction launch */
    int32_t strip;                   /* transposed: main kernel on the h x w grid + strip kernel */
    int32_t transposed_all_class;
    int32_t reserved;
    int64_t workgroups;              /* of the main launch */
} ide3d_modconv_plan_info;
int ide3d_modconv_plan(const ide3d_modconv_params* p, ide3d_modconv_plan_info* out);

/*
 * Arithmetic of the shared-weight 3x3 / transposed 3x3 layers on 16-pixel-wide tiles (everything else always runs on the
 * fp32 MFMA).  The reference computes these layers with ATen's fp32 convolution (conv2d_gradfix.py:35,40), or in fp16 for
 * the `num_fp16_res` highest resolutions of a released pickle (inversion/networks.py:1058-1060):
 *   1  fp32 MFMA: exact fp32 products, fp32 accumulation (v_mfma_f32_32x32x2_f32);
 *   6  bf16x6: each fp32 operand = 3 bf16 pieces, the 6 products above 2^-24 on v_mfma_f32_32x32x16_bf16, fp32 accumulation:
 *      fp32-grade (per-product error <= ~2^-23 relative) at 6/16 of the fp32 MFMA time;
 *   3  bf16x3: 2 pieces, 3 products, per-product error ~2^-17 relative, 3/16 of the time;
 *  16  f16x3: each operand = 2 fp16 pieces (hi = fp16(a), lo = fp16(a - hi): 22 significand bits), the 3 products hi*hi + hi*lo +
 *      lo*hi on v_mfma_f32_32x32x16_f16, fp32 accumulation: per-product error <= ~2^-21 relative to |w|max(row) * |x s|max(image),
 *      3/16 of the fp32 MFMA time.  fp16's range is handled with exact power-of-two scales: one per weight row (from max |w[o]|,
 *      applied when the weights are packed) and one per image (from `x_amax[n]` * max_i |styles[n, i]|, applied while the patch is
 *      staged), both undone together with the demodulation in the epilogue.  Operands more than 2^17 below their row / image
 *      maximum lose relative (not absolute) precision: an fp16 piece cannot be smaller than 2^-24 of the scaled range.  Needs
 *      `x_amax`; without it the launch runs in bf16x6;
 *   0  back to the process default: environment IDE3D_CONV_ARITH = fp32 | bf16x6 | bf16x3 | f16x3, else bf16x6 (round 4; fp32 before).
 *      A wave of ANOTHER kernel that executes packed fp32 VALU instructions (v_pk_fma_f32 ...) on operands fresh from global memory
 *      returns wrong results on MI355X while a wave on the same SIMD runs a loop of LDS reads + bf16 / fp16 MFMAs.
 *      Every kernel of this library that contains such a loop keeps foreign waves off its SIMDs for as long as the loop runs (8-wave
 *      workgroups whose waves hold 256 registers each, or waves that claim all 512 registers of their SIMD: "exclusive residency"), so
 *      every arithmetic may run beside kernels of other libraries on other streams.
 * The same switch selects the arithmetic of per-image 1x1 heads with <= 32 or 161..192 outputs on grids of >= 512 workgroups, and of the
 * two decoder MLPs inside ide3d_render_rays / ide3d_sample_voxel / ide3d_density_lattice at C = 32, hidden = 64: exact fp32 products
 * on v_mfma_f32_16x16x4_f32 with fp32 (1) selected, bf16x6 (fp32-grade, 3 bf16 pieces per operand on v_mfma_f32_16x16x32_bf16)
 * with any of the split arithmetics (3, 6, 16).
 * Packed weights in a modconv workspace are specific to the arithmetic they were packed for.
 */
int     ide3d_set_conv_arithmetic(int32_t arith);
int32_t ide3d_get_conv_arithmetic(void);

/* ---- per-layer style preparation ------------------------------------------------------------------ */
/*
 * One launch for what precedes a modulated convolution in inference (inversion/networks.py:432, :91-93):
 *   styles[n, :] = (w[n, :] @ affine_w^T) * affine_gain + affine_b * bias_gain          FullyConnectedLayer, :152-165
 *   dcoefs[n, o] = rsqrt( sum_i styles[n, i]^2 * wsq_t[i, o] + 1e-8 ),  wsq_t[i, o] = sum_k W[o, i, k]^2
 * w: rows of `wdim` floats, `w_stride` apart; affine_w [cin, wdim]; affine_b [cin] or NULL; wsq_t [cin, cout];
 * dcoefs may be NULL (no demodulation).
 */
int ide3d_style_demod(const float* w, int64_t w_stride, const float* affine_w, const float* affine_b, const float* wsq_t,
                      int32_t n, int32_t cin, int32_t cout, int32_t wdim, float affine_gain, float bias_gain,
                      float* styles, float* dcoefs, void* stream);

/*
 * Per-image folded weights of the two heads of a dual-path block (toRGB + toSeg share w, networks.py:1093-1130):
 *   out[n, o, i] = W_h[o, i] * (affine_h(w[n]) [i] * gain_h),  h = 0 for o < cout0 else 1;  out [n, cout0 + cout1, cin].
 * Feeds ide3d_modconv2d with w_batch_stride > 0 so that both 1x1 heads are one launch.
 */
int ide3d_fold_heads(const float* w, int64_t w_stride, int32_t n, int32_t cin, int32_t wdim, float affine_gain,
                     const float* a0, const float* b0, const float* w0, int32_t cout0, float gain0,
                     const float* a1, const float* b1, const float* w1, int32_t cout1, float gain1,
                     float* out, void* stream);

/*
 * The two entry points above for ALL layers of a synthesis pass at once: every modulated layer's affine + demodulation in two
 * launches, every block's head folding in one (round 3: as 43 separate nodes of a captured graph they delay the first convolution
 * of the pass by ~0.4 ms).  A job carries exactly the per-layer arguments of ide3d_style_demod / ide3d_fold_heads; n (images) and
 * wdim are shared by the jobs.  Same code per block as the per-layer kernels: bit-identical results.  njobs <= IDE3D_STYLE_BATCH_MAX;
 * ide3d_style_demod_batch: n <= 8.
 */
#define IDE3D_STYLE_BATCH_MAX 24
typedef struct ide3d_style_job {
    const float* w; int64_t w_stride;                 /* this layer's latent of image 0, stride to the next image */
    const float* affine_w; const float* affine_b;     /* [cin, wdim], [cin] or NULL */
    const float* wsq_t;                               /* [cin, cout] or NULL (no demodulation) */
    int32_t cin, cout;
    float affine_gain, bias_gain;
    float* styles; float* dcoefs;                     /* [n, cin]; [n, cout] or NULL */
} ide3d_style_job;
typedef struct ide3d_fold_job {
    const float* w; int64_t w_stride;
    int32_t cin; float affine_gain;
    const float *a0, *b0, *w0; int32_t cout0; float gain0;
    const float *a1, *b1, *w1; int32_t cout1; float gain1;
    float* out;                                       /* [n, cout0 + cout1, cin] */
} ide3d_fold_job;
int ide3d_style_demod_batch(const ide3d_style_job* jobs, int32_t njobs, int32_t n, int32_t wdim, void* stream);
int ide3d_fold_heads_batch(const ide3d_fold_job* jobs, int32_t njobs, int32_t n, int32_t wdim, void* stream);

/* ---- the low-resolution block group of the backbone in one launch (round 6, ABI 7) ------------------- */
/*
 * The first blocks of the dual-path StyleGAN2 backbone (inversion/networks.py:966-1139: 4^2, 8^2, 16^2 ... while all images' maps of a layer
 * fit one CU's LDS beside a weight slice) — every 3x3 / up-sampling SynthesisLayer (networks.py:330-514) with its noise, bias, lrelu,
 * gain and clamp, the 4x4 FIR of the up-sampling layers (conv2d_resample.py:112-129), the toRGB + toSeg heads (networks.py:670-713) and the
 * skip accumulation `img = upsample2d(img) + y` (networks.py:1100,1121) — as ONE launch (persistent != 0: phases separated by a grid barrier)
 * or one launch per phase (persistent == 0).  All layers have C input and C output channels (C % 32 == 0); fp32 in / fp32 out; products in
 * the bf16x6 (or bf16x3) arithmetic of ide3d_set_conv_arithmetic, heads in plain fp32 FMAs.  Inference only.
 *   layer l: up = 1: y = lrelu(d * conv3x3(x * s) + noise + b, 0.2) * act_gain, clamped;   up = 2: the same with
 *            conv_transpose2d(stride 2) -> upfirdn2d(f, pad 1, gain 4) in place of the convolution (output 2 x the input resolution);
 *            head >= 0: heads[head] is applied to this layer's output: skip = clamp(W[n] y + b) + upsample2d(heads[head - 1].skip, f).
 *   x0: the group's input [C, res0, res0] (x0_batch_stride == 0: the learned constant, shared by the batch) or [n, C, res0, res0].
 *   x_out: the last layer's output [n, C, res, res] NCHW.  heads[k].skip [n, O, res_k, res_k] NCHW (every head's skip image is written).
 * Returns IDE3D_ENOKERNEL when the arithmetic is fp32 / f16x3 or the layers do not fit (ask ide3d_lowres_layers_supported): callers then
 * run the layers one by one (ide3d_modconv2d ...).
 */
#define IDE3D_LOWRES_MAX_LAYERS 8
#define IDE3D_LOWRES_MAX_HEADS 4
typedef struct ide3d_lowres_layer {
    const float* weight;      /* [C, C, 3, 3] */
    const float* styles;      /* [n, C] */
    const float* dcoefs;      /* [n, C] */
    const float* noise;       /* [res, res], already multiplied by noise_strength, or NULL */
    const float* bias;        /* [C] or NULL */
    float act_gain, clamp;    /* clamp < 0: none */
    int32_t up;               /* 1 or 2 */
    int32_t head;             /* index into heads[], or -1 */
    int32_t weights_packed;   /* 1: the workspace already holds this layer's packed weights (same weight version, same arithmetic) */
    int32_t reserved;
} ide3d_lowres_layer;
typedef struct ide3d_lowres_head {
    const float* w;           /* [n, O, C] per-image folded weights (ide3d_fold_heads) */
    const float* bias;        /* [O] or NULL */
    float* skip;              /* out [n, O, res, res] */
    float clamp;
    int32_t O;
} ide3d_lowres_head;
typedef struct ide3d_lowres_params {
    const float* x0; int64_t x0_batch_stride;
    const float* fir;         /* [4, 4] resample filter (upfirdn2d.setup_filter([1, 3, 3, 1])) */
    float* x_out;
    void* workspace; int64_t workspace_bytes;
    int32_t n, C, res0, nlayers, nheads;
    int32_t arith;            /* 0 = process default; only 6 (bf16x6) and 3 (bf16x3) have this form */
    int32_t persistent;
    int32_t reserved;
    ide3d_lowres_layer layers[IDE3D_LOWRES_MAX_LAYERS];
    ide3d_lowres_head heads[IDE3D_LOWRES_MAX_HEADS];
} ide3d_lowres_params;
/* number of leading layers (ups[l] in {1, 2}) of a group that starts at res0 which ide3d_lowres_group accepts for this batch size */
int32_t ide3d_lowres_layers_supported(int32_t n, int32_t C, int32_t res0, const int32_t* ups, int32_t nlayers, int32_t arith);
int64_t ide3d_lowres_workspace_bytes(const ide3d_lowres_params* p);     /* only n, C, res0, nlayers, arith and layers[].up are read */
int ide3d_lowres_group(const ide3d_lowres_params* p, void* stream);

/* ---- output post-processing -------------------------------------------- */
/* ---- mapping network ---------------------------------------------------------------------- */
/*
 * `MappingNetwork.forward` (inversion/networks.py:287-325) in one launch: normalize_2nd_moment(z), embed(c) + normalize,
 * concat, `layers` x FullyConnectedLayer(lrelu, lr_multiplier), broadcast to num_ws, truncation towards w_avg.
 * z [n, z_dim], c [n, c_dim], embed_w [embed, c_dim], fc_w[l] [fc_out[l], in_l] (in_0 = z_dim + embed, in_l = fc_out[l-1]):
 * dense float32 with the RAW parameters; the runtime gains are applied here like FullyConnectedLayer does
 * (weight_gain = lr_multiplier / sqrt(in), bias_gain = lr_multiplier; the embed layer's gains are passed explicitly).
 * ws [n, num_ws, fc_out[layers-1]] is written.  truncation_psi == 1 disables truncation; truncation_cutoff < 0 = all layers.
 * workspace: ide3d_mapping_workspace_bytes() bytes of device memory (one activation buffer per layer + barrier counter + error word).
 * n <= 8, widths <= 1024 and multiples of 4; returns IDE3D_EINVAL otherwise (callers then use their generic path).
 */
typedef struct ide3d_mapping_params {
    const float* z; const float* c;
    const float* embed_w; const float* embed_b;
    const float* fc_w[16]; const float* fc_b[16];
    int32_t fc_out[16];
    const float* w_avg;
    float* ws;
    void* workspace; int64_t workspace_bytes;
    int32_t n, z_dim, c_dim, embed, layers, num_ws;
    float embed_weight_gain, embed_bias_gain, lr_multiplier, alpha, act_gain, truncation_psi;
    int32_t truncation_cutoff;
} ide3d_mapping_params;

int ide3d_mapping_workspace_bytes(void);
/* Always 1 since round 6 (ABI 7): where the one-launch kernel's 64 workgroups are not co-resident on the current device (its grid-wide barrier needs that;
 * checked against the occupancy query with a 2x margin) `ide3d_mapping` runs the same layers as one launch each. */
int ide3d_mapping_supported(void);
int ide3d_mapping(const ide3d_mapping_params* p, void* stream);

/* ---- resampling between the big kernels of G.synthesis ---------------------------------- */
/*
 * out = upsample2d(lo, [1,3,3,1]) + add, written CHANNELS-LAST: the last skip accumulation of the tri-plane backbone
 * (`img = upfirdn2d.upsample2d(img, resample_filter) + torgb(x)`, inversion/networks.py:1100-1111; upsample2d =
 * upfirdn2d(up=2, pad (2,1,2,1), gain 4), torch_utils/ops/upfirdn2d.py:313-349) producing the layout the ray-marcher gathers
 * from (32-channel texels = 128-byte lines) instead of NCHW + two transposing copies.
 * lo [n, c, h, w] and add [n, c, 2h, 2w]: float32, element strides; out: float32 [n, 2h, 2w, c] dense (= a torch
 * channels_last [n, c, 2h, 2w]), 16-byte aligned, c % 4 == 0.
 */
int ide3d_skip_upsample_add_cl(const float* lo, const int64_t lo_stride[4], const float* add, const int64_t add_stride[4],
                               int32_t n, int32_t c, int32_t h, int32_t w, float* out, void* stream);

/*
 * torch.nn.functional.interpolate(x, scale 2, mode='bilinear', align_corners=False) of x [n, c, h, w] (dense NCHW float32),
 * split into up to three dense NCHW outputs: dst[k] [n, c_count[k], 2h, 2w] takes input channels c_begin[k] ...
 * (the composited 64x64 feature image -> colour features / raw RGB / semantic logits of the super-resolution blocks,
 * SURVEY.md Appendix B; ATen source-index rule `area_pixel_compute_source_index`).  c_count[k] == 0 skips output k.
 * ABI 8: dst_batch_floats (may be NULL = all dense) — floats between consecutive images of output k, 0 = dense (c_count[k] * 2h * 2w): outputs that
 * are channel ranges of ONE tensor (raw RGB | semantic logits, which the skip up-sampler of the next block then reads in one launch).
 */
int ide3d_bilinear_up2_split(const float* x, int32_t n, int32_t c, int32_t h, int32_t w,
                             float* const dst[3], const int32_t c_begin[3], const int32_t c_count[3], const int64_t* dst_batch_floats, void* stream);

/*
 * `mask2color(seg)` (dnnlib/seg_tools.py:75-81: argmax over the class channel + palette) and
 * the uint8 conversion of `layout_grid` (dnnlib/util.py:632-646), fused: writes one
 * [n, H, 2W, 3] uint8 frame per image = RGB | coloured segmentation side by side, the
 * `image_seg` layout of gen_videos.py:133-135.
 * img [n, 3, H, W], seg [n, classes, H, W] float32 NCHW; palette uint8 [classes, 3].
 */
int ide3d_frame_u8(const float* img, const float* seg, const uint8_t* palette,
                   int32_t n, int32_t classes, int32_t H, int32_t W, uint8_t* out, void* stream);

/*
 * ABI 8.  The camera-pose helpers of training/volumetric_rendering.py as two launches (the reference spells them as ~45 one-element
 * tensor operations per pose: ~0.3 ms of host time per image in gen_images.py:104-106 / gen_videos.py:120-124).  One thread per camera;
 * every operation rounds on its own, in the reference's order.
 *
 * ide3d_sphere_points — the tail of `sample_camera_positions` (:186-193) and the middle of `LookAtPoseSampler.sample` (:281-291):
 *   theta [n], pitch [n] float32 (the caller has drawn them: the random modes keep torch's generator);
 *   pitch_is_v == 0: phi = clamp(pitch, 1e-5, pi - 1e-5);   pitch_is_v == 1: phi = arccos(1 - 2 * (clamp(pitch, ...) / pi));
 *   pos [n, 3] = r * (sin phi cos theta, cos phi, sin phi sin theta);  phi_out [n] (may be NULL) = phi.
 * ide3d_cam2world — `create_cam2world_matrix(forward_vector, origin)` (:195-213): out [n, 4, 4] = T(origin) @ R(columns -left, up, -forward)
 *   with forward normalised, left = normalise(cross((0, 1, 0), forward)), up = normalise(cross(forward, left)).
 *   lookat != NULL (LookAtPoseSampler :294-295): forward = normalise(lookat - origin) first (`forward` is ignored, may be NULL);
 *   lookat_stride = 3 (one point per camera) or 0 (one point for all).
 */
int ide3d_sphere_points(const float* theta, const float* pitch, int32_t n, float r, int32_t pitch_is_v,
                        float* pos, float* phi_out, void* stream);
int ide3d_cam2world(const float* forward, const float* origin, const float* lookat, int32_t lookat_stride, int32_t n,
                    float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IDE3D_HIP_H_ */
