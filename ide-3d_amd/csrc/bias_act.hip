// bias_act.hip — fused bias + activation + gain + clamp (and its 1st / 2nd order gradient forms).
//
// Behaviour follows the reference op `torch_utils/ops/bias_act.py:52` / kernel `bias_act.cu:23`
// (formulas per activation and gradient order), re-designed for CDNA4: a pure HBM-streaming
// kernel that moves 16 bytes per lane per access (global_load_dwordx4), grid-strides over
// 256 CUs x 8 workgroups, and resolves the bias index once per 16-byte vector whenever the
// bias stride allows it.  Algorithmic traffic: 2 * numel * sizeof(T) (+ xref/yref/dy reads for
// the gradient forms).
#include "common.h"
#include "knobs.h"

namespace ide3d {

template <class T, int N> struct alignas(sizeof(T) * N) Vec { T v[N]; };

template <class M>
__device__ __forceinline__ M act_eval(int A, int G, M x, M xref, M yy, M alpha, M& yref, M gain) {
    const M one = (M)1, two = (M)2;
    const M expRange = (M)80, halfExpRange = (M)40;
    const M seluScale = (M)1.0507009873554804934193349852946;
    const M seluAlpha = (M)1.6732632423543772848170429916717;
    M y = 0;
    switch (A) {
    case 1:  // linear
        if (G <= 1) y = x;
        break;
    case 2:  // relu
        if (G == 0) y = (x > 0) ? x : (M)0;
        if (G == 1) y = (yy > 0) ? x : (M)0;
        break;
    case 3:  // lrelu
        if (G == 0) y = (x > 0) ? x : x * alpha;
        if (G == 1) y = (yy > 0) ? x : x * alpha;
        break;
    case 4:  // tanh
        if (G == 0) y = tanh(x);
        if (G == 1) y = x * (one - yy * yy);
        if (G == 2) y = x * (one - yy * yy) * (-two * yy);
        break;
    case 5:  // sigmoid
        if (G == 0) y = (x < -expRange) ? (M)0 : one / (exp(-x) + one);
        if (G == 1) y = x * yy * (one - yy);
        if (G == 2) y = x * yy * (one - yy) * (one - two * yy);
        break;
    case 6:  // elu
        if (G == 0) y = (x >= 0) ? x : expm1(x);
        if (G == 1) y = (yy >= 0) ? x : x * (yy + one);
        if (G == 2) y = (yy >= 0) ? (M)0 : x * (yy + one);
        break;
    case 7:  // selu
        if (G == 0) y = (x >= 0) ? seluScale * x : (seluScale * seluAlpha) * expm1(x);
        if (G == 1) y = (yy >= 0) ? x * seluScale : x * (yy + seluScale * seluAlpha);
        if (G == 2) y = (yy >= 0) ? (M)0 : x * (yy + seluScale * seluAlpha);
        break;
    case 8:  // softplus
        if (G == 0) y = (x > (M)20) ? x : log1p(exp(x));
        if (G == 1) y = x * (one - exp(-yy));
        if (G == 2) { M c = exp(-yy); y = x * c * (one - c); }
        break;
    case 9:  // swish
        if (G == 0) {
            y = (x < -expRange) ? (M)0 : x / (exp(-x) + one);
        } else {
            M c = exp(xref), d = c + one;
            if (G == 1) y = (xref > halfExpRange) ? x : x * c * (xref + d) / (d * d);
            else        y = (xref > halfExpRange) ? (M)0 : x * c * (xref * (two - d) + two * d) / (d * d * d);
            yref = (xref < -expRange) ? (M)0 : xref / (exp(-xref) + one) * gain;
        }
        break;
    }
    return y;
}

template <class T, int A>
__device__ __forceinline__ typename Elem<T>::math_t
bias_act_one(typename Elem<T>::math_t x, typename Elem<T>::math_t b, typename Elem<T>::math_t xref,
             typename Elem<T>::math_t yref, typename Elem<T>::math_t dy, int G,
             typename Elem<T>::math_t alpha, typename Elem<T>::math_t gain, typename Elem<T>::math_t clamp) {
    using M = typename Elem<T>::math_t;
    M yy = (gain != 0) ? yref / gain : (M)0;
    if (G == 0) x += b; else xref += b;
    M y = act_eval<M>(A, G, x, xref, yy, alpha, yref, gain);
    y *= gain * dy;
    if (clamp >= 0) {
        if (G == 0) y = (y > -clamp && y < clamp) ? y : ((y >= 0) ? clamp : -clamp);
        else        y = (yref > -clamp && yref < clamp) ? y : (M)0;
    }
    return y;
}

// Vector kernel: each lane handles VEC contiguous elements per iteration (16 bytes).
template <class T, int A, int VEC>
__global__ void __launch_bounds__(256)
bias_act_vec_kernel(const T* __restrict__ x, const T* __restrict__ b, const T* __restrict__ xref,
                    const T* __restrict__ yref, const T* __restrict__ dy, T* __restrict__ y,
                    int G, float alpha_f, float gain_f, float clamp_f,
                    int64_t nvec, int64_t size_b, int64_t step_b, int bias_per_vec) {
    using M = typename Elem<T>::math_t;
    using V = Vec<T, VEC>;
    const M alpha = (M)alpha_f, gain = (M)gain_f, clamp = (M)clamp_f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t iv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; iv < nvec; iv += stride) {
        const int64_t i0 = iv * VEC;
        V vx = reinterpret_cast<const V*>(x)[iv];
        V vxr, vyr, vdy, vo;
        if (xref) vxr = reinterpret_cast<const V*>(xref)[iv];
        if (yref) vyr = reinterpret_cast<const V*>(yref)[iv];
        if (dy)   vdy = reinterpret_cast<const V*>(dy)[iv];
        M bv = 0;
        if (b && bias_per_vec) bv = Elem<T>::ld(b + (i0 / step_b) % size_b);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if (b && !bias_per_vec) bv = Elem<T>::ld(b + ((i0 + k) / step_b) % size_b);
            M xr = xref ? Elem<T>::ld(&vxr.v[k]) : (M)0;
            M yr = yref ? Elem<T>::ld(&vyr.v[k]) : (M)0;
            M d  = dy   ? Elem<T>::ld(&vdy.v[k]) : (M)1;
            M r = bias_act_one<T, A>(Elem<T>::ld(&vx.v[k]), bv, xr, yr, d, G, alpha, gain, clamp);
            Elem<T>::st(&vo.v[k], r);
        }
        reinterpret_cast<V*>(y)[iv] = vo;
    }
}

// Forward pass over whole bias planes (round 5).  The kernel above finds its bias with `(i / step_b) % size_b` on 64-bit indices: a
// software division of ~100 vector instructions per 16 bytes (profiles/round5/kernel_pmc.json: 4811 vector instructions per wave, VALU busy
// 64 % of the launch at [4, 64, 512, 512] - an element-wise kernel bound by its index arithmetic, not by HBM), and evaluates `yref / gain` for
// every element although the forward pass has no yref.  Here a workgroup works inside ONE plane of step_b elements (NCHW: one channel of
// one image), so the bias is a scalar found once per workgroup, the gradient operands do not exist (G = 0 at compile time) and the
// addresses are 32-bit offsets from the plane's base.  Same operations in the same order per element: bit-equal to the kernel above.
constexpr int BA_ITER = 4;                                   // vectors per lane
template <class T, int A, int VEC>
__global__ void __launch_bounds__(256)
bias_act_plane_kernel(const T* __restrict__ x, const T* __restrict__ b, T* __restrict__ y, float alpha_f, float gain_f, float clamp_f,
                      unsigned vec_per_plane, unsigned chunks, unsigned size_b) {
    using M = typename Elem<T>::math_t;
    using V = Vec<T, VEC>;
    const M alpha = (M)alpha_f, gain = (M)gain_f, clamp = (M)clamp_f;
    const unsigned plane = blockIdx.x / chunks, chunk = blockIdx.x - plane * chunks;        // uniform
    const M bv = b ? Elem<T>::ld(b + plane % size_b) : (M)0;
    const V* __restrict__ xv = reinterpret_cast<const V*>(x) + (int64_t)plane * vec_per_plane;
    V* __restrict__ yv = reinterpret_cast<V*>(y) + (int64_t)plane * vec_per_plane;
    const unsigned i0 = chunk * (256u * BA_ITER) + threadIdx.x;
    V vx[BA_ITER];
#pragma unroll
    for (int it = 0; it < BA_ITER; ++it) { const unsigned i = i0 + it * 256u; if (i < vec_per_plane) vx[it] = xv[i]; }
#pragma unroll
    for (int it = 0; it < BA_ITER; ++it) {
        const unsigned i = i0 + it * 256u;
        if (i >= vec_per_plane) break;
        V vo;
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            Elem<T>::st(&vo.v[k], bias_act_one<T, A>(Elem<T>::ld(&vx[it].v[k]), bv, (M)0, (M)0, (M)1, 0, alpha, gain, clamp));
        yv[i] = vo;
    }
}

// Scalar kernel for the tail elements and for unaligned bases.
template <class T, int A>
__global__ void __launch_bounds__(256)
bias_act_scalar_kernel(const T* __restrict__ x, const T* __restrict__ b, const T* __restrict__ xref,
                       const T* __restrict__ yref, const T* __restrict__ dy, T* __restrict__ y,
                       int G, float alpha_f, float gain_f, float clamp_f,
                       int64_t begin, int64_t end, int64_t size_b, int64_t step_b) {
    using M = typename Elem<T>::math_t;
    const M alpha = (M)alpha_f, gain = (M)gain_f, clamp = (M)clamp_f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += stride) {
        M bv = b ? Elem<T>::ld(b + (i / step_b) % size_b) : (M)0;
        M xr = xref ? Elem<T>::ld(xref + i) : (M)0;
        M yr = yref ? Elem<T>::ld(yref + i) : (M)0;
        M d  = dy   ? Elem<T>::ld(dy + i)   : (M)1;
        Elem<T>::st(y + i, bias_act_one<T, A>(Elem<T>::ld(x + i), bv, xr, yr, d, G, alpha, gain, clamp));
    }
}

template <class T, int A>
static int launch_bias_act(const void* x, const void* b, const void* xref, const void* yref,
                           const void* dy, void* y, int grad, float alpha, float gain, float clamp,
                           int64_t size_x, int64_t size_b, int64_t step_b, hipStream_t st) {
    constexpr int VEC = 16 / sizeof(T);
    auto aligned = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool can_vec = aligned(x) && aligned(xref) && aligned(yref) && aligned(dy) && aligned(y);
    int64_t nvec = can_vec ? size_x / VEC : 0;
    // forward pass over whole planes: the bias is constant inside a plane of step_b elements, planes of >= 256 vectors
    if (nvec > 0 && grad == 0 && !xref && !yref && !dy && !knob_live("IDE3D_BIAS_ACT_NO_PLANES")) {
        const int64_t plane = b ? step_b : size_x;
        if (plane % VEC == 0 && size_x % plane == 0 && plane / VEC >= 256 && plane / VEC < (1ll << 31) && size_b < (1ll << 31)) {
            const int64_t vpp = plane / VEC, chunks = (vpp + 256 * BA_ITER - 1) / (256 * BA_ITER), blocks = (size_x / plane) * chunks;
            if (blocks < (1ll << 31)) {
                hipLaunchKernelGGL((bias_act_plane_kernel<T, A, VEC>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (const T*)b, (T*)y,
                                   alpha, gain, clamp, (unsigned)vpp, (unsigned)chunks, (unsigned)size_b);
                IDE3D_CHECK_LAUNCH("bias_act");
                return IDE3D_OK;
            }
        }
    }
    if (nvec > 0) {
        const int bias_per_vec = (step_b % VEC == 0) ? 1 : 0;
        int grid = stream_grid(nvec, 256);
        hipLaunchKernelGGL((bias_act_vec_kernel<T, A, VEC>), dim3(grid), dim3(256), 0, st,
                           (const T*)x, (const T*)b, (const T*)xref, (const T*)yref, (const T*)dy, (T*)y,
                           grad, alpha, gain, clamp, nvec, size_b, step_b, bias_per_vec);
    }
    const int64_t done = nvec * VEC;
    if (done < size_x) {
        int grid = stream_grid(size_x - done, 256);
        hipLaunchKernelGGL((bias_act_scalar_kernel<T, A>), dim3(grid), dim3(256), 0, st,
                           (const T*)x, (const T*)b, (const T*)xref, (const T*)yref, (const T*)dy, (T*)y,
                           grad, alpha, gain, clamp, done, size_x, size_b, step_b);
    }
    IDE3D_CHECK_LAUNCH("bias_act");
    return IDE3D_OK;
}

template <class T>
static int dispatch_act(int act, const void* x, const void* b, const void* xref, const void* yref,
                        const void* dy, void* y, int grad, float alpha, float gain, float clamp,
                        int64_t size_x, int64_t size_b, int64_t step_b, hipStream_t st) {
#define IDE3D_ACT_CASE(A) case A: return launch_bias_act<T, A>(x, b, xref, yref, dy, y, grad, alpha, gain, clamp, size_x, size_b, step_b, st);
    switch (act) {
        IDE3D_ACT_CASE(1) IDE3D_ACT_CASE(2) IDE3D_ACT_CASE(3) IDE3D_ACT_CASE(4) IDE3D_ACT_CASE(5)
        IDE3D_ACT_CASE(6) IDE3D_ACT_CASE(7) IDE3D_ACT_CASE(8) IDE3D_ACT_CASE(9)
    }
#undef IDE3D_ACT_CASE
    set_error("bias_act: unknown activation index %d", act);
    return IDE3D_EINVAL;
}

}  // namespace ide3d

extern "C" int ide3d_bias_act(const void* x, const void* b, const void* xref, const void* yref,
                              const void* dy, void* y, int dtype, int grad, int act,
                              float alpha, float gain, float clamp,
                              int64_t size_x, int64_t size_b, int64_t step_b, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(size_x >= 0, "bias_act: negative size");
    if (size_x == 0) return IDE3D_OK;
    IDE3D_CHECK_ARG(x && y, "bias_act: x and y must be non-null");
    IDE3D_CHECK_ARG(grad >= 0 && grad <= 2, "bias_act: grad must be 0, 1 or 2 (got %d)", grad);
    IDE3D_CHECK_ARG(b == nullptr || (size_b > 0 && step_b > 0), "bias_act: bad bias geometry");
    IDE3D_CHECK_ARG(grad < 2 || dy != nullptr, "bias_act: grad=2 needs dy");
    if (b == nullptr) { size_b = 1; step_b = 1; }
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case IDE3D_F32:  return dispatch_act<float>(act, x, b, xref, yref, dy, y, grad, alpha, gain, clamp, size_x, size_b, step_b, st);
    case IDE3D_F16:  return dispatch_act<__half>(act, x, b, xref, yref, dy, y, grad, alpha, gain, clamp, size_x, size_b, step_b, st);
    case IDE3D_BF16: return dispatch_act<__hip_bfloat16>(act, x, b, xref, yref, dy, y, grad, alpha, gain, clamp, size_x, size_b, step_b, st);
    case IDE3D_F64:  return dispatch_act<double>(act, x, b, xref, yref, dy, y, grad, alpha, gain, clamp, size_x, size_b, step_b, st);
    }
    set_error("bias_act: unsupported dtype code %d", dtype);
    return IDE3D_EINVAL;
}
