// raymarch.hip — fused volumetric renderer for gfx950: ray set-up -> cam2world -> two tri-plane
// gathers -> decoder MLPs (MFMA) -> depth-ordered alpha compositing, in ONE launch.
//
// Replaces steps 3-7 of G.synthesis:
//   get_initial_rays_trig      training/volumetric_rendering.py:77-97   (points = d_cam * z)
//   perturb_points             :99-105   (offset = (U - 0.5) * (z[1] - z[0]); z += offset; p += offset * d)
//   transform_sampled_points   :108-136  (p_world = cam2world @ [p, 1])
//   sample_from_triplane x2    dnnlib/util.py:580-617
//   renderer.sample_voxel MLPs (source absent upstream; spec = DESIGN.md "decoder")
//   fancy_integration          training/volumetric_rendering.py:34-74
//
// CDNA4 mapping (round 3: DESIGN.md section 5.2c)
//   * One wavefront owns one ray and walks it in tiles of 16 depth samples, software-pipelined over the tiles of all its rays: a
//     tile's texture taps are in flight during its geometry MLP, the next tile's geometry taps during its texture MLP.
//   * Two lane layouts.  The gathers run with lane 4 j + g (sample j, 16-byte channel block g): the four lanes of a quad read 64
//     contiguous bytes of one bilinear tap — one L1 access; every tap load is the scalar-base form (plane pointer in SGPRs, 32-bit
//     byte offset, plane / slice in the immediate), all 24 of a tri-plane issued before the first is used.  The blended features
//     change lanes once (ds_bpermute) into the matrix layout lane 16 g + j (g = K block, j = column / sample).
//   * The two MLPs run in the *transposed* form out^T[features x samples] = W[features x K] * act^T[K x samples]: the features sit in
//     the B-operand layout and the D layout of layer 1 (row = 4*(l>>4)+r, col = l&15) is again a valid B operand for layer 2 once K
//     is enumerated accordingly — activations never leave registers between gather, layer 1, layer 2 and compositing.  Weights are
//     re-ordered once per workgroup into LDS in A-operand order.  Arithmetic: exact fp32 products on v_mfma_f32_16x16x4_f32
//     (library default), or bf16x6 (fp32-grade: 3 bf16 pieces per operand, 6 products) on v_mfma_f32_16x16x32_bf16 when a split
//     arithmetic is selected (ide3d_set_conv_arithmetic).
//   * Compositing: sigma is broadcast from the g = 0 lanes, alpha / transmittance are evaluated by a 16-lane segmented shuffle scan
//     with the running transmittance carried across tiles, each lane accumulates w * feature for the 16 output features it holds
//     (the geometry outputs right after their MLP: they die before the texture MLP), a 4-step xor-shuffle reduction over the 16
//     samples closes the ray.  Nothing but the final [n, feat+seg, rays] image, depth and weight sum is written.
//   * Persistent workgroups (2 per CU) take contiguous ray ranges; the range order is XCD-remapped so neighbouring rays
//     (neighbouring plane lines) share one XCD L2.
// Compulsory HBM traffic per image: 2 tri-planes + jitter/noise in + (feat+seg+2)*rays*4 out.  The planes are cache resident
// (L2 4 MiB/XCD + 256 MiB MALL); the kernel is bound by its vector instructions (~70 % of the VALU pipe at 2 waves per SIMD), not by HBM
// or the matrix pipe — see DESIGN.md.
#include "common.h"
#include "triplane_tap.h"

namespace ide3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int C, int HID>
struct RmCfg {
    static constexpr int CPL = C / 16;          // float4 chunks per lane per tap
    static constexpr int NF = C / 4;            // features (K elements) per lane
    static constexpr int MT1 = HID / 16;        // layer-1 M tiles
    static constexpr int MT2 = 2;               // layer-2 M tiles (<= 32 outputs)
    // LDS carve per MLP, in floats
    static constexpr int A1 = MT1 * CPL * 64 * 4;
    static constexpr int A2 = MT2 * MT1 * 64 * 4;
    static constexpr int B0 = MT1 * 16;         // [mt][g][4]
    static constexpr int B1 = MT2 * 16;
    static constexpr int MLP = A1 + A2 + B0 + B1;
};

// Re-order one MLP's weights into LDS (A-operand order).  w0 [HID, C], w1 [nout, HID] (nout <= 32).
template <int C, int HID>
__device__ void stage_mlp(float* __restrict__ s, const float* __restrict__ w0, const float* __restrict__ b0,
                          const float* __restrict__ w1, const float* __restrict__ b1, int nout) {
    using K = RmCfg<C, HID>;
    float* a1 = s; float* a2 = a1 + K::A1; float* sb0 = a2 + K::A2; float* sb1 = sb0 + K::B0;
    for (int i = threadIdx.x; i < K::A1; i += blockDim.x) {
        const int e = i & 3, lane = (i >> 2) & 63, rest = i >> 8;
        const int ci = rest % K::CPL, mt = rest / K::CPL;
        const int row = 16 * mt + (lane & 15), col = 4 * ((lane >> 4) + 4 * ci) + e;
        a1[i] = w0[row * C + col];
    }
    for (int i = threadIdx.x; i < K::A2; i += blockDim.x) {
        const int e = i & 3, lane = (i >> 2) & 63, rest = i >> 8;
        const int tq = rest % K::MT1, mt = rest / K::MT1;
        const int row = 16 * mt + (lane & 15), col = 16 * tq + 4 * (lane >> 4) + e;
        a2[i] = (row < nout) ? w1[row * HID + col] : 0.f;
    }
    for (int i = threadIdx.x; i < K::B0; i += blockDim.x) sb0[i] = b0[i];          // index = 16 mt + 4 g + r
    for (int i = threadIdx.x; i < K::B1; i += blockDim.x) sb1[i] = (i < nout) ? b1[i] : 0.f;
}

__device__ __forceinline__ float softplus_fast(float x) {
    // softplus(x) = max(x, 0) + log(1 + exp(-|x|)); abs error ~1e-7, saturates like threshold=20.  Raw v_exp_f32 / v_log_f32: the
    // exponent is <= 0 (a result below the normal range is 0 beside the 1 it is added to) and the logarithm's argument is in [1, 2], so
    // the denormal guards of __expf / __logf (a compare, a select and an ldexp each) have nothing to do here.
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * fabsf(x));
    return fmaf(0.69314718055994530942f, __builtin_amdgcn_logf(1.0f + e), fmaxf(x, 0.f));
}

// ---- decoder MLPs on the bf16 matrix path (round 3) -----------------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (1/16 of bf16): the two MLPs of a 16-sample tile are 128 of them = 4096 cycles
// of matrix pipe per wave and tile.  With the split arithmetic of the convolutions selected (ide3d_set_conv_arithmetic != fp32) the MLPs
// run as bf16x6 instead: every fp32 operand = 3 bf16 pieces, the 6 products above 2^-24 on v_mfma_f32_16x16x32_bf16 (K = 32 per
// instruction: layer 1 is ONE K step, layer 2 two), fp32 accumulation — 48 instead of 128 matrix instructions per tile at ~17 instead of
// 32 cycles each, fp32-grade products (no range to manage: bf16 has fp32's exponent).  The register layouts carry over: a lane (g, j)
// holds channels {4 g + e} and {16 + 4 g + e} of sample j = the 8 K values of K block g under the enumeration k(g, i) = 16 (i / 4) + 4 g
// + i % 4, and the layer-1 D registers (units 16 mt + 4 g + r) are the 8 K values of block g of K half kh = mt / 2 under
// u(kh, g, i) = 16 (2 kh + i / 4) + 4 g + i % 4; the weights are split and stored in exactly these orders once per workgroup.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int C, int HID>
struct RmSplit {
    static_assert(C == 32 && HID % 32 == 0, "split MLPs: 32 input channels (one K = 32 step), hidden width a multiple of 32");
    static constexpr int MT1 = HID / 16, KH = HID / 32;
    static constexpr int A1 = MT1 * 3 * 64;              // 16-byte units: [mt][piece][lane]
    static constexpr int A2 = 2 * KH * 3 * 64;           // [m2][k half][piece][lane]
    static constexpr int BIAS = (MT1 + 2) * 16;          // floats
    static constexpr int BYTES = (A1 + A2) * 16 + BIAS * 4;
};

// a, b -> three packed bf16 pairs (round to nearest even; the residuals are exact in fp32): low half a, high half b
// The residuals come from v_dot2c_f32_bf16: x - hi = dot((hi_a, hi_b), (-1, 0)) + x, one instruction per value instead of an unpack and
// a subtraction (exact: scripts/micro/split_dot2_check.hip compares the pieces of 2^24 values bit for bit with the mask / subtract form).
__device__ __forceinline__ void split_pair3(float a, float b, unsigned (&out)[3]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const f32x2 v = {a, b};
        const bf16x2 pk = __builtin_convertvector(v, bf16x2);
        out[q] = __builtin_bit_cast(unsigned, pk);
#if defined(IDE3D_SPLIT_NO_DOT2)
        if (q < 2) { a -= __uint_as_float(out[q] << 16); b -= __uint_as_float(out[q] & 0xffff0000u); }
#else
        // (the multipliers go through scalar registers: as immediates hipcc emits the inline constant -1.0 for (-1, 0), which the instruction
        // does not read as bf16 (-1, 0))
        unsigned clo = 0x0000bf80u, chi = 0xbf800000u;
        asm volatile("" : "+s"(clo), "+s"(chi));
        const bf16x2 mlo = __builtin_bit_cast(bf16x2, clo), mhi = __builtin_bit_cast(bf16x2, chi);
        if (q < 2) { a = __builtin_amdgcn_fdot2_f32_bf16(pk, mlo, a, false); b = __builtin_amdgcn_fdot2_f32_bf16(pk, mhi, b, false); }
#endif
    }
}
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&out)[3]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned pk[3];
        split_pair3(v[2 * e], v[2 * e + 1], pk);
        out[0][e] = pk[0]; out[1][e] = pk[1]; out[2][e] = pk[2];
    }
}
__device__ __forceinline__ f32x4 mfma6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4 acc) {
    // smallest products first
#pragma unroll
    for (int sum = 2; sum >= 0; --sum)
#pragma unroll
        for (int qa = 0; qa <= sum; ++qa)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[qa]), __builtin_bit_cast(bf16x8, b[sum - qa]), acc, 0, 0, 0);
    return acc;
}

template <int C, int HID>
__device__ void stage_mlp_split(unsigned char* __restrict__ sm, const float* __restrict__ w0, const float* __restrict__ b0,
                                const float* __restrict__ w1, const float* __restrict__ b1, int nout) {
    using K = RmSplit<C, HID>;
    u32x4* a1 = reinterpret_cast<u32x4*>(sm);
    u32x4* a2 = a1 + K::A1;
    float* sb = reinterpret_cast<float*>(a2 + K::A2);
    for (int i = threadIdx.x; i < K::MT1 * 64; i += blockDim.x) {
        const int lane = i & 63, mt = i >> 6, g = lane >> 4, row = 16 * mt + (lane & 15);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = w0[row * C + 16 * (k / 4) + 4 * g + k % 4];
        u32x4 pc[3];
        split8(v, pc);
#pragma unroll
        for (int q = 0; q < 3; ++q) a1[(mt * 3 + q) * 64 + lane] = pc[q];
    }
    for (int i = threadIdx.x; i < 2 * K::KH * 64; i += blockDim.x) {
        const int lane = i & 63, kh = (i >> 6) % K::KH, m2 = i / (64 * K::KH), g = lane >> 4, row = 16 * m2 + (lane & 15);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (row < nout) ? w1[row * HID + 16 * (2 * kh + k / 4) + 4 * g + k % 4] : 0.f;
        u32x4 pc[3];
        split8(v, pc);
#pragma unroll
        for (int q = 0; q < 3; ++q) a2[((m2 * K::KH + kh) * 3 + q) * 64 + lane] = pc[q];
    }
    for (int i = threadIdx.x; i < K::MT1 * 16; i += blockDim.x) sb[i] = b0[i];                       // index = 16 mt + 4 g + r
    for (int i = threadIdx.x; i < 32; i += blockDim.x) sb[K::MT1 * 16 + i] = (i < nout) ? b1[i] : 0.f;
}

template <int C, int HID>
__device__ __forceinline__ void mlp_hidden_split(const unsigned char* __restrict__ sm, const float (&f)[C / 4], f32x4 (&h)[HID / 16]) {
    using K = RmSplit<C, HID>;
    const int lane = lane_id(), g = lane >> 4;
    const u32x4* a1 = reinterpret_cast<const u32x4*>(sm);
    const f32x4* sb0 = reinterpret_cast<const f32x4*>(sm + (K::A1 + K::A2) * 16);
    u32x4 fb[3];
    split8(f, fb);
#pragma unroll
    for (int mt = 0; mt < K::MT1; ++mt) {
        const u32x4 a[3] = {a1[(mt * 3 + 0) * 64 + lane], a1[(mt * 3 + 1) * 64 + lane], a1[(mt * 3 + 2) * 64 + lane]};
        h[mt] = mfma6(a, fb, sb0[mt * 4 + g]);
    }
#pragma unroll
    for (int mt = 0; mt < K::MT1; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[mt][r] = softplus_fast(h[mt][r]);
}

template <int C, int HID>
__device__ __forceinline__ void mlp_tile_split(const unsigned char* __restrict__ sm, const float (&f)[C / 4], f32x4 (&out)[2]) {
    using K = RmSplit<C, HID>;
    const int lane = lane_id(), g = lane >> 4;
    const u32x4* a2 = reinterpret_cast<const u32x4*>(sm) + K::A1;
    const f32x4* sb1 = reinterpret_cast<const f32x4*>(sm + (K::A1 + K::A2) * 16) + K::MT1 * 4;
    f32x4 h[K::MT1];
    mlp_hidden_split<C, HID>(sm, f, h);
    u32x4 hb[K::KH][3];
#pragma unroll
    for (int kh = 0; kh < K::KH; ++kh) {
        const float v[8] = {h[2 * kh][0], h[2 * kh][1], h[2 * kh][2], h[2 * kh][3], h[2 * kh + 1][0], h[2 * kh + 1][1], h[2 * kh + 1][2], h[2 * kh + 1][3]};
        split8(v, hb[kh]);
    }
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2) {
        f32x4 acc = sb1[m2 * 4 + g];
#pragma unroll
        for (int kh = 0; kh < K::KH; ++kh) {
            const int o = ((m2 * K::KH + kh) * 3) * 64 + lane;
            const u32x4 a[3] = {a2[o], a2[o + 64], a2[o + 128]};
            acc = mfma6(a, hb[kh], acc);
        }
        out[m2] = acc;
    }
}

// (density only, split form) hidden layer on the bf16 matrix path, row 0 of the second layer as a VALU dot product like mlp_sigma
template <int C, int HID>
__device__ __forceinline__ float mlp_sigma_split(const unsigned char* __restrict__ sm, const float* __restrict__ s_row, const float (&f)[C / 4]) {
    using K = RmSplit<C, HID>;
    const int g = lane_id() >> 4;
    f32x4 h[K::MT1];
    mlp_hidden_split<C, HID>(sm, f, h);
    float acc = 0.f;
#pragma unroll
    for (int mt = 0; mt < K::MT1; ++mt) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(s_row + 16 * mt + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc += w[r] * h[mt][r];
    }
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    return acc + s_row[HID];
}

// Gather this lane's NF features of one sample from one tri-plane (channels_last, channel stride 1).
// Lane layouts.  The matrix instructions want lane 16 g + j to hold K block g of sample (column) j: neighbouring lanes are
// DIFFERENT samples.  Gathering in that layout makes every lane of a 16-byte load its own cache-line access (16 useful bytes per L1
// access; 64 accesses per instruction; the kernel then runs at exactly the L1 access rate: 2 tri-planes x 12 taps x 8 accesses per
// sample = 0.49 ms for the 1.57 M samples of the benchmark pass, measured 0.50 — round 3).  So the gathers run in a second layout,
// lane 4 j + g (`gather layout`: four neighbouring lanes read 64 contiguous bytes of one tap of one sample — one access), each lane
// blends its 16-byte slices with the tap weights of sample lane / 4, and the blended features — 8 values per lane and tri-plane —
// change lanes once (ds_bpermute, no memory) into the matrix layout.
__device__ __forceinline__ int gather_sample(int lane) { return lane >> 2; }
__device__ __forceinline__ int gather_block(int lane) { return lane & 3; }

template <int NF>
__device__ __forceinline__ void to_matrix_lanes(float (&f)[NF]) {
    const int lane = lane_id();
    const int src = (((lane & 15) << 2) | (lane >> 4)) << 2;        // lane 16 g + j reads lane 4 j + g (byte address of its dword)
#pragma unroll
    for (int i = 0; i < NF; ++i) f[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(f[i])));
}

template <int C> struct TapBuf { float4 v[C / 16][3][4]; };

// a global-memory address known to be the same in every lane -> scalar registers (what the scalar-base load form needs)
typedef const __attribute__((address_space(1))) char* GlobalBytes;
template <class T>
__device__ __forceinline__ GlobalBytes uniform_ptr(const T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<GlobalBytes>(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ float4 ld4_at(GlobalBytes base, unsigned byte_off, unsigned imm) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) v4f* GlobalF4;
    const v4f v = *reinterpret_cast<GlobalF4>(base + (size_t)byte_off + imm);
    return make_float4(v[0], v[1], v[2], v[3]);
}

template <int C>
__device__ __forceinline__ void issue_taps(GlobalBytes pbc, const TapAddr (&t)[3], int g, TapBuf<C>& b, unsigned lane_extra = 0) {
    constexpr int CPL = C / 16;
    // Every 16-byte tap load of the sample — 12 per channel slice, all CPL slices — is issued before the first one is consumed; left
    // to itself the compiler loads a plane's four taps, waits, blends.
    // `pb` is wave-uniform and the tap offsets are unsigned 32-bit byte counts (planes_fast bounds them), so every load is the
    // scalar-base form — global_load_dwordx4 v, v_off, s[base:base+1] offset:imm — with the plane / slice part in the immediate: no
    // 64-bit address arithmetic per tap.
    const unsigned lane_b = 16u * (unsigned)g + lane_extra;          // lane_extra: a per-lane byte offset on top (an image further on)
#pragma unroll
    for (int ci = 0; ci < CPL; ++ci) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const unsigned imm = (unsigned)(pl * C + 16 * ci) * 4u;
            b.v[ci][pl][0] = ld4_at(pbc, (unsigned)t[pl].o00 * 4u + lane_b, imm); b.v[ci][pl][1] = ld4_at(pbc, (unsigned)t[pl].o01 * 4u + lane_b, imm);
            b.v[ci][pl][2] = ld4_at(pbc, (unsigned)t[pl].o10 * 4u + lane_b, imm); b.v[ci][pl][3] = ld4_at(pbc, (unsigned)t[pl].o11 * 4u + lane_b, imm);
        }
    }
}

template <int C>
__device__ __forceinline__ void blend_taps(TapBuf<C>& b, const TapAddr (&t)[3], float (&f)[C / 4]) {
    constexpr int CPL = C / 16;
#pragma unroll
    for (int ci = 0; ci < CPL; ++ci)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(b.v[ci][pl][k].x), "+v"(b.v[ci][pl][k].y), "+v"(b.v[ci][pl][k].z), "+v"(b.v[ci][pl][k].w));
#pragma unroll
    for (int ci = 0; ci < CPL; ++ci) {
        float4 a[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            acc = f4_fma(b.v[ci][pl][0], t[pl].w00, acc);
            acc = f4_fma(b.v[ci][pl][1], t[pl].w01, acc);
            acc = f4_fma(b.v[ci][pl][2], t[pl].w10, acc);
            acc = f4_fma(b.v[ci][pl][3], t[pl].w11, acc);
            a[pl] = acc;
        }
        f[4 * ci + 0] = (a[0].x + a[1].x) + a[2].x;
        f[4 * ci + 1] = (a[0].y + a[1].y) + a[2].y;
        f[4 * ci + 2] = (a[0].z + a[1].z) + a[2].z;
        f[4 * ci + 3] = (a[0].w + a[1].w) + a[2].w;
    }
}

// `t`: taps of sample gather_sample(lane); `g` = gather_block(lane).  Result in the gather layout (f[4 ci + e] = channel
// 16 ci + 4 g + e of that sample): follow with to_matrix_lanes.
template <int C>
__device__ __forceinline__ void gather_features(const float* __restrict__ pb, const TapAddr (&t)[3], int g, float (&f)[C / 4]) {
    TapBuf<C> b;
    issue_taps<C>(uniform_ptr(pb), t, g, b);
    blend_taps<C>(b, t, f);
}

// Two-layer MLP on a 16-sample tile, transposed MFMA form.  out[mt][r] = feature 16 mt + 4 g + r of sample j.
template <int C, int HID>
__device__ __forceinline__ void mlp_hidden(const float* __restrict__ s, const float (&f)[C / 4], f32x4 (&h)[HID / 16]) {
    using K = RmCfg<C, HID>;
    const int lane = lane_id();
    const f32x4* a1 = reinterpret_cast<const f32x4*>(s);
    const f32x4* sb0 = reinterpret_cast<const f32x4*>(s + K::A1 + K::A2);
    const int g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < K::MT1; ++mt) h[mt] = sb0[mt * 4 + g];
#pragma unroll
    for (int ci = 0; ci < K::CPL; ++ci) {
        f32x4 w[K::MT1];
#pragma unroll
        for (int mt = 0; mt < K::MT1; ++mt) w[mt] = a1[(mt * K::CPL + ci) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < K::MT1; ++mt)
                h[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[mt][e], f[4 * ci + e], h[mt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < K::MT1; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[mt][r] = softplus_fast(h[mt][r]);
}

template <int C, int HID>
__device__ __forceinline__ void mlp_tile(const float* __restrict__ s, const float (&f)[C / 4], f32x4 (&out)[2]) {
    using K = RmCfg<C, HID>;
    const int lane = lane_id();
    const f32x4* a2 = reinterpret_cast<const f32x4*>(s + K::A1);
    const f32x4* sb1 = reinterpret_cast<const f32x4*>(s + K::A1 + K::A2 + K::B0);
    const int g = lane >> 4;
    f32x4 h[K::MT1];
    mlp_hidden<C, HID>(s, f, h);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) out[mt] = sb1[mt * 4 + g];
#pragma unroll
    for (int tq = 0; tq < K::MT1; ++tq) {
        const f32x4 w0 = a2[(0 * K::MT1 + tq) * 64 + lane];
        const f32x4 w1 = a2[(1 * K::MT1 + tq) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            out[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[e], h[tq][e], out[0], 0, 0, 0);
            out[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[e], h[tq][e], out[1], 0, 0, 0);
        }
    }
}

// Density only: of the second layer just row 0 (sigma) is needed — a 64-term dot product per sample.  The hidden vector of a
// sample is spread over its four lanes (lane (g, j) holds units 16 mt + 4 g + r), so each lane forms a 16-term partial sum on
// the VALU and two xor-shuffles across g close it: ~20 vector instructions instead of the 32 MFMAs (1024 cycles) of the
// 32-row layer.  s_row: w1[0][0..HID) followed by b1[0].
template <int C, int HID>
__device__ __forceinline__ float mlp_sigma(const float* __restrict__ s, const float* __restrict__ s_row, const float (&f)[C / 4]) {
    using K = RmCfg<C, HID>;
    const int g = lane_id() >> 4;
    f32x4 h[K::MT1];
    mlp_hidden<C, HID>(s, f, h);
    float acc = 0.f;
#pragma unroll
    for (int mt = 0; mt < K::MT1; ++mt) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(s_row + 16 * mt + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc += w[r] * h[mt][r];
    }
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    return acc + s_row[HID];
}

__device__ __forceinline__ float seg16_excl_prod(float v, float& total) {
    // exclusive prefix product inside each aligned group of 16 lanes
    const int j = lane_id() & 15;
    float incl = v;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const float o = __shfl_up(incl, off, 16);
        if (j >= off) incl *= o;
    }
    total = __shfl(incl, 15, 16);
    const float prev = __shfl_up(incl, 1, 16);
    return j == 0 ? 1.0f : prev;
}

// MLP storage of one decoder branch in LDS, by arithmetic (SPLIT: bf16x6 on v_mfma_f32_16x16x32_bf16; else fp32 MFMA)
template <int C, int HID, bool SPLIT> struct MlpBytes { static constexpr int value = RmCfg<C, HID>::MLP * 4; };
template <int C, int HID> struct MlpBytes<C, HID, true> { static constexpr int value = RmSplit<C, HID>::BYTES; };

template <int C, int HID, bool SPLIT>
__device__ __forceinline__ void stage_branch(unsigned char* sm, const float* w0, const float* b0, const float* w1, const float* b1, int nout) {
    if constexpr (SPLIT) stage_mlp_split<C, HID>(sm, w0, b0, w1, b1, nout);
    else stage_mlp<C, HID>(reinterpret_cast<float*>(sm), w0, b0, w1, b1, nout);
}
template <int C, int HID, bool SPLIT>
__device__ __forceinline__ void mlp_branch(const unsigned char* sm, const float (&f)[C / 4], f32x4 (&out)[2]) {
    if constexpr (SPLIT) mlp_tile_split<C, HID>(sm, f, out);
    else mlp_tile<C, HID>(reinterpret_cast<const float*>(sm), f, out);
}

// Exclusive residency (modconv.hip `kSpExclusive`): the SPLIT forms of the three kernels below run LDS-fed bf16
// MFMA loops, beside which a foreign wave's packed-fp32 instructions return wrong values on MI355X.  They are therefore launched as 8-wave
// workgroups, one per CU: two of the workgroup's own waves, 256 registers each (`rm_claim_half_simd`), fill every SIMD, and no wave leaves
// before all have finished (`__syncthreads()` at the end), so no foreign wave ever shares a SIMD with a running MLP.  The fp32 forms
// (v_mfma_f32_16x16x4_f32 was never observed as a neighbour that matters) keep 4-wave workgroups, two per CU.
template <bool SPLIT> struct RmWaves { static constexpr int value = SPLIT ? 8 : 4; };
template <bool SPLIT> __device__ __forceinline__ void rm_claim_half_simd() { if constexpr (SPLIT) asm volatile("" ::: "v255"); }

template <int C, int HID, bool SPLIT>
__global__ void __launch_bounds__(64 * RmWaves<SPLIT>::value, 2)
render_rays_kernel(ide3d_render_params p, int64_t rays_per_block) {
    using K = RmCfg<C, HID>;
    constexpr int NW = RmWaves<SPLIT>::value;
    rm_claim_half_simd<SPLIT>();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned char* s_geo = reinterpret_cast<unsigned char*>(lds);
    unsigned char* s_tex = s_geo + MlpBytes<C, HID, SPLIT>::value;
    stage_branch<C, HID, SPLIT>(s_geo, p.geo_w0, p.geo_b0, p.geo_w1, p.geo_b1, 1 + p.seg_ch);
    stage_branch<C, HID, SPLIT>(s_tex, p.tex_w0, p.tex_b0, p.tex_w1, p.tex_b1, p.feat_ch);
    __syncthreads();

    const int lane = lane_id(), wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15, gl = gather_block(lane);
    const int64_t total_rays = (int64_t)p.n * p.rays_per_img;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t ray_begin = (int64_t)blk * rays_per_block;
    int64_t ray_end = ray_begin + rays_per_block;
    if (ray_end > total_rays) ray_end = total_rays;
    const int S = p.steps;
    const int nch = p.feat_ch + p.seg_ch;
    const int sH = (int)p.tex_stride[2], sW = (int)p.tex_stride[3];   // host guarantees tex / geo share the layout
    const float zstep = (S > 1) ? (p.z_lin[1] - p.z_lin[0]) : 0.f;

    // Depth of sample s0 + lane / 4 of `ray` (gather layout): the two small loads the tap addresses hang on, issued one tile ahead.
    auto tile_depth = [&](int64_t ray, int s0, float& zl, float& jl) {
        const int sl = min(s0 + gather_sample(lane), S - 1);
        zl = p.z_lin[sl];
        jl = p.jitter ? p.jitter[ray * S + sl] : 0.5f;
    };
    // Taps of that sample: camera space -> jitter -> world.  Everything per ray is wave-uniform (scalar loads), so a tile of the NEXT
    // ray costs the same as one of this ray.
    auto tile_taps = [&](int n, int r, float zl, float jl, TapAddr (&t)[3]) {
        const float dx = p.rays_d_cam[r * 3 + 0], dy = p.rays_d_cam[r * 3 + 1], dz = p.rays_d_cam[r * 3 + 2];
        const float* M = p.cam2world + n * 16;
        float px = __fmul_rn(dx, zl), py = __fmul_rn(dy, zl), pz = __fmul_rn(dz, zl);
        if (p.jitter) {
            const float off = __fmul_rn(__fsub_rn(jl, 0.5f), zstep);
            px = __fadd_rn(px, __fmul_rn(off, dx));
            py = __fadd_rn(py, __fmul_rn(off, dy));
            pz = __fadd_rn(pz, __fmul_rn(off, dz));
        }
        const float wx = fmaf(M[0], px, fmaf(M[1], py, fmaf(M[2], pz, M[3])));
        const float wy = fmaf(M[4], px, fmaf(M[5], py, fmaf(M[6], pz, M[7])));
        const float wz = fmaf(M[8], px, fmaf(M[9], py, fmaf(M[10], pz, M[11])));
        t[0] = make_tap_addr(wx, wy, p.W, p.H, sH, sW);
        t[1] = make_tap_addr(wy, wz, p.W, p.H, sH, sW);
        t[2] = make_tap_addr(wx, wz, p.W, p.H, sH, sW);
    };

    // Software pipeline over the tiles (ray, s0) of this wave, flattened across rays.  The loads of a tile's texture taps fly during
    // its geometry MLP and the geometry taps of the NEXT tile during its texture MLP: with the L1 access rate out of the way (gather
    // layout) the kernel is otherwise the sum of its exposed gather round trips and its arithmetic (2 waves per SIMD hide little).
    // vmcnt retires in order, so nothing that is needed during an MLP may be loaded after the taps that fly across it: the small
    // loads of a tile (its depths, the next tile's depths) are issued first, and the kernel must not spill (a scratch reload inside
    // an MLP would wait for the taps).
    // (ray, image n, ray-in-image r, s0) of the current tile: wave-uniform, kept in scalar registers (the image index advances by
    // comparison, not by a 64-bit division per tile; plane bases are scalar pointers for the scalar-base loads)
    int64_t ray = ray_begin + wid;
    int n = __builtin_amdgcn_readfirstlane((int)(ray / p.rays_per_img));
    int r = __builtin_amdgcn_readfirstlane((int)(ray - (int64_t)n * p.rays_per_img));
    int s0 = 0;
    bool have = ray < ray_end;
    TapAddr t[3];
    TapBuf<C> buf;
    if (have) {
        float zl, jl;
        tile_depth(ray, 0, zl, jl);
        tile_taps(n, r, zl, jl, t);
        issue_taps<C>(uniform_ptr(p.geo_planes + n * p.geo_stride[0]), t, gl, buf);
    }
    float acc_t[8], acc_g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc_t[i] = 0.f; acc_g[i] = 0.f; }
    float carry = 1.0f, wsum = 0.f, dsum = 0.f;

    while (have) {
        // successor tile
        int s0n = s0 + 16, nn = n, rn = r;
        int64_t rayn = ray;
        if (s0n >= S) {
            s0n = 0; rayn = ray + NW; rn = r + NW;
            while (rn >= p.rays_per_img) { rn -= (int)p.rays_per_img; ++nn; }
        }
        const bool haven = rayn < ray_end;
        // the prefetches below are unconditional (a conditional definition of the loop-carried tap buffer costs a register copy per
        // value and iteration): past the last tile they re-read this one
        const int64_t rayp = haven ? rayn : ray;
        const int np = haven ? nn : n, rp = haven ? rn : r, s0p = haven ? s0n : s0;
        // --- matrix layout: depth of sample s0 + lane % 16 for the compositing; gather layout: depths of the next tile ---
        const int s = s0 + j;
        const bool live = s < S;
        const int sc = live ? s : S - 1;
        float z = p.z_lin[sc];
        float znext = (sc + 1 < S) ? p.z_lin[sc + 1] : 0.f;
        if (p.jitter) {
            const float* jit = p.jitter + ray * S;
            z = __fadd_rn(z, __fmul_rn(__fsub_rn(jit[sc], 0.5f), zstep));
            if (sc + 1 < S) znext = __fadd_rn(znext, __fmul_rn(__fsub_rn(jit[sc + 1], 0.5f), zstep));
        }
        const float noise = p.sigma_noise ? p.sigma_noise[ray * S + sc] : 0.f;
        float zln, jln;
        tile_depth(rayp, s0p, zln, jln);

        float fg[K::NF], ft[K::NF];
        f32x4 og[2], ot[2];
        blend_taps<C>(buf, t, fg);
        to_matrix_lanes(fg);
        issue_taps<C>(uniform_ptr(p.tex_planes + n * p.tex_stride[0]), t, gl, buf);          // in flight during the geometry MLP
        mlp_branch<C, HID, SPLIT>(s_geo, fg, og);

        // --- compositing weights (before the texture MLP: the geometry outputs die here) ---
        float sigma = __shfl(og[0][0], j);                    // feature 0 lives in lanes g = 0
        sigma += noise;
        const float dens = p.clamp_mode == 0 ? softplus_fast(sigma) : fmaxf(sigma, 0.f);
        float dnorm;
        {
            const float dx = p.rays_d_cam[r * 3 + 0], dy = p.rays_d_cam[r * 3 + 1], dz = p.rays_d_cam[r * 3 + 2];
            dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
        }
        const float delta = (sc + 1 < S) ? (znext - z) * dnorm : 1e10f;
        const float alpha = live ? 1.0f - __expf(-delta * dens) : 0.f;
        const float fac = live ? (1.0f - alpha + 1e-10f) : 1.0f;
        float tot;
        const float excl = seg16_excl_prod(fac, tot);
        const float w = alpha * (carry * excl);
        carry *= tot;
        wsum += w; dsum += w * z;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) acc_g[mt * 4 + rr] += w * og[mt][rr];

        blend_taps<C>(buf, t, ft);
        to_matrix_lanes(ft);
        tile_taps(np, rp, zln, jln, t);                                         // next tile's geometry taps: during the texture MLP
        issue_taps<C>(uniform_ptr(p.geo_planes + np * p.geo_stride[0]), t, gl, buf);
        mlp_branch<C, HID, SPLIT>(s_tex, ft, ot);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) acc_t[mt * 4 + rr] += w * ot[mt][rr];

        if (s0n == 0) {
            // --- close the ray: reduce over the 16 samples held by lanes with equal g ---
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                wsum += __shfl_xor(wsum, off); dsum += __shfl_xor(dsum, off);
#pragma unroll
                for (int i = 0; i < 8; ++i) { acc_g[i] += __shfl_xor(acc_g[i], off); acc_t[i] += __shfl_xor(acc_t[i], off); }
            }
            if (j == 0) {
                // last_back needs the un-weighted features of the final sample; not supported in the fused
                // kernel (host guards), white_back / max_depth are.
                const float bg = p.white_back ? (1.0f - wsum) : 0.f;
                float* of = p.out_feat + (int64_t)n * nch * p.rays_per_img + r;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int idx = 16 * mt + 4 * g + rr;
                        if (idx < p.feat_ch) of[(int64_t)idx * p.rays_per_img] = acc_t[mt * 4 + rr] + bg;
                        if (idx >= 1 && idx <= p.seg_ch) of[(int64_t)(p.feat_ch + idx - 1) * p.rays_per_img] = acc_g[mt * 4 + rr] + bg;
                    }
                if (g == 0) {
                    if (p.out_depth) p.out_depth[ray] = dsum + ((p.max_depth != 0.f) ? (1.0f - wsum) * p.max_depth : 0.f);
                    if (p.out_wsum) p.out_wsum[ray] = wsum;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { acc_t[i] = 0.f; acc_g[i] = 0.f; }
            carry = 1.0f; wsum = 0.f; dsum = 0.f;
        }
        ray = rayn; n = nn; r = rn; s0 = s0n; have = haven;
    }
    if constexpr (SPLIT) __syncthreads();          // exclusive residency: nobody leaves while another wave of the workgroup still multiplies
}

// Where sample_voxel takes its points from: an [n, m, 3] array, or the extract_shapes.py lattice generated in registers.
struct PointsFromMemory {
    const float* pts;
    // `row`: global row (image * m + idx); `idx`: the row inside its image
    __device__ __forceinline__ void get(int64_t row, int64_t /*idx*/, float& x, float& y, float& z) const {
        x = pts[row * 3 + 0]; y = pts[row * 3 + 1]; z = pts[row * 3 + 2];
    }
};

// extract_shapes.py:74-96 + the 0.9 scale of :103, evaluated like the reference's fp32 tensor ops, one rounding each:
// i -> float, column 2 = i % N, column 1 = (i / N) % N, column 0 = ((i / N) / N) % N with *float* division (the
// reference does not floor: its y / x "indices" carry a fractional part), then (v * voxel_size + corner) * scale.
struct PointsFromLattice {
    ide3d_lattice lat;
    int64_t first;
    __device__ __forceinline__ void get(int64_t /*row*/, int64_t idx, float& x, float& y, float& z) const {
        const int64_t i = first + idx;
        const float fn = (float)lat.n, fi = (float)i;                       // int64 -> fp32, round to nearest even
        const float q = __fdiv_rn(fi, fn);
        const int64_t r2 = (i >> 32) == 0 ? (int64_t)((unsigned)i % (unsigned)lat.n) : i % lat.n;    // 32-bit remainder when it fits
        const float s2 = (float)r2, s1 = fmodf(q, fn), s0 = fmodf(__fdiv_rn(q, fn), fn);
        x = __fmul_rn(__fadd_rn(__fmul_rn(s0, lat.voxel_size), lat.corner[2]), lat.scale);
        y = __fmul_rn(__fadd_rn(__fmul_rn(s1, lat.voxel_size), lat.corner[1]), lat.scale);
        z = __fmul_rn(__fadd_rn(__fmul_rn(s2, lat.voxel_size), lat.corner[0]), lat.scale);
    }
};

template <class Src>
__global__ void lattice_points_kernel(Src src, int64_t count, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        float x, y, z;
        src.get(i, i, x, y, z);
        out[i * 3 + 0] = x; out[i * 3 + 1] = y; out[i * 3 + 2] = z;
    }
}

// sample_voxel: gathers + MLPs for arbitrary points, rows of [feat | seg | sigma] (or sigma only).
template <int C, int HID, class Src, bool SPLIT>
__global__ void __launch_bounds__(64 * RmWaves<SPLIT>::value, 2)
sample_voxel_kernel(ide3d_render_params p, const Src src, int64_t m, float* __restrict__ out, int64_t tiles_per_block) {
    using K = RmCfg<C, HID>;
    constexpr int NW = RmWaves<SPLIT>::value;
    rm_claim_half_simd<SPLIT>();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int MB = MlpBytes<C, HID, SPLIT>::value;
    unsigned char* s_geo = reinterpret_cast<unsigned char*>(lds);
    unsigned char* s_tex = s_geo + MB;
    float* s_stage = reinterpret_cast<float*>(s_geo + 2 * MB);                 // [NW waves][16 samples][width] row staging
    stage_branch<C, HID, SPLIT>(s_geo, p.geo_w0, p.geo_b0, p.geo_w1, p.geo_b1, 1 + p.seg_ch);
    stage_branch<C, HID, SPLIT>(s_tex, p.tex_w0, p.tex_b0, p.tex_w1, p.tex_b1, p.feat_ch);
    __syncthreads();
    const int lane = lane_id(), wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15, gl = gather_block(lane);
    const int width = p.feat_ch + p.seg_ch + 1;
    const int sH = (int)p.tex_stride[2], sW = (int)p.tex_stride[3];
    float* stage = s_stage + wid * 16 * width;
    const int64_t rows = (int64_t)p.n * m;
    const int64_t ntiles = cdiv64(rows, 16);
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t tile_begin = (int64_t)blk * tiles_per_block;
    int64_t tile_end = tile_begin + tiles_per_block;
    if (tile_end > ntiles) tile_end = ntiles;

    // Software pipeline over this wave's tiles (same scheme as render_rays_kernel): the texture taps of a tile fly during its geometry
    // MLP, the geometry taps of the next tile during its texture MLP.  Wave-uniform per tile: the
    // image n0 of its first row (scalar plane bases); a lane whose row lies `dn` images further on (a tile straddling images: m not a
    // multiple of 16) adds dn image strides to its byte offsets.
    int64_t tile = tile_begin + wid;
    int n0 = 0;
    int64_t img_end = m;                                   // first row of image n0 + 1
    auto advance_image = [&](int64_t row0) { while (row0 >= img_end && n0 + 1 < p.n) { ++n0; img_end += m; } };
    // the point of row row0 + lane / 4 (gather layout) and how many images past n0 it lies
    auto tile_point = [&](int64_t row0, float& wx, float& wy, float& wz, int& dn) {
        const int64_t rl = row0 + gather_sample(lane);
        const int64_t rc = rl < rows ? rl : rows - 1;
        dn = 0;
        int64_t e = img_end;
        while (rc >= e) { ++dn; e += m; }
        src.get(rc, rc - (e - m), wx, wy, wz);
    };
    auto tile_taps = [&](float wx, float wy, float wz, TapAddr (&t)[3]) {
        t[0] = make_tap_addr(wx, wy, p.W, p.H, sH, sW);
        t[1] = make_tap_addr(wy, wz, p.W, p.H, sH, sW);
        t[2] = make_tap_addr(wx, wz, p.W, p.H, sH, sW);
    };
    const unsigned geo_img_b = (unsigned)p.geo_stride[0] * 4u, tex_img_b = (unsigned)p.tex_stride[0] * 4u;

    bool have = tile < tile_end;
    TapAddr t[3];
    TapBuf<C> buf;
    int dn = 0;
    if (have) {
        float wx, wy, wz;
        advance_image(tile * 16);
        tile_point(tile * 16, wx, wy, wz, dn);
        tile_taps(wx, wy, wz, t);
        issue_taps<C>(uniform_ptr(p.geo_planes + n0 * p.geo_stride[0]), t, gl, buf, (unsigned)dn * geo_img_b);
    }
    while (have) {
        const int64_t row0 = tile * 16;
        const int64_t tilen = tile + NW;
        const bool haven = tilen < tile_end;
        const int n0c = n0, dnc = dn;
        // next tile's points first (PointsFromMemory: three small loads, ahead of the taps that fly across the MLPs).  The prefetch is
        // unconditional (a conditional definition of the loop-carried tap buffer costs a register copy per value and iteration): past
        // the last tile it re-reads this one.
        const int64_t tilep = haven ? tilen : tile;
        float wxn, wyn, wzn;
        advance_image(tilep * 16);
        tile_point(tilep * 16, wxn, wyn, wzn, dn);
        float fg[K::NF];
        blend_taps<C>(buf, t, fg);
        to_matrix_lanes(fg);
        issue_taps<C>(uniform_ptr(p.tex_planes + n0c * p.tex_stride[0]), t, gl, buf, (unsigned)dnc * tex_img_b);
        f32x4 og[2];
        mlp_branch<C, HID, SPLIT>(s_geo, fg, og);
        float ft[K::NF];
        blend_taps<C>(buf, t, ft);
        to_matrix_lanes(ft);
        tile_taps(wxn, wyn, wzn, t);
        issue_taps<C>(uniform_ptr(p.geo_planes + n0 * p.geo_stride[0]), t, gl, buf, (unsigned)dn * geo_img_b);
        f32x4 ot[2];
        mlp_branch<C, HID, SPLIT>(s_tex, ft, ot);
        // stage the 16 x width row block, then write it out contiguously
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int idx = 16 * mt + 4 * g + rr;
                if (idx < p.feat_ch) stage[j * width + idx] = ot[mt][rr];
                if (idx >= 1 && idx <= p.seg_ch) stage[j * width + p.feat_ch + idx - 1] = og[mt][rr];
                if (idx == 0) stage[j * width + width - 1] = og[mt][rr];
            }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const int64_t live_rows = (rows - row0 < 16) ? rows - row0 : 16;
        const int64_t nel = live_rows * width;
        float* dst = out + row0 * width;
        for (int i = lane; i < nel; i += kWave) dst[i] = stage[i];
        __builtin_amdgcn_wave_barrier();
        tile = tilen; have = haven;
    }
    if constexpr (SPLIT) __syncthreads();          // exclusive residency (see render_rays_kernel)
}

// Densities only (extract_shapes.py's cube, sample_voxel(sigma_only)): one gather, the hidden layer, row 0 of the second layer.  The
// kernel is bound by its vector instructions (90 % VALU-busy before this form), a third of them the point and the tap addresses of
// the tile — which every lane of a quad computes alike.  So the points and taps are computed for 64 rows at once, lane = row (the
// `super tile`), and each of its four 16-row tiles fetches its own from the lanes that hold them (ds_bpermute: 25 values per tile
// instead of ~300 vector instructions).  Same software pipeline as above: the taps of the next tile fly during the MLP of this one.
template <int C, int HID, class Src, bool SPLIT>
__global__ void __launch_bounds__(64 * RmWaves<SPLIT>::value, 2)
density_kernel(ide3d_render_params p, const Src src, int64_t m, float* __restrict__ out_sigma, int64_t supers_per_block) {
    using K = RmCfg<C, HID>;
    constexpr int NW = RmWaves<SPLIT>::value;
    rm_claim_half_simd<SPLIT>();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned char* s_geo = reinterpret_cast<unsigned char*>(lds);
    float* const s_row = reinterpret_cast<float*>(s_geo + MlpBytes<C, HID, SPLIT>::value);     // row 0 of geo_w1 + its bias
    stage_branch<C, HID, SPLIT>(s_geo, p.geo_w0, p.geo_b0, p.geo_w1, p.geo_b1, 1 + p.seg_ch);
    for (int i = threadIdx.x; i <= HID; i += blockDim.x) s_row[i] = (i < HID) ? p.geo_w1[i] : p.geo_b1[0];
    __syncthreads();
    const int lane = lane_id(), wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15, gl = gather_block(lane);
    const int sH = (int)p.geo_stride[2], sW = (int)p.geo_stride[3];
    const int64_t rows = (int64_t)p.n * m;
    const int64_t nsuper = cdiv64(rows, 64);
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t super_begin = (int64_t)blk * supers_per_block;
    int64_t super_end = super_begin + supers_per_block;
    if (super_end > nsuper) super_end = nsuper;
    const unsigned geo_img_b = (unsigned)p.geo_stride[0] * 4u;

    int n0 = 0;                                            // image of the first row of the super tile whose taps `tl` holds
    int64_t img_end = m;
    // taps of row 64 st + lane, and how many images past n0 that row lies (a super tile that straddles images)
    auto super_taps = [&](int64_t st, TapAddr (&tl)[3], int& dnl) {
        const int64_t row0 = st * 64;
        while (row0 >= img_end && n0 + 1 < p.n) { ++n0; img_end += m; }
        const int64_t rl = row0 + lane;
        const int64_t rc = rl < rows ? rl : rows - 1;
        dnl = 0;
        int64_t e = img_end;
        while (rc >= e) { ++dnl; e += m; }
        float wx, wy, wz;
        src.get(rc, rc - (e - m), wx, wy, wz);
        tl[0] = make_tap_addr(wx, wy, p.W, p.H, sH, sW);
        tl[1] = make_tap_addr(wy, wz, p.W, p.H, sH, sW);
        tl[2] = make_tap_addr(wx, wz, p.W, p.H, sH, sW);
    };
    // tile k of the super tile, gather layout: lane 4 j + g takes the taps of row 16 k + j from lane 16 k + j
    auto tile_taps = [&](const TapAddr (&tl)[3], int dnl, int k, TapAddr (&t)[3], int& dn) {
        const int src_b = (16 * k + gather_sample(lane)) << 2;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            t[pl].o00 = __builtin_amdgcn_ds_bpermute(src_b, tl[pl].o00); t[pl].o01 = __builtin_amdgcn_ds_bpermute(src_b, tl[pl].o01);
            t[pl].o10 = __builtin_amdgcn_ds_bpermute(src_b, tl[pl].o10); t[pl].o11 = __builtin_amdgcn_ds_bpermute(src_b, tl[pl].o11);
            t[pl].w00 = __int_as_float(__builtin_amdgcn_ds_bpermute(src_b, __float_as_int(tl[pl].w00)));
            t[pl].w01 = __int_as_float(__builtin_amdgcn_ds_bpermute(src_b, __float_as_int(tl[pl].w01)));
            t[pl].w10 = __int_as_float(__builtin_amdgcn_ds_bpermute(src_b, __float_as_int(tl[pl].w10)));
            t[pl].w11 = __int_as_float(__builtin_amdgcn_ds_bpermute(src_b, __float_as_int(tl[pl].w11)));
        }
        dn = __builtin_amdgcn_ds_bpermute(src_b, dnl);
    };

    int64_t st = super_begin + wid;
    if (st < super_end) {
    TapAddr tl[3], t[3];
    TapBuf<C> buf;
    int dnl, dn;
    super_taps(st, tl, dnl);
    tile_taps(tl, dnl, 0, t, dn);
    issue_taps<C>(uniform_ptr(p.geo_planes + n0 * p.geo_stride[0]), t, gl, buf, (unsigned)dn * geo_img_b);
    while (st < super_end) {
        const int64_t stn = st + NW;
        const int64_t stp = stn < super_end ? stn : st;              // the prefetch past the last super tile re-reads this one
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t row = st * 64 + 16 * k + j;
            float fg[K::NF];
            blend_taps<C>(buf, t, fg);
            to_matrix_lanes(fg);
            if (k == 3) super_taps(stp, tl, dnl);
            tile_taps(tl, dnl, (k + 1) & 3, t, dn);
            issue_taps<C>(uniform_ptr(p.geo_planes + n0 * p.geo_stride[0]), t, gl, buf, (unsigned)dn * geo_img_b);
            float sig;
            if constexpr (SPLIT) sig = mlp_sigma_split<C, HID>(s_geo, s_row, fg);
            else sig = mlp_sigma<C, HID>(reinterpret_cast<const float*>(s_geo), s_row, fg);
            if (g == 0 && row < rows) out_sigma[row] = sig;
        }
        st = stn;
    }
    }
    if constexpr (SPLIT) __syncthreads();          // exclusive residency (see render_rays_kernel)
}

template <int C, int HID, bool SPLIT = false>
static int launch_render(const ide3d_render_params& p, hipStream_t st) {
    const size_t lds_bytes = (size_t)2 * MlpBytes<C, HID, SPLIT>::value;
    const int64_t total_rays = (int64_t)p.n * p.rays_per_img;
    constexpr int NW = RmWaves<SPLIT>::value;
    int64_t nblk = kNumCU * 8 / NW;                                // 8 waves per CU either way
    int64_t rpb = cdiv64(cdiv64(total_rays, nblk), NW) * NW;
    if (rpb < NW) rpb = NW;
    nblk = cdiv64(total_rays, rpb);
    auto kern = render_rays_kernel<C, HID, SPLIT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if constexpr (SPLIT) IDE3D_EXCL_LAUNCH(kern, dim3((unsigned)nblk), 64 * NW, lds_bytes, st, p, rpb);
    else hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * NW), lds_bytes, st, p, rpb);
    IDE3D_CHECK_LAUNCH("render_rays");
    return IDE3D_OK;
}

template <int C, int HID, class Src, bool SPLIT = false>
static int launch_voxel(const ide3d_render_params& p, const Src& src, int64_t m, float* out, float* out_sigma,
                        int sigma_only, hipStream_t st) {
    constexpr int NW = RmWaves<SPLIT>::value;
    // a lane whose row lies in a later image than the first row of its (super) tile adds whole image strides to its 32-bit byte
    // offsets: at most rows_per_tile - 1 images when m = 1
    {
        const int64_t rows_per_tile = sigma_only ? 64 : 16;
        const int64_t max_dn = (p.n > 1) ? ((rows_per_tile - 1) / (m > 0 ? m : 1) + 1 < p.n - 1 ? (rows_per_tile - 1) / (m > 0 ? m : 1) + 1 : p.n - 1) : 0;
        const int64_t img = p.geo_stride[0] > p.tex_stride[0] ? p.geo_stride[0] : p.tex_stride[0];
        if ((max_dn * img + img) * 4 >= 0xffffffffLL) { set_error("sample_voxel: too few points per image for planes this large"); return IDE3D_ENOKERNEL; }
    }
    if (sigma_only) {
        const size_t lds_bytes = (size_t)MlpBytes<C, HID, SPLIT>::value + (size_t)(HID + 4) * sizeof(float);
        const int64_t nsuper = cdiv64((int64_t)p.n * m, 64);
        int64_t nblk = kNumCU * 8 / NW;
        int64_t spb = cdiv64(cdiv64(nsuper, nblk), NW) * NW;
        if (spb < NW) spb = NW;
        nblk = cdiv64(nsuper, spb);
        auto kern = density_kernel<C, HID, Src, SPLIT>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if constexpr (SPLIT) IDE3D_EXCL_LAUNCH(kern, dim3((unsigned)nblk), 64 * NW, lds_bytes, st, p, src, m, out_sigma, spb);
        else hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * NW), lds_bytes, st, p, src, m, out_sigma, spb);
        IDE3D_CHECK_LAUNCH("sample_voxel (densities)");
        return IDE3D_OK;
    }
    const int width = p.feat_ch + p.seg_ch + 1;
    const size_t lds_bytes = (size_t)2 * MlpBytes<C, HID, SPLIT>::value + (size_t)NW * 16 * width * sizeof(float);
    const int64_t ntiles = cdiv64((int64_t)p.n * m, 16);
    int64_t nblk = kNumCU * 8 / NW;
    int64_t tpb = cdiv64(cdiv64(ntiles, nblk), NW) * NW;
    if (tpb < NW) tpb = NW;
    nblk = cdiv64(ntiles, tpb);
    auto kern = sample_voxel_kernel<C, HID, Src, SPLIT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if constexpr (SPLIT) IDE3D_EXCL_LAUNCH(kern, dim3((unsigned)nblk), 64 * NW, lds_bytes, st, p, src, m, out, tpb);
    else hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * NW), lds_bytes, st, p, src, m, out, tpb);
    IDE3D_CHECK_LAUNCH("sample_voxel");
    return IDE3D_OK;
}

static int check_render_params(const ide3d_render_params& p, const char* who, bool need_rays) {
    IDE3D_CHECK_ARG(p.tex_planes && p.geo_planes, "%s: null tri-plane pointer", who);
    IDE3D_CHECK_ARG(p.geo_w0 && p.geo_b0 && p.geo_w1 && p.geo_b1 && p.tex_w0 && p.tex_b0 && p.tex_w1 && p.tex_b1,
                    "%s: null MLP weight pointer", who);
    IDE3D_CHECK_ARG(p.n > 0 && p.C > 0 && p.H > 0 && p.W > 0, "%s: bad tri-plane shape", who);
    IDE3D_CHECK_ARG(p.feat_ch >= 1 && p.feat_ch <= 32 && p.seg_ch >= 0 && p.seg_ch <= 31,
                    "%s: feat_ch <= 32 and seg_ch <= 31 required", who);
    if (need_rays) {
        IDE3D_CHECK_ARG(p.rays_d_cam && p.z_lin && p.cam2world && p.out_feat, "%s: null ray / output pointer", who);
        IDE3D_CHECK_ARG(p.rays_per_img > 0 && p.steps > 0, "%s: bad ray shape", who);
        IDE3D_CHECK_ARG(p.clamp_mode == 0 || p.clamp_mode == 1, "%s: Need to choose clamp mode", who);
    }
    return IDE3D_OK;
}

static bool planes_fast(const ide3d_render_params& p) {
    auto ok = [&](const float* base, const int64_t* s) {
        return s[1] == 1 && (s[0] % 4 == 0) && (s[2] % 4 == 0) && (s[3] % 4 == 0) &&
               ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
    };
    return ok(p.tex_planes, p.tex_stride) && ok(p.geo_planes, p.geo_stride) &&
           p.tex_stride[2] == p.geo_stride[2] && p.tex_stride[3] == p.geo_stride[3] &&
           (p.tex_stride[2] * p.H + p.tex_stride[3] * p.W + 3 * p.C) * 4 < 0x7fffffffLL;       // byte offsets inside an image: 31 bits (launch_voxel bounds the image strides a straddling tile adds)
}

}  // namespace ide3d

// The decoder MLPs follow the arithmetic selected for the convolutions (ide3d_set_conv_arithmetic, modconv.hip): exact fp32 products on
// the fp32 matrix path by default, bf16x6 (fp32-grade) with any of the split arithmetics.
static bool mlp_split_selected() { return ide3d_get_conv_arithmetic() != 1; }

extern "C" int ide3d_render_rays(const ide3d_render_params* pp, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(pp != nullptr, "render_rays: null params");
    const ide3d_render_params& p = *pp;
    int rc = check_render_params(p, "render_rays", true);
    if (rc) return rc;
    if (p.last_back) { set_error("render_rays: last_back is not fused; use the step-wise ops"); return IDE3D_ENOKERNEL; }
    if (!planes_fast(p)) { set_error("render_rays: tri-planes must be channels_last, 16-byte aligned"); return IDE3D_ENOKERNEL; }
    hipStream_t st = (hipStream_t)stream;
    if (p.C == 32 && p.hidden == 64) return mlp_split_selected() ? launch_render<32, 64, true>(p, st) : launch_render<32, 64>(p, st);
    if (p.C == 16 && p.hidden == 32) return launch_render<16, 32>(p, st);
    set_error("render_rays: no fused kernel for C=%d hidden=%d", p.C, p.hidden);
    return IDE3D_ENOKERNEL;
}

extern "C" int ide3d_sample_voxel(const ide3d_render_params* pp, const float* pts, int64_t m,
                                  float* out, float* out_sigma, int sigma_only, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(pp != nullptr && pts != nullptr, "sample_voxel: null params");
    const ide3d_render_params& p = *pp;
    int rc = check_render_params(p, "sample_voxel", false);
    if (rc) return rc;
    IDE3D_CHECK_ARG(m >= 0, "sample_voxel: bad point count");
    IDE3D_CHECK_ARG(sigma_only ? out_sigma != nullptr : out != nullptr, "sample_voxel: null output");
    if (m == 0) return IDE3D_OK;
    if (!planes_fast(p)) { set_error("sample_voxel: tri-planes must be channels_last, 16-byte aligned"); return IDE3D_ENOKERNEL; }
    hipStream_t st = (hipStream_t)stream;
    const PointsFromMemory src{pts};
    if (p.C == 32 && p.hidden == 64) return mlp_split_selected() ? launch_voxel<32, 64, PointsFromMemory, true>(p, src, m, out, out_sigma, sigma_only, st)
                                                                  : launch_voxel<32, 64>(p, src, m, out, out_sigma, sigma_only, st);
    if (p.C == 16 && p.hidden == 32) return launch_voxel<16, 32>(p, src, m, out, out_sigma, sigma_only, st);
    set_error("sample_voxel: no fused kernel for C=%d hidden=%d", p.C, p.hidden);
    return IDE3D_ENOKERNEL;
}

static int check_lattice(const ide3d_lattice* lat, int64_t first, int64_t count, const char* what) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(lat != nullptr && lat->n >= 2 && lat->n <= 2048, "%s: lattice resolution must be in [2, 2048]", what);
    const int64_t total = (int64_t)lat->n * lat->n * lat->n;
    IDE3D_CHECK_ARG(first >= 0 && count >= 0 && first + count <= total, "%s: point range outside the lattice", what);
    return IDE3D_OK;
}

extern "C" int ide3d_lattice_points(const ide3d_lattice* lat, int64_t first, int64_t count, float* pts, void* stream) {
    using namespace ide3d;
    int rc = check_lattice(lat, first, count, "lattice_points");
    if (rc) return rc;
    IDE3D_CHECK_ARG(pts != nullptr || count == 0, "lattice_points: null output");
    if (count == 0) return IDE3D_OK;
    const PointsFromLattice src{*lat, first};
    hipLaunchKernelGGL(lattice_points_kernel<PointsFromLattice>, dim3(stream_grid(count, 256)), dim3(256), 0, (hipStream_t)stream,
                       src, count, pts);
    IDE3D_CHECK_LAUNCH("lattice_points");
    return IDE3D_OK;
}

extern "C" int ide3d_density_lattice(const ide3d_render_params* pp, const ide3d_lattice* lat, int64_t first, int64_t count,
                                     float* out_sigma, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(pp != nullptr, "density_lattice: null params");
    const ide3d_render_params& p = *pp;
    int rc = check_render_params(p, "density_lattice", false);
    if (rc) return rc;
    rc = check_lattice(lat, first, count, "density_lattice");
    if (rc) return rc;
    IDE3D_CHECK_ARG(out_sigma != nullptr || count == 0, "density_lattice: null output");
    if (count == 0) return IDE3D_OK;
    if (!planes_fast(p)) { set_error("density_lattice: tri-planes must be channels_last, 16-byte aligned"); return IDE3D_ENOKERNEL; }
    hipStream_t st = (hipStream_t)stream;
    const PointsFromLattice src{*lat, first};
    if (p.C == 32 && p.hidden == 64) return mlp_split_selected() ? launch_voxel<32, 64, PointsFromLattice, true>(p, src, count, nullptr, out_sigma, 1, st)
                                                                  : launch_voxel<32, 64>(p, src, count, nullptr, out_sigma, 1, st);
    if (p.C == 16 && p.hidden == 32) return launch_voxel<16, 32>(p, src, count, nullptr, out_sigma, 1, st);
    set_error("density_lattice: no fused kernel for C=%d hidden=%d", p.C, p.hidden);
    return IDE3D_ENOKERNEL;
}

namespace ide3d {
const char* raymarch_build_flags() {
    return ""
#ifdef IDE3D_SPLIT_NO_DOT2
        "IDE3D_SPLIT_NO_DOT2 "
#endif
        ;
}
}  // namespace ide3d
