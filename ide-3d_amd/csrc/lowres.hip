// lowres.hip — the low-resolution block group of the tri-plane backbone (4^2 .. 16^2 / 32^2) in ONE launch.
//
// Replaces, for the StyleGAN2 dual-path blocks of inversion/networks.py:966-1139 whose maps are so small that every per-layer kernel of
// modconv.hip / upfirdn2d.hip is a latency chain ("persistent low-resolution backbone"):
//   SynthesisLayer up = 1 (networks.py:330-514): modulated 3x3 conv + noise + bias + lrelu                     -> phases G, R
//   SynthesisLayer up = 2: transposed 3x3 stride-2 conv, 4x4 FIR (conv2d_resample.py:112-129), noise, bias, lrelu -> phases G, R (FIR inside R)
//   ToRGBLayer x 2 (torgb + toseg, networks.py:670-713, :1109,1130) + `img = upsample2d(img) + y` (:1100,1121)    -> side work H
// The per-layer path costs 2-4 launches per layer + 3 per head (26 launches for 4^2 .. 32^2 at batch 1, 0.64 ms of a 1.67 ms pass for ~3 % of its
// FLOPs): each launch pays its own fill, prologue (weights + patch round trip), split-K partial round trip and drain.
//
// Structure (one 8-wave workgroup per CU, 160 KB LDS each, a grid barrier between phases):
//   * GEMM view per layer: D[pixel, cout] = sum_{tap, cin} X[pixel + tap, cin] * W[tap, cin, cout] on v_mfma_f32_32x32x16_bf16 with every fp32
//     operand as PARTS bf16 pieces (bf16x6 / bf16x3) — M = pixels of ALL images (<= ~512), N = 32 output channels, K = 32 input
//     channels x 9 taps per work item: item (cb, s) = (output-channel block, K slice), C/32 x C/32 = 256 items at C = 512.
//   * WEIGHT-STATIONARY: an item's whole weight slice (9 x PARTS x 4 x 32 units of 16 B = 55 KB, pre-split and packed once per weight version in
//     exactly the LDS image) is copied to LDS in one burst; in the persistent form the NEXT layer's slice is requested while the workgroup
//     sits in the reduction phase / the grid barrier (it depends on nothing computed here).
//   * Activations travel between layers PRE-MODULATED by the consumer's styles and PRE-SPLIT into bf16 pieces, channel-slice major with a zero
//     halo: act[n][slice][piece][k octet][(res + 2)^2] units of 16 B (8 channels of one pixel): an item's input is one contiguous run per image
//     and, in LDS, the A operand of the MFMA is one ds_read_b128 per (tap, piece, k half) at slot + tap offset — no bounds checks, no conversion
//     in the loop (modconv.hip spends 4 of its 8 waves' issue slots on modulate + split + commit per chunk).
//   * split-K partials: slab[s][cb][pixel][32] fp32, written coalesced from the accumulator layout (lane = cout, 128 B per pixel row);
//     phase R: item (image, cb, row band) sums the C/32 slabs in slice order (deterministic), applies the 4x4 FIR for up layers (the
//     (2h + 1)^2 transposed result never exists in fp32 NCHW), demodulation, noise, bias, lrelu, gain, clamp, and writes what the consumers need:
//     the next layer's act (x its styles, split), the fp32 pixel-major copy the heads read, and / or the NCHW tensor that leaves the group.
//   * heads: per-image folded 1x1 weights [n, O, C] (style.hip), plain fp32 FMAs (exact products, like head_small_kernel): a wave owns 8 pixels x
//     8 outputs, lanes split the channels, the 64 partial sums meet in a 6-step transposing butterfly (63 shuffles instead of 64 x 6); bias,
//     clamp and `+ upsample2d(previous skip)` (4 taps of the 2 x 2 polyphase) are applied by the lane that ends up with the sum.  Run as side
//     work behind the GEMM of the next layer's G phase (same dependency, off the critical path).
// Two launch forms of the SAME phase functions: `persistent = 1` one cooperative-style launch (co-residency checked against the occupancy
// query, bounded spins), `persistent = 0` one launch per phase (what MI355X_MICROARCH.md's price list favours for GEMM -> GEMM seams: a kernel
// boundary is 1.5-1.9 us, a 256-workgroup barrier 4-7 us).  Both are measured in profiles/round6/lowres_ab.txt.
#include "common.h"
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <utility>

namespace ide3d {
namespace lr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int THREADS = 512, WAVES = 8;
constexpr int KS = 32, BC = 32, KOCT = KS / 8;                 // channels per K slice, output channels per block, 16-byte units per pixel and piece
constexpr int LDS_BYTES = 160 * 1024, TAB_BYTES = 4096;
constexpr int MAX_LAYERS = IDE3D_LOWRES_MAX_LAYERS, MAX_HEADS = IDE3D_LOWRES_MAX_HEADS, MAX_PHASES = 2 * MAX_LAYERS + 3;
__host__ __device__ constexpr int w_units(int parts) { return 9 * parts * KOCT * BC; }
__host__ __device__ constexpr int img_pad(int parts, int nslot) { return (parts * KOCT * nslot + 63) & ~63; }
__host__ __device__ constexpr int act_capacity_units(int parts) { return (LDS_BYTES - TAB_BYTES) / 16 - w_units(parts); }

enum { PH_G = 1, PH_R = 2, PH_H = 3 };

struct LayerDev {
    const u32x4* wq;            // [items][w_units]
    const u32x4* act_in;        // [n][S][PARTS * KOCT][nslot_in]
    u32x4* act_out;             // the next layer's act_in, or null
    const float* dcoefs; const float* noise; const float* bias;
    const float* styles;        // this layer's styles [n, C] (phase PREP modulates the group's input with them)
    const float* styles_next;   // the next layer's (phase R modulates this layer's output with them), or null
    float* xf;                  // [n][res^2][C] fp32 (what the heads read), or null
    float* x_out;               // [n][C][res][res], or null
    float act_gain, clamp;
    int up, res, rin;           // output / input resolution
    int nslot_in, nslot_out;    // (rin + 2)^2, (res + 2)^2
    int npos, outpix;           // GEMM rows per image: res^2 | (rin + 1)^2; partial pixels per image: res^2 | (2 rin + 1)^2
    int bands;                  // row bands per (image, channel block) in phase R
    int img_pad_in, img_pad_out; // units per (image, K slice) of act_in / act_out: PARTS * KOCT * nslot rounded up to whole 1 KB chunks
    int wpt;                    // waves per GEMM row tile (waves_per_tile): the layer has S * wpt partial slabs
    int first;                  // the group's first layer: phase G builds its input from x0
    int head_side;              // head computed as side work of this layer's G phase, or -1
};
struct HeadDev {
    const float* w; const float* bias; const float* xf; const float* skip_prev; float* skip;
    float clamp; int O, res;
};
struct Phase { int kind, idx; };
struct GroupDev {
    LayerDev L[MAX_LAYERS];
    HeadDev H[MAX_HEADS];
    Phase ph[MAX_PHASES];
    const float* x0; long long x0_bs;
    const float* fir;           // [4][4] resample filter (device)
    float* partial;
    unsigned* counter;          // [0] barrier, [1] error word
    int n, C, S, CB, nlayers, nheads, nphases, res0;
};

// ---- small helpers --------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// a, b -> PARTS packed bf16 pairs whose sums reproduce a and b (round to nearest at each step, residuals exact in fp32): modconv.hip split_pair
template <int PARTS>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&out)[PARTS]) {
#pragma unroll
    for (int q = 0; q < PARTS; ++q) {
        const unsigned pk = pk_bf16(a, b);
        out[q] = pk;
        if (q + 1 < PARTS) { a -= __uint_as_float(pk << 16); b -= __uint_as_float(pk & 0xffff0000u); }
    }
}
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
__device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- weight packing: w [C, C, 3, 3] -> [item = cb * S + s][tap][piece][k octet][co 32] units of 8 input channels ----------------
template <int PARTS>
__global__ void __launch_bounds__(256)
pack_kernel(const float* __restrict__ w, int C, u32x4* __restrict__ out) {
    const int S = C / KS, CB = C / BC;
    const long long total = (long long)CB * S * 9 * KOCT * BC;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long r = i;
        const int co_l = (int)(r % BC); r /= BC;
        const int ko = (int)(r % KOCT); r /= KOCT;
        const int tap = (int)(r % 9); r /= 9;
        const int s = (int)(r % S); const int cb = (int)(r / S);
        const int co = cb * BC + co_l, ci0 = s * KS + ko * 8;
        unsigned pk[4][PARTS];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            split_pair<PARTS>(w[((long long)co * C + ci0 + 2 * e) * 9 + tap], w[((long long)co * C + ci0 + 2 * e + 1) * 9 + tap], pk[e]);
        const long long base = ((long long)(cb * S + s) * 9 + tap) * PARTS * KOCT * BC + ko * BC + co_l;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) {
            const u32x4 v = {pk[0][q], pk[1][q], pk[2][q], pk[3][q]};
            out[base + (long long)q * KOCT * BC] = v;
        }
    }
}

// ---- grid barrier (mapping.hip's, MI355X_MICROARCH.md "Workgroup dispatch ..."): monotonic counter, release -> relaxed arrive, relaxed poll
// with s_sleep, one acquire; bounded spin that traps with the error word set ----------------------------------------------------
constexpr unsigned SPIN_LIMIT = 1u << 22;
template <class F>
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, F&& between) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    between();
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { __hip_atomic_store(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_trap(); }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// ---- staging --------------------------------------------------------------------------------------------------------------
// LDS-DMA (global_load_lds, 16 B per lane): a wave copies 64 consecutive 16-byte units = 1 KB to a wave-uniform LDS address; every copy of a
// phase is in flight at once, no registers, no ds_write (the first version staged through registers, 4 loads per thread in flight: a
// batch-4 G phase spent 12 dependent round trips on its activations)
__device__ __forceinline__ void dma64(const u32x4* __restrict__ src_chunk, u32x4* dst_chunk, int lane) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_chunk + lane),
                                     (__attribute__((address_space(3))) void*)dst_chunk, 16, 0, 0);
}
template <int PARTS>
__device__ __forceinline__ void stage_weights(const LayerDev& L, int item, unsigned char* smem) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    constexpr int CH = w_units(PARTS) / 64;
    static_assert(w_units(PARTS) % 64 == 0, "weight slice = whole 1 KB chunks");
    const u32x4* src = L.wq + (long long)item * w_units(PARTS);
    u32x4* dst = reinterpret_cast<u32x4*>(smem);
    for (int c = wid; c < CH; c += WAVES) dma64(src + c * 64, dst + c * 64, lane);
}

// ---- phase G: one (cb, s) item ------------------------------------------------------------------------------------------------
// (Round 6, measured and removed: dealing a row tile's 6 (k step, kernel row) groups to 2 / 3 / 6 waves when a layer has fewer than 5 tiles, each wave
// writing its own partial slab.  The G phase did not get shorter — it is a latency chain of launch, weight DMA, one MFMA burst, stores — and the
// R phase, which then sums S x wpt slabs in batches of 8 dependent round trips, got 5 us longer: profiles/round6/lowres_ab.txt.)
__host__ __device__ inline int waves_per_tile(int) { return 1; }

template <int PARTS, int UP>
__device__ __forceinline__ void g_item(const GroupDev& g, const LayerDev& L, int item, unsigned char* smem, bool weights_ready) {
    constexpr int NACC = UP ? 4 : 2;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, khalf = lane >> 5;
    const int cb = item / g.S, s = item - cb * g.S;
    u32x4* const s_w = reinterpret_cast<u32x4*>(smem);
    u32x4* const s_a = s_w + w_units(PARTS);
    int* const s_tab = reinterpret_cast<int*>(smem + LDS_BYTES - TAB_BYTES);
    const int nslot = L.nslot_in, img_pad = L.img_pad_in;
    const int R = L.res, rin = L.rin, G = rin + 1, T = 2 * rin + 1, pitch = rin + 2;
    if (!weights_ready) stage_weights<PARTS>(L, item, smem);
    if (L.first) {
        // the group's input x0 [C, rin, rin] (+ n * x0_bs) x this layer's styles, split, straight into the LDS image (halo zeroed first)
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int i = tid; i < g.n * img_pad; i += THREADS) s_a[i] = z;
        __syncthreads();
        const int hw = rin * rin;
        for (int e = tid; e < g.n * hw * KOCT; e += THREADS) {
            const int pix = e % hw, r2 = e / hw, ko = r2 % KOCT, n = r2 / KOCT;
            const int c0 = s * KS + ko * 8;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = g.x0[n * g.x0_bs + (long long)(c0 + j) * hw + pix] * L.styles[(long long)n * g.C + c0 + j];
            unsigned pk[4][PARTS];
#pragma unroll
            for (int j = 0; j < 4; ++j) split_pair<PARTS>(v[2 * j], v[2 * j + 1], pk[j]);
            const int slot = (pix / rin + 1) * pitch + pix % rin + 1;
#pragma unroll
            for (int q = 0; q < PARTS; ++q) {
                const u32x4 u = {pk[0][q], pk[1][q], pk[2][q], pk[3][q]};
                s_a[n * img_pad + (q * KOCT + ko) * nslot + slot] = u;
            }
        }
    } else {
        // activations: per image one contiguous run [piece][k octet][slot] of this K slice (padded to whole 1 KB chunks)
        const int ich = img_pad >> 6;
        for (int i = wid; i < g.n * ich; i += WAVES) {
            const int n = i / ich, c = i - n * ich;
            dma64(L.act_in + ((long long)n * g.S + s) * img_pad + c * 64, s_a + n * img_pad + c * 64, lane);
        }
    }
    const int mpix = g.n * L.npos;
    const int mtiles = (mpix + 31) >> 5;
    if (UP) {
        // output index of class (0, 0) of every GEMM row + edge flags: the last position row / column only has the even class
        for (int p = tid; p < mtiles * 32; p += THREADS) {
            int e = 1 << 26;
            if (p < mpix) {
                const int n = p / L.npos, rem = p - n * L.npos, gy = rem / G, gx = rem - gy * G;
                e = (n * L.outpix + 2 * gy * T + 2 * gx) | ((gx == rin) << 24) | ((gy == rin) << 25);
            }
            s_tab[p] = e;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int wpt = L.wpt;
    const int sub = wpt > 1 ? wid % wpt : 0;
    float* const slab = g.partial + ((long long)((s * wpt + sub) * g.CB + cb) * g.n * L.outpix) * BC;
    for (int mt = (wpt > 1 ? wid / wpt : wid); mt < mtiles; mt += (wpt > 1 ? WAVES : WAVES)) {
        if (wpt > 1 && wid >= mtiles * wpt) break;
        const int p = min(mt * 32 + l32, mpix - 1);
        const int n = p / L.npos, rem = p - n * L.npos;
        int base;
        if (UP) { const int gy = rem / G, gx = rem - gy * G; base = n * img_pad + gy * pitch + gx; }
        else    { const int y = rem / R, x = rem - y * R;    base = n * img_pad + y * pitch + x; }
        f32x16 acc[NACC];
#pragma unroll
        for (int c = 0; c < NACC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        // 6 groups (k step, kernel row) of 3 taps: a group's 6 PARTS operand reads, then its products; scheduling barriers between groups keep
        // hipcc from hoisting all 108 ds_read_b128 of the unrolled loop in front of the first product (it did: 100 registers spilled).
        // Independent accumulators: the four output classes of the transposed form; two alternating sets for the plain convolution (a 32x32x16
        // MFMA that accumulates into the previous one's result waits out its 16 passes)
        static_for<6>([&](auto gg) {
            constexpr int gi = decltype(gg)::value, ks = gi / 3, ky = gi % 3;
            if (wpt == 1 || gi % wpt == sub) {
                const int ko = 2 * ks + khalf;
                u32x4 a[3][PARTS], b[3][PARTS];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int tap = ky * 3 + kx;
                    const int aoff = UP ? ((ky == 2 ? 0 : 1) * pitch + (kx == 2 ? 0 : 1)) : (ky * pitch + kx);
#pragma unroll
                    for (int q = 0; q < PARTS; ++q) {
                        b[kx][q] = s_w[((tap * PARTS + q) * KOCT + ko) * BC + l32];
                        a[kx][q] = s_a[base + (q * KOCT + ko) * nslot + aoff];
                    }
                }
                // products (qa, qb) with qa + qb < PARTS, grouped by max(qa, qb) like modconv.hip's split loop
#pragma unroll
                for (int Q = 0; Q < PARTS; ++Q)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int ai = UP ? ((ky & 1) * 2 + (kx & 1)) : (kx & 1);
#pragma unroll
                        for (int qa = 0; qa <= Q; ++qa)
#pragma unroll
                            for (int qb = 0; qb <= Q; ++qb)
                                if ((qa == Q || qb == Q) && qa + qb < PARTS) acc[ai] = mfma(a[kx][qa], b[kx][qb], acc[ai]);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (!UP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] += acc[1][r];
        }
        // accumulator layout: lane = output channel l32, register r = GEMM row 8 (r / 4) + 4 khalf + r % 4: 128-byte runs per row
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pi = mt * 32 + 8 * (r >> 2) + 4 * khalf + (r & 3);
            if (UP) {
                const int e = s_tab[pi];
                if (e & (1 << 26)) continue;
                const int idx = e & 0xffffff;
                const bool xe = (e >> 24) & 1, ye = (e >> 25) & 1;
                slab[(long long)idx * BC + l32] = acc[0][r];
                if (!xe) slab[(long long)(idx + 1) * BC + l32] = acc[1][r];
                if (!ye) slab[(long long)(idx + T) * BC + l32] = acc[2][r];
                if (!xe && !ye) slab[(long long)(idx + T + 1) * BC + l32] = acc[3][r];
            } else if (pi < mpix) slab[(long long)pi * BC + l32] = acc[0][r];
        }
    }
}

// ---- heads as side work: wave-item (image, 8 pixels, 8 outputs) ------------------------------------------------------------------
__device__ __forceinline__ void head_items(const GroupDev& g, const HeadDev& H, int first_wave, int wave_stride) {
    const int lane = threadIdx.x & 63;
    const int hw = H.res * H.res, PG = (hw + 7) / 8, OG = (H.O + 7) / 8;
    const int total = g.n * PG * OG;
    for (int wi = first_wave; wi < total; wi += wave_stride) {
        const int n = wi / (PG * OG), rem = wi - n * PG * OG, pg = rem / OG, og = rem - pg * OG;
        float acc[8][8];                                            // [output][pixel]
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[o][p] = 0.f;
        for (int c0 = 0; c0 < g.C; c0 += 512) {
            const int c = c0 + lane * 8;
            const bool live = c < g.C;
            const int cc = live ? c : 0;
            float4 xa[8], xb[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int pix = min(pg * 8 + p, hw - 1);
                const float* src = H.xf + ((long long)n * hw + pix) * g.C + cc;
                xa[p] = *reinterpret_cast<const float4*>(src); xb[p] = *reinterpret_cast<const float4*>(src + 4);
            }
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const int oo = min(og * 8 + o, H.O - 1);
                const float* src = H.w + ((long long)n * H.O + oo) * g.C + cc;
                float4 wa = *reinterpret_cast<const float4*>(src), wb = *reinterpret_cast<const float4*>(src + 4);
                if (!live) { wa = make_float4(0.f, 0.f, 0.f, 0.f); wb = wa; }
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    float v = acc[o][p];
                    v = fmaf(xa[p].x, wa.x, v); v = fmaf(xa[p].y, wa.y, v); v = fmaf(xa[p].z, wa.z, v); v = fmaf(xa[p].w, wa.w, v);
                    v = fmaf(xb[p].x, wb.x, v); v = fmaf(xb[p].y, wb.y, v); v = fmaf(xb[p].z, wb.z, v); v = fmaf(xb[p].w, wb.w, v);
                    acc[o][p] = v;
                }
            }
        }
        // transposing butterfly: 64 values per lane -> lane l holds the sum over all lanes of value l (= output l / 8, pixel l % 8); fixed order
        float v[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = acc[i >> 3][i & 7];
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            const int half = 32 >> st;
            const bool up = (lane & half) != 0;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const float keep = up ? v[i + half] : v[i], send = up ? v[i] : v[i + half];
                v[i] = keep + __shfl_xor(send, half);
            }
        }
        const int o = og * 8 + (lane >> 3), pix = pg * 8 + (lane & 7);
        if (o < H.O && pix < hw) {
            float r = v[0] + (H.bias ? H.bias[o] : 0.f);
            if (H.clamp >= 0.f) r = fminf(fmaxf(r, -H.clamp), H.clamp);
            if (H.skip_prev) {
                // upsample2d(prev, f): zero insertion x 2, pad (2, 1), 4x4 filter (flipped: true convolution) x gain 4: out[2m] takes taps 0, 2
                // of rows m - 1, m; out[2m + 1] taps 1, 3 of rows m, m + 1 (upfirdn2d.py:313-349)
                const int y = pix / H.res, x = pix - y * H.res, hp = H.res >> 1;
                const int my = y >> 1, mx = x >> 1, py = y & 1, px = x & 1;
                const float* prev = H.skip_prev + ((long long)n * H.O + o) * hp * hp;
                float up2 = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int ry = my - 1 + py + i, rx = mx - 1 + px + j;
                        const int ky = py + 2 * i, kx = px + 2 * j;
                        if (ry >= 0 && ry < hp && rx >= 0 && rx < hp) up2 = fmaf(prev[ry * hp + rx], g.fir[(3 - ky) * 4 + (3 - kx)] * 4.f, up2);
                    }
                r += up2;
            }
            H.skip[((long long)n * H.O + o) * hw + pix] = r;
        }
    }
}

// ---- phase R: item (image, channel block, row band) ----------------------------------------------------------------------------------
template <int PARTS>
__device__ __forceinline__ void r_item(const GroupDev& g, const LayerDev& L, int item, unsigned char* smem_r) {
    const int tid = threadIdx.x;
    const int per_n = g.CB * L.bands;
    const int n = item / per_n, cg = (item - n * per_n) / L.bands, band = item % L.bands;
    const int R = L.res;
    const int y0 = band * R / L.bands, y1 = (band + 1) * R / L.bands, npix = (y1 - y0) * R;
    const bool fir = L.up == 2;
    const int T = 2 * L.rin + 1, width = fir ? T : R;
    const int ry0 = fir ? max(y0 - 1, 0) : y0, ry1 = fir ? min(y1 + 2, T) : y1;
    const int npos = (ry1 - ry0) * width;
    float* const red = reinterpret_cast<float*>(smem_r);           // [npos][32]
    float* const outs = red + ((npos * BC + 3) & ~3);              // [npix][33]
    float* const par = outs + ((npix * 33 + 3) & ~3);              // demodulation, bias, next styles of the 32 channels; noise of the band's pixels
    // the small operands of the epilogue are requested FIRST and parked in LDS after the slab loads have been issued: four dependent
    // round trips (d, noise, b, styles) in front of every store otherwise
    float pv = 0.f;
    {
        const int c = tid & 31, k = tid >> 5;
        if (k == 0) pv = L.dcoefs[(long long)n * g.C + cg * BC + c];
        else if (k == 1) pv = L.bias ? L.bias[cg * BC + c] : 0.f;
        else if (k == 2) pv = L.styles_next ? L.styles_next[(long long)n * g.C + cg * BC + c] : 1.f;
    }
    float nz[2] = {0.f, 0.f};
    if (L.noise) {
        if (tid < npix) nz[0] = L.noise[y0 * R + tid];
        if (tid + THREADS < npix) nz[1] = L.noise[y0 * R + tid + THREADS];
    }
    const int nsplit = g.S * L.wpt;
    const long long slab_stride = (long long)g.CB * g.n * L.outpix * BC;            // floats between slabs
    const float* src0 = g.partial + ((long long)cg * g.n * L.outpix + (long long)n * L.outpix + ry0 * width) * BC;
    for (int e = tid; e < npos * 8; e += THREADS) {
        const float* src = src0 + e * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < nsplit; s0 += 8) {                   // 8 slabs in flight, summed in slab order (deterministic)
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(src + (long long)min(s0 + j, nsplit - 1) * slab_stride);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + j < nsplit) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
        }
        *reinterpret_cast<float4*>(red + e * 4) = acc;
    }
    if (tid < 96) par[tid] = pv;
    if (tid < npix) par[96 + tid] = nz[0];
    if (tid + THREADS < npix) par[96 + tid + THREADS] = nz[1];
    __syncthreads();
    {
        const float e_clamp = L.clamp >= 0.f ? L.clamp : __builtin_inff();
        float gf[16];
        if (fir) {
#pragma unroll
            for (int i = 0; i < 16; ++i) gf[i] = g.fir[15 - i] * 4.f;             // flipped (true convolution), gain 4 (conv2d_resample.py:125-126)
        }
        for (int e = tid; e < npix * BC; e += THREADS) {
            const int c = e & 31, pix = e >> 5;
            const int y = y0 + pix / R, x = pix % R;
            float v;
            if (fir) {
                v = 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int ty = y + a - 1;
                    if (ty < 0 || ty >= T) continue;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int tx = x + b - 1;
                        if (tx >= 0 && tx < T) v = fmaf(red[((ty - ry0) * T + tx) * BC + c], gf[a * 4 + b], v);
                    }
                }
            } else v = red[pix * BC + c];
            v = v * par[c] + par[96 + pix] + par[32 + c];
            v = (v > 0.f ? v : v * 0.2f) * L.act_gain;
            v = fminf(fmaxf(v, -e_clamp), e_clamp);
            outs[pix * 33 + c] = v;
        }
    }
    __syncthreads();
    // (a) the next layer's input: x its styles, PARTS bf16 pieces, slice-major with halo
    if (L.act_out) {
        const int nslot = L.nslot_out, pitch = R + 2;
        for (int e = tid; e < npix * KOCT; e += THREADS) {
            const int ko = e / npix, pix = e - ko * npix;
            const int y = y0 + pix / R, x = pix % R;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = outs[pix * 33 + ko * 8 + j] * par[64 + ko * 8 + j];
            unsigned pk[4][PARTS];
#pragma unroll
            for (int j = 0; j < 4; ++j) split_pair<PARTS>(v[2 * j], v[2 * j + 1], pk[j]);
            u32x4* dst = L.act_out + ((long long)n * g.S + cg) * L.img_pad_out + (long long)ko * nslot + (y + 1) * pitch + (x + 1);
#pragma unroll
            for (int q = 0; q < PARTS; ++q) {
                const u32x4 u = {pk[0][q], pk[1][q], pk[2][q], pk[3][q]};
                dst[(long long)q * KOCT * nslot] = u;
            }
        }
    }
    if (L.xf) {
        for (int e = tid; e < npix * BC; e += THREADS) {
            const int c = e & 31, pix = e >> 5;
            L.xf[((long long)n * R * R + y0 * R + pix) * g.C + cg * BC + c] = outs[pix * 33 + c];
        }
    }
    if (L.x_out) {
        for (int e = tid; e < npix * BC; e += THREADS) {
            const int c = e / npix, pix = e - c * npix;
            L.x_out[((long long)n * g.C + cg * BC + c) * R * R + y0 * R + pix] = outs[pix * 33 + c];
        }
    }
    __syncthreads();
}

template <int PARTS>
__device__ __forceinline__ void run_phase(const GroupDev& g, const Phase& ph, unsigned char* smem, int wg, int nwg, bool weights_ready) {
    const LayerDev& L = g.L[ph.idx < MAX_LAYERS ? ph.idx : 0];
    if (ph.kind == PH_G) {
        const int items = g.CB * g.S;
        for (int it = wg; it < items; it += nwg) {
            if (L.up == 2) g_item<PARTS, 1>(g, L, it, smem, weights_ready && it == wg);
            else g_item<PARTS, 0>(g, L, it, smem, weights_ready && it == wg);
            if (it + nwg < items) __syncthreads();
        }
        if (L.head_side >= 0) head_items(g, g.H[L.head_side], wg * WAVES + (threadIdx.x >> 6), nwg * WAVES);
    } else if (ph.kind == PH_H) {
        head_items(g, g.H[ph.idx], wg * WAVES + (threadIdx.x >> 6), nwg * WAVES);
    } else {
        const int items = g.n * g.CB * L.bands;
        // R scratch sits BEHIND the weight region: in the persistent form the next layer's weights land there meanwhile
        for (int it = wg; it < items; it += nwg) r_item<PARTS>(g, L, it, smem + w_units(PARTS) * 16);
    }
}

// one launch per phase
template <int PARTS>
__global__ void __launch_bounds__(THREADS, 2)
phase_kernel(const GroupDev g, int phase) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    asm volatile("" ::: "v255");                       // 8 waves x 256 registers: the workgroup owns its CU's register files
    run_phase<PARTS>(g, g.ph[phase], smem, blockIdx.x, gridDim.x, false);
    __syncthreads();                                   // no wave leaves while a partner on its SIMD still multiplies
}

// all phases in one launch: grid = min(items, co-resident workgroups); the next G phase's weight slice is requested inside the barrier
template <int PARTS>
__global__ void __launch_bounds__(THREADS, 2)
persistent_kernel(const GroupDev g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    asm volatile("" ::: "v255");
    const int wg = blockIdx.x, nwg = gridDim.x;
    bool ready = false;
    for (int p = 0; p < g.nphases; ++p) {
        run_phase<PARTS>(g, g.ph[p], smem, wg, nwg, ready);
        ready = false;
        if (p + 1 == g.nphases) break;
        grid_barrier(g.counter, (unsigned)(p + 1) * nwg, [&] {
            // the weight slice of the NEXT phase, when that is a G phase, depends on nothing computed in this launch: it is requested between
            // this workgroup's arrival and its first poll (this phase was PREP / R: its scratch sits behind the weight region, and every
            // wave has passed the barrier's first __syncthreads)
            if (g.ph[p + 1].kind == PH_G && wg < g.CB * g.S) { stage_weights<PARTS>(g.L[g.ph[p + 1].idx], wg, smem); ready = true; }
        });
    }
    __syncthreads();
}

}  // namespace lr
}  // namespace ide3d

// ---- host -------------------------------------------------------------------------------------------------------------------
namespace {
using namespace ide3d;
using namespace ide3d::lr;

inline int64_t align256(int64_t v) { return (v + 255) & ~int64_t(255); }

struct Layout {
    int64_t wq[MAX_LAYERS], act[MAX_LAYERS], xf[MAX_LAYERS], partial, counter, total;
    int rin[MAX_LAYERS];
};

// how many leading layers of the group fit (LDS: all images' input slots of a K slice beside the weight slice)
int layers_that_fit(int n, int C, int res0, const int32_t* ups, int nlayers, int parts) {
    if (n < 1 || C < KS || C % KS != 0 || res0 < 2) return 0;
    int res = res0, fit = 0;
    for (int l = 0; l < nlayers && l < MAX_LAYERS; ++l) {
        const int rin = res;
        if ((int64_t)n * img_pad(parts, (rin + 2) * (rin + 2)) > act_capacity_units(parts)) break;
        if (ups[l] == 2) res *= 2; else if (ups[l] != 1) break;
        const int64_t rows = (int64_t)n * (ups[l] == 2 ? (rin + 1) * (rin + 1) : res * res);
        if (rows > 1024 || (int64_t)n * (2 * rin + 1) * (2 * rin + 1) >= (1 << 24)) break;
        ++fit;
    }
    return fit;
}

bool make_layout(const ide3d_lowres_params& p, int parts, Layout& lo) {
    const int S = p.C / KS;
    int64_t off = 0;
    int res = p.res0;
    int64_t part_max = 0;
    for (int l = 0; l < p.nlayers; ++l) {
        const ide3d_lowres_layer& L = p.layers[l];
        lo.rin[l] = res;
        lo.wq[l] = off; off = align256(off + (int64_t)(p.C / BC) * S * w_units(parts) * 16);
        lo.act[l] = off; off = align256(off + (int64_t)p.n * S * img_pad(parts, (res + 2) * (res + 2)) * 16);
        const int rout = L.up == 2 ? 2 * res : res;
        const int64_t outpix = L.up == 2 ? (int64_t)(2 * res + 1) * (2 * res + 1) : (int64_t)rout * rout;
        const int64_t rows = (int64_t)p.n * (L.up == 2 ? (res + 1) * (res + 1) : rout * rout);
        part_max = std::max<int64_t>(part_max, (int64_t)S * waves_per_tile((int)((rows + 31) / 32)) * p.n * outpix * p.C * 4);
        lo.xf[l] = off; off = align256(off + (int64_t)p.n * rout * rout * p.C * 4);
        res = rout;
    }
    lo.partial = off; off = align256(off + part_max);
    lo.counter = off; off = align256(off + 256);
    lo.total = off;
    return true;
}

int resolve_parts(int arith) {
    const int a = arith ? arith : ide3d_get_conv_arithmetic();
    return a == 6 ? 3 : (a == 3 ? 2 : 0);
}

// co-resident workgroups of the persistent kernel on the current device (one per CU by LDS); cached per device
template <int PARTS>
int persistent_grid() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int c = cache[dev].load(std::memory_order_relaxed);
    if (c) return c > 0 ? c : 0;
    int per_cu = 0;
    hipDeviceProp_t prop;
    int grid = -1;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(persistent_kernel<PARTS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persistent_kernel<PARTS>, THREADS, LDS_BYTES) == hipSuccess && per_cu == 1)
        grid = prop.multiProcessorCount;
    cache[dev].store(grid > 0 ? grid : -1, std::memory_order_relaxed);
    return grid > 0 ? grid : 0;
}

template <int PARTS>
int launch_group(const ide3d_lowres_params& p, hipStream_t st) {
    Layout lo;
    make_layout(p, PARTS, lo);
    IDE3D_CHECK_ARG(p.workspace && p.workspace_bytes >= lo.total, "lowres_group: workspace too small (%lld < %lld bytes)", (long long)p.workspace_bytes, (long long)lo.total);
    unsigned char* ws = reinterpret_cast<unsigned char*>(p.workspace);
    GroupDev g{};
    g.n = p.n; g.C = p.C; g.S = p.C / KS; g.CB = p.C / BC; g.nlayers = p.nlayers; g.nheads = p.nheads; g.res0 = p.res0;
    g.x0 = p.x0; g.x0_bs = p.x0_batch_stride; g.fir = p.fir;
    g.partial = reinterpret_cast<float*>(ws + lo.partial);
    g.counter = reinterpret_cast<unsigned*>(ws + lo.counter);
    const int items = g.CB * g.S;
    int np = 0;
    int pending_head = -1;                                  // a head whose block output exists and that has not run yet
    for (int l = 0; l < p.nlayers; ++l) {
        const ide3d_lowres_layer& L = p.layers[l];
        LayerDev& D = g.L[l];
        const int rin = lo.rin[l], res = L.up == 2 ? 2 * rin : rin;
        IDE3D_CHECK_ARG(L.weight && L.styles && L.dcoefs, "lowres_group: layer %d: null weight / styles / dcoefs", l);
        D.wq = reinterpret_cast<const u32x4*>(ws + lo.wq[l]);
        D.act_in = reinterpret_cast<const u32x4*>(ws + lo.act[l]);
        D.act_out = l + 1 < p.nlayers ? reinterpret_cast<u32x4*>(ws + lo.act[l + 1]) : nullptr;
        D.styles = L.styles;
        D.styles_next = l + 1 < p.nlayers ? p.layers[l + 1].styles : nullptr;
        D.dcoefs = L.dcoefs; D.noise = L.noise; D.bias = L.bias;
        D.act_gain = L.act_gain; D.clamp = L.clamp;
        D.up = L.up; D.res = res; D.rin = rin;
        D.nslot_in = (rin + 2) * (rin + 2); D.nslot_out = (res + 2) * (res + 2);
        D.npos = L.up == 2 ? (rin + 1) * (rin + 1) : res * res;
        D.outpix = L.up == 2 ? (2 * rin + 1) * (2 * rin + 1) : res * res;
        // row bands of phase R: enough items to fill the chip; an up layer's band also reads 3 halo rows of every partial slab, so its bands
        // keep >= 4 rows (2-row bands read 2.5 x the slabs: 31.6 -> us at 16^2 -> 32^2, batch 1)
        int bands = 1;
        while (bands * 2 <= res && g.n * g.CB * bands * 2 <= 2 * kNumCU && (L.up != 2 || res / (bands * 2) >= 4)) bands *= 2;
        D.bands = bands;
        D.img_pad_in = img_pad(PARTS, D.nslot_in); D.img_pad_out = img_pad(PARTS, D.nslot_out);
        D.wpt = waves_per_tile((g.n * D.npos + 31) / 32);
        D.first = (l == 0);
        D.xf = L.head >= 0 ? reinterpret_cast<float*>(ws + lo.xf[l]) : nullptr;
        D.x_out = (l + 1 == p.nlayers) ? p.x_out : nullptr;
        D.head_side = pending_head; pending_head = -1;
        if (!L.weights_packed) {
            const int64_t total = (int64_t)items * 9 * KOCT * BC;
            hipLaunchKernelGGL((pack_kernel<PARTS>), dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, st, L.weight, p.C,
                               reinterpret_cast<u32x4*>(ws + lo.wq[l]));
        }
        g.ph[np++] = Phase{PH_G, l};
        g.ph[np++] = Phase{PH_R, l};
        if (L.head >= 0) {
            IDE3D_CHECK_ARG(L.head < p.nheads, "lowres_group: layer %d: head index out of range", l);
            const ide3d_lowres_head& Hh = p.heads[L.head];
            HeadDev& H = g.H[L.head];
            IDE3D_CHECK_ARG(Hh.w && Hh.skip && Hh.O > 0, "lowres_group: head %d: null weight / output", L.head);
            H.w = Hh.w; H.bias = Hh.bias; H.clamp = Hh.clamp; H.O = Hh.O; H.res = res;
            H.xf = D.xf; H.skip = Hh.skip;
            H.skip_prev = L.head > 0 ? p.heads[L.head - 1].skip : nullptr;
            pending_head = L.head;
        }
    }
    if (pending_head >= 0) g.ph[np++] = Phase{PH_H, pending_head};
    g.nphases = np;
    IDE3D_CHECK_ARG(p.x_out != nullptr, "lowres_group: null x_out");
    const bool persistent = p.persistent != 0;
    if (persistent) {
        const int grid = std::min(persistent_grid<PARTS>(), std::max(items, 1));
        if (grid <= 0) { set_error("lowres_group: the persistent form needs one resident workgroup per CU on this device"); return IDE3D_ENOKERNEL; }
        if (hipMemsetAsync(g.counter, 0, 2 * sizeof(unsigned), st) != hipSuccess) { set_error("lowres_group: hipMemsetAsync failed"); return IDE3D_ELAUNCH; }
        IDE3D_EXCL_LAUNCH((persistent_kernel<PARTS>), dim3(grid), THREADS, LDS_BYTES, st, g);
        IDE3D_CHECK_LAUNCH("lowres_group (persistent)");
        return IDE3D_OK;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(phase_kernel<PARTS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    for (int ph = 0; ph < np; ++ph) {
        const Phase& P = g.ph[ph];
        int grid;
        if (P.kind == PH_G) grid = items;
        else if (P.kind == PH_H) { const HeadDev& H = g.H[P.idx]; grid = std::max(1, (g.n * ((H.res * H.res + 7) / 8) * ((H.O + 7) / 8) + WAVES - 1) / WAVES); }
        else grid = g.n * g.CB * g.L[P.idx].bands;
        IDE3D_EXCL_LAUNCH((phase_kernel<PARTS>), dim3(grid), THREADS, LDS_BYTES, st, g, ph);
        IDE3D_CHECK_LAUNCH("lowres_group (phase)");
    }
    return IDE3D_OK;
}

}  // namespace

extern "C" int32_t ide3d_lowres_layers_supported(int32_t n, int32_t C, int32_t res0, const int32_t* ups, int32_t nlayers, int32_t arith) {
    const int parts = resolve_parts(arith);
    if (parts == 0 || !ups) return 0;
    return layers_that_fit(n, C, res0, ups, nlayers, parts);
}

extern "C" int64_t ide3d_lowres_workspace_bytes(const ide3d_lowres_params* pp) {
    if (!pp) return -1;
    const int parts = resolve_parts(pp->arith);
    if (parts == 0) return -1;
    Layout lo;
    make_layout(*pp, parts, lo);
    return lo.total;
}

extern "C" int ide3d_lowres_group(const ide3d_lowres_params* pp, void* stream) {
    IDE3D_CHECK_ARG(pp != nullptr, "lowres_group: null params");
    const ide3d_lowres_params& p = *pp;
    const int parts = resolve_parts(p.arith);
    if (parts == 0) { set_error("lowres_group: only the bf16x6 / bf16x3 arithmetics have this form"); return IDE3D_ENOKERNEL; }
    IDE3D_CHECK_ARG(p.n >= 1 && p.C >= lr::KS && p.C % lr::KS == 0 && p.C % 8 == 0, "lowres_group: C must be a multiple of %d", lr::KS);
    IDE3D_CHECK_ARG(p.nlayers >= 1 && p.nlayers <= lr::MAX_LAYERS && p.nheads >= 0 && p.nheads <= lr::MAX_HEADS, "lowres_group: 1..%d layers, 0..%d heads", lr::MAX_LAYERS, lr::MAX_HEADS);
    IDE3D_CHECK_ARG(p.x0 && p.fir, "lowres_group: null input / filter");
    int32_t ups[lr::MAX_LAYERS];
    for (int l = 0; l < p.nlayers; ++l) ups[l] = p.layers[l].up;
    if (layers_that_fit(p.n, p.C, p.res0, ups, p.nlayers, parts) < p.nlayers) {
        set_error("lowres_group: the layers do not fit (ask ide3d_lowres_layers_supported first)");
        return IDE3D_ENOKERNEL;
    }
    hipStream_t st = (hipStream_t)stream;
    return parts == 3 ? launch_group<3>(p, st) : launch_group<2>(p, st);
}
