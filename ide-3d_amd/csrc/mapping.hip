// mapping.hip — the whole mapping network z, c -> ws in ONE launch.
//
// Replaces `MappingNetwork.forward` (inversion/networks.py:287-325) in inference:
//   x = normalize_2nd_moment(z)                                   networks.py:39-40, :298
//   y = normalize_2nd_moment(embed(c))                            :301 (FullyConnectedLayer, linear)
//   x = cat([x, y]); for idx in range(num_layers): x = fc{idx}(x) :302-307 (lrelu, lr_multiplier 0.01: weight_gain =
//                                                                  lr / sqrt(in), bias_gain = lr; bias_act gain sqrt(2))
//   ws = x.unsqueeze(1).repeat(1, num_ws, 1)                      :315-316
//   ws[:, :cutoff] = w_avg.lerp(ws[:, :cutoff], psi)              :319-324
// which the framework runs as 9 small GEMMs + ~20 element-wise launches: ~30 dependent 4-8 us kernels with a gap after
// each, 300 us at the head of every synthesis pass (rocprofv3 trace of bench.py) for 8.4 MB of weights.
//
// One kernel, MAP_WGS workgroups of 256 threads: every workgroup keeps the current activation vector of all images in LDS
// and computes a slice of each layer's outputs (a wave = a few rows, lanes split K with 16-byte loads, all rows of a 256-column
// slab requested before any is used: the GEMV is latency-bound); the new activations go through a global buffer and a
// grid-wide barrier (monotonic atomic counter; all MAP_WGS = 64 workgroups must be co-resident: checked against the occupancy query
// on the host, and the spin is bounded).
#include "common.h"
#include "knobs.h"
#include <atomic>

namespace ide3d {
namespace {

constexpr int MAP_WGS = 64;
constexpr int MAP_MAX_N = 8;            // images per launch (LDS: MAP_MAX_N x MAP_MAX_K floats)
constexpr int MAP_NB = 4;               // images per register block of the GEMV
constexpr int MAP_MAX_K = 1024;
constexpr int MAP_MAX_LAYERS = 16;

struct MapArgs {
    const float* z; const float* c;
    const float* embed_w; const float* embed_b;            // [embed, c_dim], [embed]
    const float* fc_w[MAP_MAX_LAYERS]; const float* fc_b[MAP_MAX_LAYERS];
    int fc_in[MAP_MAX_LAYERS], fc_out[MAP_MAX_LAYERS];
    const float* w_avg;
    float* act;                                             // [layers][MAP_MAX_N][MAP_MAX_K]: one buffer per layer — an address is written once and read once per launch, so no stale L1 line can exist
    unsigned* counter;                                      // zeroed by the host before the launch
    float* ws;                                              // [n, num_ws, w_dim]
    int n, z_dim, c_dim, embed, layers, num_ws;
    float embed_wgain, embed_bgain, lr_mul, alpha, act_gain, psi;
    int cutoff;                                             // layers [0, cutoff) are truncated (num_ws = all)
};

// Grid-wide barrier on one monotonic counter, in the form MI355X_MICROARCH.md prescribes for inter-workgroup hand-offs (per-XCD L2s
// are not coherent and a CU's L1 is never refreshed by other CUs' stores): plain payload stores -> __syncthreads -> lane 0: agent-scope
// RELEASE fence (L2 write-back) + explicit vmcnt(0) (hipcc may drop the fence's own wait) -> relaxed arrive; RELAXED polling with
// s_sleep (an acquire per poll would invalidate this CU's L1 every iteration) -> ONE agent-scope ACQUIRE fence -> __syncthreads.
// The spin is bounded: all MAP_WGS workgroups must be co-resident for the barrier to complete (checked on the host against the
// occupancy query, `mapping_resident`), and should that ever fail — a CU mask, a foreign kernel pinning the LDS — the kernel traps after
// ~2^22 polls (seconds) with the error word set: a reported launch failure, also inside a hipGraph replay, instead of a hung GPU.
constexpr unsigned MAP_SPIN_LIMIT = 1u << 22;
// `between` (all threads) runs after this workgroup has arrived and before it starts polling: work that does not depend on the other workgroups
// (round 4: the loads of the next layer's weights) and must not delay the arrival.
template <class F>
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, F&& between) {
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
            __builtin_amdgcn_s_sleep(2);
            if (++spins > MAP_SPIN_LIMIT) { __hip_atomic_store(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_trap(); }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// rows [r0, r1) of  out[n, r] = act(sum_k x[n, k] * W[r, k] * wg + b[r] * bg)  for all n; x in LDS [n][K]
template <int RB>
__device__ __forceinline__ void gemv_rows(const float* __restrict__ s_x, int n, int K, const float* __restrict__ W, const float* __restrict__ b,
                                          int r0, int r1, float wg, float bg, float alpha, float gain, float* __restrict__ out, int out_pitch) {
    const int lane = lane_id(), wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int i0 = r0 + wid * RB; i0 < r1; i0 += nw * RB)
    for (int m0 = 0; m0 < n; m0 += MAP_NB) {              // images in register blocks of MAP_NB (weights re-read from L2 per block)
        float acc[RB][MAP_NB];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int m = 0; m < MAP_NB; ++m) acc[r][m] = 0.f;
        // all (row, 256-column slab) loads of this row group are requested before the first is used (K <= 1024: 4 slabs)
        constexpr int NS = MAP_MAX_K / (kWave * 4);
        float4 a[NS][RB];
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int k = lane * 4 + t * kWave * 4;
#pragma unroll
            for (int r = 0; r < RB; ++r)
                a[t][r] = (k < K) ? *reinterpret_cast<const float4*>(W + (int64_t)min(i0 + r, r1 - 1) * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            const int k = min(lane * 4 + t * kWave * 4, K - 4);          // a[t] is zero beyond K
#pragma unroll
            for (int m = 0; m < MAP_NB; ++m) {
                const float4 xk = *reinterpret_cast<const float4*>(s_x + min(m0 + m, n - 1) * MAP_MAX_K + k);
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r][m] += (a[t][r].x * xk.x + a[t][r].y * xk.y) + (a[t][r].z * xk.z + a[t][r].w * xk.w);
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int m = 0; m < MAP_NB; ++m) {
                float v = acc[r][m];
#pragma unroll
                for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
                const int i = i0 + r;
                if (lane == 0 && i < r1 && m0 + m < n) {
                    v = v * wg + (b ? b[i] * bg : 0.f);
                    v = (v > 0.f ? v : v * alpha) * gain;
                    out[(m0 + m) * out_pitch + i] = v;
                }
            }
    }
}

// The common shape (every wave owns ONE block of RB rows per layer and all images fit one register block: 512-wide layers on 64 workgroups,
// n <= MAP_NB): the weight block of a layer as a separate step, so that the kernel can request layer l + 1's weights BEFORE it enters the
// grid barrier of layer l (they do not depend on the activations): the ~2-3 us of HBM / MALL latency per layer overlap the barrier.
constexpr int MAP_NS = MAP_MAX_K / (kWave * 4);
template <int RB>
__device__ __forceinline__ void gemv_load_block(float4 (&a)[MAP_NS][RB], int K, const float* __restrict__ W, int i0, int r1) {
    const int lane = lane_id();
#pragma unroll
    for (int t = 0; t < MAP_NS; ++t) {
        const int k = lane * 4 + t * kWave * 4;
#pragma unroll
        for (int r = 0; r < RB; ++r)
            a[t][r] = (k < K) ? *reinterpret_cast<const float4*>(W + (int64_t)min(i0 + r, r1 - 1) * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int RB>
__device__ __forceinline__ void gemv_block(const float4 (&a)[MAP_NS][RB], const float* __restrict__ s_x, int n, int K, const float* __restrict__ b,
                                           int i0, int r1, float wg, float bg, float alpha, float gain, float* __restrict__ out, int out_pitch) {
    const int lane = lane_id();
    float acc[RB][MAP_NB];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int m = 0; m < MAP_NB; ++m) acc[r][m] = 0.f;
#pragma unroll
    for (int t = 0; t < MAP_NS; ++t) {
        const int k = min(lane * 4 + t * kWave * 4, K - 4);          // a[t] is zero beyond K
#pragma unroll
        for (int m = 0; m < MAP_NB; ++m) {
            const float4 xk = *reinterpret_cast<const float4*>(s_x + min(m, n - 1) * MAP_MAX_K + k);
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[r][m] += (a[t][r].x * xk.x + a[t][r].y * xk.y) + (a[t][r].z * xk.z + a[t][r].w * xk.w);
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int m = 0; m < MAP_NB; ++m) {
            float v = acc[r][m];
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
            const int i = i0 + r;
            if (lane == 0 && i < r1 && m < n) {
                v = v * wg + (b ? b[i] * bg : 0.f);
                v = (v > 0.f ? v : v * alpha) * gain;
                out[m * out_pitch + i] = v;
            }
        }
}

__device__ __forceinline__ float block_sum(float v, float* s_red) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if (lane_id() == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += s_red[i];
    return t;
}

__global__ void __launch_bounds__(256)
mapping_kernel(const MapArgs p) {
    __shared__ __attribute__((aligned(16))) float s_x[MAP_MAX_N * MAP_MAX_K];
    __shared__ float s_red[8];
    const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
    // ---- input stage (redundantly in every workgroup: 512 + 512 x 25 multiply-adds per image) ----
    const int K0 = p.z_dim + p.embed;
    constexpr int RB = 2;
    const int wid = tid >> 6, nw = (int)(blockDim.x >> 6);
    // one block of RB rows per wave and layer, all images in one register block: the weights of a layer are requested one step early — layer 0's
    // here, in front of the input stage; layer l + 1's inside the grid barrier of layer l
    auto one_block = [&](int l) { return l < p.layers && cdiv(p.fc_out[l], nwg) <= nw * RB && p.n <= MAP_NB; };
    float4 a_pre[MAP_NS][RB];
    bool have_pre = false;
    if (one_block(0)) {
        const int O0 = p.fc_out[0], per0 = cdiv(O0, nwg), q0 = wg * per0, q1 = min(O0, q0 + per0), j0 = q0 + wid * RB;
        if (j0 < q1) gemv_load_block<RB>(a_pre, K0, p.fc_w[0], j0, q1);
        have_pre = true;
    }
    for (int m = 0; m < p.n; ++m) {
        float sq = 0.f;
        for (int k = tid; k < p.z_dim; k += blockDim.x) { const float v = p.z[m * p.z_dim + k]; sq += v * v; }
        const float zs = (p.z_dim > 0) ? rsqrtf(block_sum(sq, s_red) / (float)p.z_dim + 1e-8f) : 0.f;
        for (int k = tid; k < p.z_dim; k += blockDim.x) s_x[m * MAP_MAX_K + k] = p.z[m * p.z_dim + k] * zs;
        float sq2 = 0.f;
        for (int e = tid; e < p.embed; e += blockDim.x) {
            float acc = 0.f;
            for (int k = 0; k < p.c_dim; ++k) acc += p.c[m * p.c_dim + k] * p.embed_w[e * p.c_dim + k];
            const float v = acc * p.embed_wgain + (p.embed_b ? p.embed_b[e] * p.embed_bgain : 0.f);
            s_x[m * MAP_MAX_K + p.z_dim + e] = v; sq2 += v * v;
        }
        const float es = (p.embed > 0) ? rsqrtf(block_sum(sq2, s_red) / (float)p.embed + 1e-8f) : 0.f;
        for (int e = tid; e < p.embed; e += blockDim.x) s_x[m * MAP_MAX_K + p.z_dim + e] *= es;
    }
    __syncthreads();
    // ---- layers ----
    int K = K0;
    for (int l = 0; l < p.layers; ++l) {
        const int O = p.fc_out[l];
        const int per = cdiv(O, nwg), r0 = wg * per, r1 = min(O, r0 + per);
        float* out = p.act + (size_t)l * MAP_MAX_N * MAP_MAX_K;
        if (one_block(l)) {
            const int i0 = r0 + wid * RB;
            if (!have_pre && i0 < r1) gemv_load_block<RB>(a_pre, K, p.fc_w[l], i0, r1);
            if (i0 < r1) gemv_block<RB>(a_pre, s_x, p.n, K, p.fc_b[l], i0, r1, p.lr_mul * rsqrtf((float)K), p.lr_mul, p.alpha, p.act_gain, out, MAP_MAX_K);
        } else if (r0 < r1)
            gemv_rows<2>(s_x, p.n, K, p.fc_w[l], p.fc_b[l], r0, r1, p.lr_mul * rsqrtf((float)K), p.lr_mul, p.alpha, p.act_gain, out, MAP_MAX_K);
        have_pre = false;
        grid_barrier(p.counter, (unsigned)(l + 1) * nwg, [&] {
            if (one_block(l + 1)) {
                const int O2 = p.fc_out[l + 1], per2 = cdiv(O2, nwg), q0 = wg * per2, q1 = min(O2, q0 + per2), j0 = q0 + wid * RB;
                if (j0 < q1) gemv_load_block<RB>(a_pre, O, p.fc_w[l + 1], j0, q1);
                have_pre = true;
            }
        });
        for (int i = tid * 4; i < p.n * O; i += blockDim.x * 4) {          // O % 4 == 0: 16-byte loads
            const int m = i / O, k = i - m * O;
            *reinterpret_cast<float4*>(s_x + m * MAP_MAX_K + k) = *reinterpret_cast<const float4*>(out + m * MAP_MAX_K + k);
        }
        __syncthreads();
        K = O;
    }
    // ---- broadcast + truncation: this workgroup writes its share of the [n, num_ws] rows ----
    const int rows = p.n * p.num_ws;
    for (int row = wg; row < rows; row += nwg) {
        const int m = row / p.num_ws, j = row - m * p.num_ws;
        const bool trunc = (p.psi != 1.0f) && j < p.cutoff;
        for (int k = tid; k < K; k += blockDim.x) {
            float v = s_x[m * MAP_MAX_K + k];
            if (trunc) {                                                               // torch.lerp(w_avg, v, psi), ATen's two-sided formula
                const float a = p.w_avg[k], d = v - a;
                v = (fabsf(p.psi) < 0.5f) ? a + p.psi * d : v - d * (1.0f - p.psi);
            }
            p.ws[(size_t)row * K + k] = v;
        }
    }
}

// ---- one launch per layer (round 6) ----------------------------------------------------------------------------------------------
// The same stages as mapping_kernel with a KERNEL BOUNDARY where that kernel has a grid barrier: launch l (0 <= l < layers) stages its input
// (l = 0: the normalised z and embedded c; else layer l - 1's activations from the workspace), computes its slice of layer l with the same
// gemv code (bit-identical results) and writes the workspace; launch `layers` broadcasts + truncates.  On MI355X a dependent kernel boundary
// costs a few microseconds like the 64-workgroup barrier does (measured: a tie, see ide3d_mapping) — and nothing needs to be co-resident: this is
// the form a device takes on which the one-launch kernel's workgroups would not all be resident.
__global__ void __launch_bounds__(256)
mapping_layer_kernel(const MapArgs p, int l) {
    __shared__ __attribute__((aligned(16))) float s_x[MAP_MAX_N * MAP_MAX_K];
    __shared__ float s_red[8];
    const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
    const int K0 = p.z_dim + p.embed;
    constexpr int RB = 2;
    const int wid = tid >> 6, nw = (int)(blockDim.x >> 6);
    const bool last = (l == p.layers);
    const int K = (l == 0) ? K0 : p.fc_out[l - 1];
    // this layer's weights first (they depend on nothing): in flight while the input is staged
    float4 a_pre[MAP_NS][RB];
    bool have_pre = false;
    int r0 = 0, r1 = 0;
    if (!last) {
        const int O = p.fc_out[l], per = cdiv(O, nwg);
        r0 = wg * per; r1 = min(O, r0 + per);
        if (per <= nw * RB && p.n <= MAP_NB) {
            const int i0 = r0 + wid * RB;
            if (i0 < r1) gemv_load_block<RB>(a_pre, K, p.fc_w[l], i0, r1);
            have_pre = true;
        }
    }
    if (l == 0) {
        for (int m = 0; m < p.n; ++m) {
            float sq = 0.f;
            for (int k = tid; k < p.z_dim; k += blockDim.x) { const float v = p.z[m * p.z_dim + k]; sq += v * v; }
            const float zs = (p.z_dim > 0) ? rsqrtf(block_sum(sq, s_red) / (float)p.z_dim + 1e-8f) : 0.f;
            for (int k = tid; k < p.z_dim; k += blockDim.x) s_x[m * MAP_MAX_K + k] = p.z[m * p.z_dim + k] * zs;
            float sq2 = 0.f;
            for (int e = tid; e < p.embed; e += blockDim.x) {
                float acc = 0.f;
                for (int k = 0; k < p.c_dim; ++k) acc += p.c[m * p.c_dim + k] * p.embed_w[e * p.c_dim + k];
                const float v = acc * p.embed_wgain + (p.embed_b ? p.embed_b[e] * p.embed_bgain : 0.f);
                s_x[m * MAP_MAX_K + p.z_dim + e] = v; sq2 += v * v;
            }
            const float es = (p.embed > 0) ? rsqrtf(block_sum(sq2, s_red) / (float)p.embed + 1e-8f) : 0.f;
            for (int e = tid; e < p.embed; e += blockDim.x) s_x[m * MAP_MAX_K + p.z_dim + e] *= es;
        }
    } else {
        const float* prev = p.act + (size_t)(l - 1) * MAP_MAX_N * MAP_MAX_K;
        for (int i = tid * 4; i < p.n * K; i += blockDim.x * 4) {
            const int m = i / K, k = i - m * K;
            *reinterpret_cast<float4*>(s_x + m * MAP_MAX_K + k) = *reinterpret_cast<const float4*>(prev + m * MAP_MAX_K + k);
        }
    }
    __syncthreads();
    if (!last) {
        float* out = p.act + (size_t)l * MAP_MAX_N * MAP_MAX_K;
        if (have_pre) {
            const int i0 = r0 + wid * RB;
            if (i0 < r1) gemv_block<RB>(a_pre, s_x, p.n, K, p.fc_b[l], i0, r1, p.lr_mul * rsqrtf((float)K), p.lr_mul, p.alpha, p.act_gain, out, MAP_MAX_K);
        } else if (r0 < r1)
            gemv_rows<2>(s_x, p.n, K, p.fc_w[l], p.fc_b[l], r0, r1, p.lr_mul * rsqrtf((float)K), p.lr_mul, p.alpha, p.act_gain, out, MAP_MAX_K);
        return;
    }
    const int rows = p.n * p.num_ws;
    for (int row = wg; row < rows; row += nwg) {
        const int m = row / p.num_ws, j = row - m * p.num_ws;
        const bool trunc = (p.psi != 1.0f) && j < p.cutoff;
        for (int k = tid; k < K; k += blockDim.x) {
            float v = s_x[m * MAP_MAX_K + k];
            if (trunc) {                                                               // torch.lerp(w_avg, v, psi), ATen's two-sided formula
                const float a = p.w_avg[k], d = v - a;
                v = (fabsf(p.psi) < 0.5f) ? a + p.psi * d : v - d * (1.0f - p.psi);
            }
            p.ws[(size_t)row * K + k] = v;
        }
    }
}

}  // namespace
}  // namespace ide3d

extern "C" int ide3d_mapping_workspace_bytes(void) {
    return (int)(ide3d::MAP_MAX_LAYERS * ide3d::MAP_MAX_N * ide3d::MAP_MAX_K * sizeof(float) + 256);
}

// 1 if MAP_WGS workgroups of the mapping kernel are co-resident on the current device with a 2x margin (the occupancy API can be one
// block per CU high on gfx950 — MI355X_MICROARCH.md "Residency and cooperative launch" — and other kernels may hold part of the chip).
// Cached PER DEVICE (the answer belongs to the device that is current when the question is asked: a CU-masked or different GPU of the
// same host must not inherit another device's answer).
static bool mapping_resident() {
    constexpr int MAX_DEV = 64;
    static std::atomic<int> cache[MAX_DEV];                               // 0 unknown, 1 no, 2 yes
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const bool cached = dev >= 0 && dev < MAX_DEV;
    if (cached) { const int c = cache[dev].load(std::memory_order_relaxed); if (c) return c == 2; }
    int per_cu = 0;
    hipDeviceProp_t prop;
    bool ok = false;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ide3d::mapping_kernel, 256, 0) == hipSuccess) {
        if (per_cu > 1) per_cu -= 1;                                       // the API's possible overcount
        ok = (int64_t)per_cu * prop.multiProcessorCount >= 2 * ide3d::MAP_WGS;
    }
    if (cached) cache[dev].store(ok ? 2 : 1, std::memory_order_relaxed);
    return ok;
}
extern "C" int ide3d_mapping_supported(void) { return 1; }          /* (since round 6: the per-layer form needs no co-residency) */

extern "C" int ide3d_mapping(const ide3d_mapping_params* q, void* stream) {
    using namespace ide3d;
    // One launch with a grid barrier per layer where its 64 workgroups are co-resident (the occupancy query, with margin); else — a CU-masked or
    // partitioned device, or IDE3D_MAPPING_PER_LAYER=1 — one launch per layer, same arithmetic.  Measured on MI355X (round 6, same box, bench.py):
    // 901-903 frames/s at batch 4 / 591 at batch 1 in one launch, 904-906 / 586-590 per layer: a tie (a ~5 us dependent launch costs what a
    // 64-workgroup barrier + the weight prefetch inside it costs).
    const bool one_launch = !knobs().mapping_per_layer && mapping_resident();
    IDE3D_CHECK_ARG(q != nullptr, "mapping: null params");
    IDE3D_CHECK_ARG(q->n > 0 && q->n <= MAP_MAX_N, "mapping: batch must be 1..%d (got %d)", MAP_MAX_N, q->n);
    IDE3D_CHECK_ARG(q->layers >= 1 && q->layers <= MAP_MAX_LAYERS, "mapping: 1..%d layers", MAP_MAX_LAYERS);
    IDE3D_CHECK_ARG(q->z_dim >= 0 && q->embed >= 0 && q->z_dim + q->embed > 0 && q->z_dim + q->embed <= MAP_MAX_K && (q->z_dim + q->embed) % 4 == 0,
                    "mapping: z_dim + embed must be a multiple of 4, at most %d", MAP_MAX_K);
    IDE3D_CHECK_ARG((q->z_dim == 0 || q->z) && (q->embed == 0 || (q->c && q->embed_w && q->c_dim > 0)), "mapping: null input pointer");
    IDE3D_CHECK_ARG(q->ws && q->workspace && q->workspace_bytes >= ide3d_mapping_workspace_bytes(), "mapping: null output / workspace too small");
    IDE3D_CHECK_ARG(q->num_ws >= 1, "mapping: num_ws must be positive");
    IDE3D_CHECK_ARG(q->truncation_psi == 1.0f || q->w_avg, "mapping: truncation needs w_avg");
    MapArgs a{};
    a.z = q->z; a.c = q->c; a.embed_w = q->embed_w; a.embed_b = q->embed_b;
    int k = q->z_dim + q->embed;
    for (int l = 0; l < q->layers; ++l) {
        IDE3D_CHECK_ARG(q->fc_w[l] && q->fc_out[l] > 0 && q->fc_out[l] <= MAP_MAX_K && q->fc_out[l] % 4 == 0 &&
                        ((reinterpret_cast<uintptr_t>(q->fc_w[l]) & 15) == 0), "mapping: layer %d: bad weight / width", l);
        a.fc_w[l] = q->fc_w[l]; a.fc_b[l] = q->fc_b[l]; a.fc_in[l] = k; a.fc_out[l] = q->fc_out[l];
        k = q->fc_out[l];
    }
    a.w_avg = q->w_avg;
    a.act = reinterpret_cast<float*>(q->workspace);
    a.counter = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(q->workspace) + (size_t)MAP_MAX_LAYERS * MAP_MAX_N * MAP_MAX_K * sizeof(float));
    a.ws = q->ws;
    a.n = q->n; a.z_dim = q->z_dim; a.c_dim = q->c_dim; a.embed = q->embed; a.layers = q->layers; a.num_ws = q->num_ws;
    a.embed_wgain = q->embed_weight_gain; a.embed_bgain = q->embed_bias_gain; a.lr_mul = q->lr_multiplier;
    a.alpha = q->alpha; a.act_gain = q->act_gain; a.psi = q->truncation_psi;
    a.cutoff = (q->truncation_cutoff < 0) ? q->num_ws : q->truncation_cutoff;
    hipStream_t st = (hipStream_t)stream;
    if (!one_launch) {
        for (int l = 0; l <= q->layers; ++l) hipLaunchKernelGGL(mapping_layer_kernel, dim3(MAP_WGS), dim3(256), 0, st, a, l);
        IDE3D_CHECK_LAUNCH("mapping (per layer)");
        return IDE3D_OK;
    }
    if (hipMemsetAsync(a.counter, 0, 2 * sizeof(unsigned), st) != hipSuccess) { set_error("mapping: hipMemsetAsync failed"); return IDE3D_ELAUNCH; }
    hipLaunchKernelGGL(mapping_kernel, dim3(MAP_WGS), dim3(256), 0, st, a);
    IDE3D_CHECK_LAUNCH("mapping");
    return IDE3D_OK;
}
