// Minimal reproducer (no torch, no library code) for the interaction of DESIGN.md section 4.2:
//   a wave executing packed fp32 VALU instructions (v_pk_fma_f32) returns wrong results while a wave of ANOTHER kernel executes
//   v_mfma_f32_32x32x16_bf16 on the same SIMD (MI355X / gfx950, ROCm 7.2).
// Two streams: A spins on MFMAs (bf16 32x32x16 fed from LDS or from registers, or fp32 32x32x2 as a control), B repeats a small
// dot-product kernel with fixed inputs, and every result of B is compared bitwise with the result it produced alone.  Both victims
// compile to v_pk_fma_f32 under plain -O3 (the second one through the SLP vectoriser); built with the packed-fp32 subtarget feature
// off they compile to v_fma_f32 / v_fmac_f32 and are the control:
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/pk_mfma_hazard.cpp -o /tmp/hazard_pk
//   hipcc -O3 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops scripts/micro/pk_mfma_hazard.cpp -o /tmp/hazard_nopk
// (scripts/micro/pk_mfma_hazard.sh builds and runs both).  Measured on MI355X, round 2: profiles/round2/pk_mfma_hazard.txt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int BF16>       // 0: fp32 MFMA, 1: bf16 MFMA on register operands, 2: bf16 MFMA fed from LDS (ds_read_b128) with a barrier per round
__global__ void __launch_bounds__(256) mfma_spin(float* out, int iters) {
    __shared__ u32x4 lds[2048];
    const unsigned l = threadIdx.x;
    u32x4 a = {0x3f803f80u + (l << 8), 0x3f003f00u, 0x3e803e80u + l, 0x3f803f00u}, b = {0x3f003f80u, 0x3e803f00u + (l << 4), 0x3f803f80u, 0x3f003f00u};
    if (BF16 == 2) { for (int i = l; i < 2048; i += 256) lds[i] = u32x4{0x3f803f80u + (unsigned)i, 0x3f003f00u, 0x3e803e80u, 0x3f803f00u + (unsigned)(i << 3)}; __syncthreads(); }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
        if (BF16 == 2) {
            u32x4 av[2], bv[2];
            av[0] = lds[(l + i * 64) & 2047]; av[1] = lds[(l + i * 64 + 512) & 2047]; bv[0] = lds[(l * 3 + i) & 2047]; bv[1] = lds[(l * 5 + i + 1024) & 2047];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[t & 1]), __builtin_bit_cast(bf16x8, bv[t >> 1]), acc[t], 0, 0, 0);
            if ((i & 7) == 7) __syncthreads();
            continue;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (BF16) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[t], 0, 0, 0);
            else      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[0]), __uint_as_float(b[0]), acc[t], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + l] = s;
}

// y[row] = sum_k A[row, k] * x[k]; a wave owns 8 rows, lanes stride over k.  PACKED: float2 multiply-adds (v_pk_fma_f32).
template <int PACKED>
__global__ void __launch_bounds__(256) dot_rows(const float* __restrict__ A, const float* __restrict__ x, float* __restrict__ y, int K) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const float* a = A + (size_t)wave * 8 * K;
    if (PACKED) {
        // the shape of the library's style GEMV: x staged in LDS, 8 rows x 16-byte loads in flight, float2 arithmetic
        __shared__ __attribute__((aligned(16))) float s_x[1024];
        for (int k = threadIdx.x; k < K; k += 256) s_x[k] = x[k];
        __syncthreads();
        f32x2 acc[8];
        for (int r = 0; r < 8; ++r) acc[r] = f32x2{0.f, 0.f};
        for (int k = lane * 4; k < K; k += 256) {
            float4 av[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) av[r] = *reinterpret_cast<const float4*>(a + r * K + k);
            const float4 xv = *reinterpret_cast<const float4*>(s_x + k);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc[r] = f32x2{av[r].x, av[r].z} * f32x2{xv.x, xv.z} + acc[r];
                acc[r] = f32x2{av[r].y, av[r].w} * f32x2{xv.y, xv.w} + acc[r];
            }
        }
        for (int r = 0; r < 8; ++r) {
            float v = acc[r][0] + acc[r][1];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) y[wave * 8 + r] = v;
        }
    } else {
        float acc0[8], acc1[8];
        for (int r = 0; r < 8; ++r) acc0[r] = acc1[r] = 0.f;
        for (int k = lane * 2; k < K; k += 128) {
            const float x0 = x[k], x1 = x[k + 1];
#pragma unroll
            for (int r = 0; r < 8; ++r) { acc0[r] = fmaf(a[r * K + k], x0, acc0[r]); acc1[r] = fmaf(a[r * K + k + 1], x1, acc1[r]); }
        }
        for (int r = 0; r < 8; ++r) {
            float v = acc0[r] + acc1[r];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) y[wave * 8 + r] = v;
        }
    }
}

int main() {
    const int K = 512, ROWS = 64 * 4 * 8, REPS = 3000;              // 64 workgroups x 4 waves x 8 rows
    std::vector<float> hA((size_t)ROWS * K), hx(K);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hx) v = rnd();
    float *dA, *dx, *dy, *dref, *dspin;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dx, K * 4)); CK(hipMalloc(&dy, (size_t)REPS * ROWS * 4)); CK(hipMalloc(&dref, ROWS * 4));
    CK(hipMalloc(&dspin, 1024 * 256 * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, hx.data(), K * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    std::vector<float> ref(ROWS), got((size_t)REPS * ROWS);
    int rc = 0;
    for (int packed = 1; packed >= 0; --packed)
        for (int neighbour = 3; neighbour >= 0; --neighbour) {       // 3: bf16 MFMA fed from LDS, 2: bf16 MFMA, 1: fp32 MFMA, 0: none
            auto victim = [&](float* out) {
                if (packed) hipLaunchKernelGGL(dot_rows<1>, dim3(64), dim3(256), 0, sb, dA, dx, out, K);
                else        hipLaunchKernelGGL(dot_rows<0>, dim3(64), dim3(256), 0, sb, dA, dx, out, K);
            };
            victim(dref);
            CK(hipStreamSynchronize(sb));
            CK(hipMemcpy(ref.data(), dref, ROWS * 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < REPS; ++i) {
                if (i % 40 == 0) {                                    // the neighbour keeps every SIMD busy for the whole series
                    if (neighbour == 3) hipLaunchKernelGGL(mfma_spin<2>, dim3(1024), dim3(256), 0, sa, dspin, 1500);
                    if (neighbour == 2) hipLaunchKernelGGL(mfma_spin<1>, dim3(1024), dim3(256), 0, sa, dspin, 1500);
                    if (neighbour == 1) hipLaunchKernelGGL(mfma_spin<0>, dim3(1024), dim3(256), 0, sa, dspin, 750);
                }
                victim(dy + (size_t)i * ROWS);
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), dy, got.size() * 4, hipMemcpyDeviceToHost));
            int bad_launches = 0; long bad_values = 0;
            for (int i = 0; i < REPS; ++i) {
                int b = 0;
                for (int r = 0; r < ROWS; ++r) b += memcmp(&got[(size_t)i * ROWS + r], &ref[r], 4) != 0;
                bad_launches += b != 0; bad_values += b;
            }
            printf("victim %-30s beside %-26s: %3d of %d launches differ from the result computed alone (%ld values)\n",
                   packed ? "A (float2 source, x in LDS)" : "B (fmaf source, SLP-packed)",
                   neighbour == 3 ? "bf16 MFMA fed from LDS" : neighbour == 2 ? "v_mfma_f32_32x32x16_bf16" : neighbour == 1 ? "v_mfma_f32_32x32x2_f32" : "nothing", bad_launches, REPS, bad_values);
            if (bad_launches) rc = 1;
        }
    return rc;
}
