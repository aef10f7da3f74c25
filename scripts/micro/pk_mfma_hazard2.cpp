// Round 4, DESIGN.md section 4.2: one more bounded look at the packed-fp32 / LDS-fed-MFMA interaction, stand-alone (no torch, no library).
// pk_mfma_hazard.cpp established: victim B (v_pk_fma_f32 on operands that come straight from global loads) returns wrong values beside a
// compiler-generated LDS-fed v_mfma_f32_32x32x16_bf16 loop on the same SIMD; the same victim beside register-fed MFMAs is clean.
// This file varies ONE side at a time to see whether the NEIGHBOUR can be made harmless (which would make the split arithmetics safe beside
// any foreign kernel), and what the victim needs to be hit:
//   neighbours  1 LDS-fed bf16 MFMA (baseline)          2 same, operands copied through v_mov_b32 before the MFMA
//               3 same, s_nop 7 x2 between the LDS wait and the first MFMA     4 same LDS reads, MFMA on register constants (reads kept live by VALU)
//               5 LDS-fed with ds_read_b64 pairs         6 LDS-fed v_mfma_f32_32x32x16_f16       7 LDS reads + VALU only (no MFMA)
//               8 LDS-fed, one s_nop 1 after every MFMA  9 LDS-fed 16x16x32 bf16
//               a / b : neighbour 4's loop in ONE-wave workgroups, 2 per CU: (a) claims all 512 registers of its SIMD, so the victim's waves share
//               the CU (LDS, L1, TA, scalar cache) but never the SIMD; (b) the same without the claim (positive control)
//   victims     B  fmaf source, SLP-packed (baseline)    M  same, every loaded value copied through v_mov_b32 before use
//               R  pure register v_pk_fma_f32 chain (no memory operands in the loop)     S  scalar v_fma_f32 on the same loads (control)
// For the first failing values it prints got / expected / xor so the kind of corruption is visible.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/pk_mfma_hazard2.cpp -o scripts/micro/_bin/hazard2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned vmov(unsigned v) { unsigned o; asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(v)); return o; }
__device__ __forceinline__ u32x4 vmov4(u32x4 v) { return u32x4{vmov(v[0]), vmov(v[1]), vmov(v[2]), vmov(v[3])}; }

template <int NB>
__global__ void __launch_bounds__(256) neighbour(float* out, int iters) {
    __shared__ u32x4 lds[2048];
    const unsigned l = threadIdx.x;
    for (int i = l; i < 2048; i += 256) lds[i] = u32x4{0x3f803f80u + (unsigned)i, 0x3f003f00u, 0x3e803e80u, 0x3f803f00u + (unsigned)(i << 3)};
    __syncthreads();
    f32x16 acc[4];
    f32x4 acc16[4];
    for (int t = 0; t < 4; ++t) { for (int r = 0; r < 16; ++r) acc[t][r] = 0.f; for (int r = 0; r < 4; ++r) acc16[t][r] = 0.f; }
    const u32x4 ca = {0x3f803f80u + (l << 8), 0x3f003f00u, 0x3e803e80u + l, 0x3f803f00u}, cb = {0x3f003f80u, 0x3e803f00u + (l << 4), 0x3f803f80u, 0x3f003f00u};
    unsigned live = 0;
    for (int i = 0; i < iters; ++i) {
        u32x4 av[2], bv[2];
        if (NB == 5) {
            const u32x2* l2 = reinterpret_cast<const u32x2*>(lds);
            auto ld = [&](unsigned idx) { const u32x2 a = l2[2 * idx], b = l2[2 * idx + 1]; return u32x4{a[0], a[1], b[0], b[1]}; };
            av[0] = ld((l + i * 64) & 2047); av[1] = ld((l + i * 64 + 512) & 2047); bv[0] = ld((l * 3 + i) & 2047); bv[1] = ld((l * 5 + i + 1024) & 2047);
        } else {
            av[0] = lds[(l + i * 64) & 2047]; av[1] = lds[(l + i * 64 + 512) & 2047]; bv[0] = lds[(l * 3 + i) & 2047]; bv[1] = lds[(l * 5 + i + 1024) & 2047];
        }
        if (NB == 2) { av[0] = vmov4(av[0]); av[1] = vmov4(av[1]); bv[0] = vmov4(bv[0]); bv[1] = vmov4(bv[1]); }
        if (NB == 3) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(av[0]), "+v"(av[1]), "+v"(bv[0]), "+v"(bv[1]));
        if (NB == 4 || NB == 7) {
            live ^= av[0][0] ^ av[1][1] ^ bv[0][2] ^ bv[1][3] ^ av[0][3] ^ bv[1][0];
            if (NB == 7) { live = live * 1664525u + 1013904223u; }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (NB == 7) break;
            const u32x4 a = (NB == 4) ? ca : av[t & 1], b = (NB == 4) ? cb : bv[t >> 1];
            if (NB == 6)      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a & 0x3bff3bffu), __builtin_bit_cast(f16x8, b & 0x3bff3bffu), acc[t], 0, 0, 0);
            else if (NB == 9) acc16[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc16[t], 0, 0, 0);
            else              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[t], 0, 0, 0);
            if (NB == 8) asm volatile("s_nop 1" ::: "memory");
        }
        if ((i & 7) == 7) __syncthreads();
    }
    float s = __uint_as_float(live & 0x3fffffffu);
    for (int t = 0; t < 4; ++t) { for (int r = 0; r < 16; ++r) s += acc[t][r]; for (int r = 0; r < 4; ++r) s += acc16[t][r]; }
    out[blockIdx.x * 256 + l] = s;
}


// SIMD-locality check: one-wave workgroups running neighbour 4's loop (LDS reads kept live by VALU + MFMAs on register constants, the
// variant that corrupts victim B most).  CLAIM = 1: the wave claims v255 + a255 = the whole 512-register file of its SIMD.
template <int CLAIM>
__global__ void __launch_bounds__(64) neighbour_1wave(float* out, int iters) {
    if (CLAIM) asm volatile("" ::: "v255", "a255");
    __shared__ u32x4 lds[2048];
    const unsigned l = threadIdx.x;
    for (int i = l; i < 2048; i += 64) lds[i] = u32x4{0x3f803f80u + (unsigned)i, 0x3f003f00u, 0x3e803e80u, 0x3f803f00u + (unsigned)(i << 3)};
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const u32x4 ca = {0x3f803f80u + (l << 8), 0x3f003f00u, 0x3e803e80u + l, 0x3f803f00u}, cb = {0x3f003f80u, 0x3e803f00u + (l << 4), 0x3f803f80u, 0x3f003f00u};
    unsigned live = 0;
    for (int i = 0; i < iters; ++i) {
        const u32x4 a0 = lds[(l + i * 64) & 2047], a1 = lds[(l + i * 64 + 512) & 2047], b0 = lds[(l * 3 + i) & 2047], b1 = lds[(l * 5 + i + 1024) & 2047];
        live ^= a0[0] ^ a1[1] ^ b0[2] ^ b1[3] ^ a0[3] ^ b1[0];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ca), __builtin_bit_cast(bf16x8, cb), acc[t], 0, 0, 0);
    }
    float s = __uint_as_float(live & 0x3fffffffu);
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 64 + l] = s;
}

// y[row] = sum_k A[row, k] * x[k]; a wave owns 8 rows, lanes stride over k (the fmaf source that hipcc SLP-packs into v_pk_fma_f32)
template <int V>      // 0 = B (packed), 1 = M (loaded values through v_mov first), 3 = S (scalar fma, control: no SLP)
__global__ void __launch_bounds__(256) dot_rows(const float* __restrict__ A, const float* __restrict__ x, float* __restrict__ y, int K) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const float* a = A + (size_t)wave * 8 * K;
    float acc0[8], acc1[8];
    for (int r = 0; r < 8; ++r) acc0[r] = acc1[r] = 0.f;
    for (int k = lane * 2; k < K; k += 128) {
        float x0 = x[k], x1 = x[k + 1];
        if (V == 1) { x0 = __uint_as_float(vmov(__float_as_uint(x0))); x1 = __uint_as_float(vmov(__float_as_uint(x1))); }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float a0 = a[r * K + k], a1 = a[r * K + k + 1];
            if (V == 1) { a0 = __uint_as_float(vmov(__float_as_uint(a0))); a1 = __uint_as_float(vmov(__float_as_uint(a1))); }
            if (V == 3) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc0[r]) : "v"(a0), "v"(x0));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc1[r]) : "v"(a1), "v"(x1));
            } else { acc0[r] = fmaf(a0, x0, acc0[r]); acc1[r] = fmaf(a1, x1, acc1[r]); }
        }
    }
    for (int r = 0; r < 8; ++r) {
        float v = acc0[r] + acc1[r];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) y[wave * 8 + r] = v;
    }
}

// victim R: no memory operands in the loop — 8 independent v_pk_fma_f32 chains on VALU-produced registers
__global__ void __launch_bounds__(256) pk_chain(float* __restrict__ y, int iters) {
    const unsigned g = blockIdx.x * 256 + threadIdx.x;
    f32x2 acc[8], m[8], c[8];
    for (int r = 0; r < 8; ++r) {
        acc[r] = f32x2{1.0f + (float)((g * 7 + r) & 255) / 256.f, 1.0f + (float)((g * 13 + r) & 255) / 512.f};
        m[r] = f32x2{1.0f - 1.0f / (float)(1024 + ((g + r) & 63)), 1.0f - 1.0f / (float)(2048 + ((g * 3 + r) & 63))};
        c[r] = f32x2{1.0f / (float)(4096 + r), 1.0f / (float)(8192 + (g & 31))};
    }
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int r = 0; r < 8; ++r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[r]) : "v"(m[r]), "v"(c[r]));
    float s = 0.f;
    for (int r = 0; r < 8; ++r) s += acc[r][0] - acc[r][1];
    y[g] = s;
}

int main(int argc, char** argv) {
    const int K = 512, ROWS = 64 * 4 * 8;
    const int REPS = argc > 1 ? atoi(argv[1]) : 3000;
    const char* only_v = argc > 2 ? argv[2] : "BMRS";
    const char* only_n = argc > 3 ? argv[3] : "0123456789ab";
    const int NOUT = 64 * 256;                     // outputs per victim launch (rows for the dot kernels use the first ROWS)
    std::vector<float> hA((size_t)ROWS * K), hx(K);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hx) v = rnd();
    float *dA, *dx, *dy, *dref, *dspin;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dx, K * 4)); CK(hipMalloc(&dy, (size_t)REPS * NOUT * 4)); CK(hipMalloc(&dref, NOUT * 4));
    CK(hipMalloc(&dspin, 1024 * 256 * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, hx.data(), K * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    std::vector<float> ref(NOUT), got((size_t)REPS * NOUT);
    const char vict[] = "BMRS";
    for (int vi = 0; vi < 4; ++vi) {
        if (!strchr(only_v, vict[vi])) continue;
        const int nout = vict[vi] == 'R' ? NOUT : ROWS;
        for (int nb = 0; nb <= 11; ++nb) {
            if (!strchr(only_n, nb < 10 ? '0' + nb : 'a' + nb - 10)) continue;
            auto victim = [&](float* out) {
                switch (vict[vi]) {
                case 'B': hipLaunchKernelGGL(dot_rows<0>, dim3(64), dim3(256), 0, sb, dA, dx, out, K); break;
                case 'M': hipLaunchKernelGGL(dot_rows<1>, dim3(64), dim3(256), 0, sb, dA, dx, out, K); break;
                case 'S': hipLaunchKernelGGL(dot_rows<3>, dim3(64), dim3(256), 0, sb, dA, dx, out, K); break;
                case 'R': hipLaunchKernelGGL(pk_chain, dim3(64), dim3(256), 0, sb, out, 200); break;
                }
            };
            victim(dref);
            CK(hipStreamSynchronize(sb));
            CK(hipMemcpy(ref.data(), dref, nout * 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < REPS; ++i) {
                if (i % 40 == 0) {
                    const int it = 1500;
                    switch (nb) {
                    case 1: hipLaunchKernelGGL(neighbour<1>, dim3(1024), dim3(256), 0, sa, dspin, it); break;
                    case 2: hipLaunchKernelGGL(neighbour<2>, dim3(1024), dim3(256), 0, sa, dspin, it); break;
                    case 3: hipLaunchKernelGGL(neighbour<3>, dim3(1024), dim3(256), 0, sa, dspin, it); break;
                    case 4: hipLaunchKernelGGL(neighbour<4>, dim3(1024), dim3(256), 0, sa, dspin, it); break;
                    case 5: hipLaunchKernelGGL(neighbour<5>, dim3(1024), dim3(256), 0, sa, dspin, it); break;
                    case 6: hipLaunchKernelGGL(neighbour<6>, dim3(1024), dim3(256), 0, sa, dspin, it); break;
                    case 7: hipLaunchKernelGGL(neighbour<7>, dim3(1024), dim3(256), 0, sa, dspin, it * 4); break;
                    case 8: hipLaunchKernelGGL(neighbour<8>, dim3(1024), dim3(256), 0, sa, dspin, it); break;
                    case 10: hipLaunchKernelGGL(neighbour_1wave<1>, dim3(512), dim3(64), 0, sa, dspin, it * 4); break;
                    case 11: hipLaunchKernelGGL(neighbour_1wave<0>, dim3(512), dim3(64), 0, sa, dspin, it * 4); break;
                    case 9: hipLaunchKernelGGL(neighbour<9>, dim3(1024), dim3(256), 0, sa, dspin, it * 2); break;
                    }
                }
                victim(dy + (size_t)i * nout);
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), dy, (size_t)REPS * nout * 4, hipMemcpyDeviceToHost));
            int bad_launches = 0; long bad_values = 0; int shown = 0;
            for (int i = 0; i < REPS; ++i) {
                int b = 0;
                for (int r = 0; r < nout; ++r)
                    if (memcmp(&got[(size_t)i * nout + r], &ref[r], 4) != 0) {
                        ++b;
                        if (shown < 4) {
                            unsigned g, e; memcpy(&g, &got[(size_t)i * nout + r], 4); memcpy(&e, &ref[r], 4);
                            printf("    launch %4d out %5d: got %.9g (%08x) expected %.9g (%08x) xor %08x\n", i, r, got[(size_t)i * nout + r], g, ref[r], e, g ^ e);
                            ++shown;
                        }
                    }
                bad_launches += b != 0; bad_values += b;
            }
            printf("victim %c beside neighbour %x: %4d of %d launches differ (%ld values)\n", vict[vi], nb, bad_launches, REPS, bad_values);
            fflush(stdout);
        }
    }
    return 0;
}
