// Is the residual of the bf16 split exact when it is formed by v_dot2c_f32_bf16 (r = x - hi as dot((hi_a, hi_b), (-1, 0)) + x)?
// Compares, bit for bit, the three pieces of every test value against the shift / mask / subtract form.  Build & run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/split_dot2_check.hip -o /tmp/split_dot2_check && /tmp/split_dot2_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ void split_ref(float a, float b, unsigned (&out)[3]) {
    for (int q = 0; q < 3; ++q) {
        const f32x2 v = {a, b};
        const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
        out[q] = pk;
        if (q < 2) { a -= __uint_as_float(pk << 16); b -= __uint_as_float(pk & 0xffff0000u); }
    }
}
__device__ void split_dot2(float a, float b, unsigned (&out)[3]) {
    // the multipliers go through scalar registers: as immediates hipcc (ROCm 7.2) emits the inline constant -1.0 for (-1, 0), which the
    // instruction does not read as bf16 (-1, 0) — every residual comes out wrong
    unsigned clo = 0x0000bf80u, chi = 0xbf800000u;
    asm volatile("" : "+s"(clo), "+s"(chi));
    const bf16x2 mlo = __builtin_bit_cast(bf16x2, clo), mhi = __builtin_bit_cast(bf16x2, chi);
    for (int q = 0; q < 3; ++q) {
        const f32x2 v = {a, b};
        const bf16x2 pk = __builtin_convertvector(v, bf16x2);
        out[q] = __builtin_bit_cast(unsigned, pk);
        if (q < 2) { a = __builtin_amdgcn_fdot2_f32_bf16(pk, mlo, a, false); b = __builtin_amdgcn_fdot2_f32_bf16(pk, mhi, b, false); }
    }
}
__global__ void check(const float* x, int n, unsigned long long* bad, float* first_bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned r[3], d[3];
    split_ref(x[2 * i], x[2 * i + 1], r);
    split_dot2(x[2 * i], x[2 * i + 1], d);
    if (r[0] != d[0] || r[1] != d[1] || r[2] != d[2]) {
        if (atomicAdd(bad, 1ull) == 0) { first_bad[0] = x[2 * i]; first_bad[1] = x[2 * i + 1]; }
    }
}
int main() {
    const int n = 1 << 24;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        unsigned bits = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
        const int mode = i & 3;
        if (mode == 0) { unsigned e = 100 + rand() % 56; bits = (bits & 0x807fffffu) | (e << 23); }          // moderate exponents
        else if (mode == 1) { unsigned e = 1 + rand() % 253; bits = (bits & 0x807fffffu) | (e << 23); }       // any normal exponent
        else if (mode == 2) { bits = (bits & 0x80000000u) | (127u << 23) | (bits & 0x7f) << (rand() % 17); }  // sparse mantissas
        else { bits = (bits & 0x807fffffu) | (127u << 23) | 0x007f8000u * (rand() & 1); }                    // ties / near-ties
        memcpy(&h[i], &bits, 4);
    }
    float *dx, *dfb; unsigned long long* dbad;
    hipMalloc(&dx, n * 4); hipMalloc(&dbad, 8); hipMalloc(&dfb, 8);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(dbad, 0, 8);
    check<<<n / 2 / 256, 256>>>(dx, n, dbad, dfb);
    unsigned long long bad = 0; float fb[2] = {0, 0};
    hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost); hipMemcpy(fb, dfb, 8, hipMemcpyDeviceToHost);
    printf("split_dot2_check: %d values, %llu mismatching pairs", n, bad);
    if (bad) printf(" (first: %a %a)", fb[0], fb[1]);
    printf("\n");
    return bad ? 1 : 0;
}
