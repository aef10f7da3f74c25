// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate (the ceiling of csrc/modconv.hip) for 1, 2 and 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run(int blocks_per_cu, const char* what) {
    const int cus = 256, iters = 4096;
    float* out; (void)hipMalloc(&out, (size_t)cus * blocks_per_cu * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mfma_loop<NACC><<<cus * blocks_per_cu, 256>>>(out, 16, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mfma_loop<NACC><<<cus * blocks_per_cu, 256>>>(out, iters, 1.f, 2.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 4096.0 * NACC * iters * 4.0 * cus * blocks_per_cu;
    printf("%-34s %d acc, %d wave(s)/SIMD: %8.1f us  %7.1f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", what, NACC, blocks_per_cu, ms * 1e3,
           flops / ms / 1e9, ms * 1e6 / ((double)NACC * iters * blocks_per_cu));
    (void)hipFree(out);
}

int main() {
    run<8>(1, "independent accumulators"); run<8>(2, "independent accumulators"); run<8>(3, "independent accumulators");
    run<2>(1, "two accumulators"); run<2>(2, "two accumulators");
    run<1>(1, "one accumulator (dependent chain)"); run<1>(2, "one accumulator (dependent chain)"); run<1>(4, "one accumulator (dependent chain)");
    return 0;
}
