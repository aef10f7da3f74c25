// lds_mask_bench — does the cost of a ds_read_b128 on gfx950 depend on how many lanes are active?
//
//   hipcc -O2 --offload-arch=gfx950 -o scripts/micro/bin/lds_mask_bench scripts/micro/lds_mask_bench.cpp && scripts/micro/bin/lds_mask_bench
//
// One 8-wave workgroup per CU (the blending waves of the tri-plane gather), every wave issues ITER x 12 ds_read_b128: each group of 8 lanes
// reads one 128-byte texel (4 channels per lane), texel index from a per-group table, conflict-free.  Variants by EXEC mask: all 64 lanes;
// lane groups 0-3 only (lanes 0-31); every other lane group; one lane group in four; every other LANE.  If the LDS pipe skips inactive
// lanes the time scales with the active share - the question behind a register tap cache in triplane_tile.hip (round 5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[32 * 1024];            // 128 KB: 1024 texels of 128 bytes
    for (int i = threadIdx.x; i < 32 * 1024; i += 512) lds[i] = (float)(i & 255);
    __syncthreads();
    const int lane = threadIdx.x & 63, grp = lane >> 3, sub = lane & 7, wid = threadIdx.x >> 6;
    bool on = true;
    if (MODE == 1) on = grp < 4;
    if (MODE == 2) on = (grp & 1) == 0;
    if (MODE == 3) on = (grp & 3) == 0;
    if (MODE == 4) on = (lane & 1) == 0;
    if (MODE == 5) on = grp == 0;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned t = (wid * 131 + grp * 17) & 1023;
    if (on) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                const f4 v = *reinterpret_cast<const f4*>(lds + ((t + r * 37) & 1023) * 32 + sub * 4);
                acc += v;
            }
            t = (t * 5 + 1) & 1023;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == -1.f) out[threadIdx.x] = acc.x;
}

template <int MODE>
static float run(float* d, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5 * 1e3f;
}

int main() {
    float* d; CK(hipMalloc(&d, 4096));
    const int iters = 2000;
    const char* names[6] = {"all 64 lanes", "lane groups 0-3 (lanes 0-31)", "every other lane group", "one lane group in four", "every other lane", "one lane group"};
    float t[6] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters), run<5>(d, iters)};
    const double reads = 8.0 * iters * 12;                      // ds_read_b128 per SIMD pair ... per CU: 8 waves
    for (int m = 0; m < 6; ++m)
        printf("%-34s %9.1f us   %6.1f cycles per ds_read_b128 of one wave (8 waves share the CU's LDS; 2.4 GHz assumed)   x%.2f of all-lanes\n", names[m], t[m],
               t[m] * 2400.0 / (iters * 12) , t[m] / t[0]);
    (void)reads;
    return 0;
}
