// What does HBM give a kernel with the gather's read / write mix?  Streams with 16-byte accesses, 1 KB per wave instruction:
//   copy 160 MB -> 160 MB | write only 201 MB | read 120 MB + write 201 MB (the tri-plane gather's algorithmic mix), plain and
//   non-temporal stores, 256 x 1024 and 2048 x 256 threads.   hipcc -O3 --offload-arch=gfx950 scripts/micro/rw_mix.hip -o scripts/micro/bin/rw_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void stream(const f32x4* __restrict__ src, size_t nread, f32x4* __restrict__ dst, size_t nwrite, float* sink) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t n = nread > nwrite ? nread : nwrite;
    for (size_t i = tid; i < n; i += nthr) {
        if (i < nread) acc += src[i];
        if (i < nwrite) {
            const f32x4 v = {acc[0], (float)i, acc[2], 1.f};
            if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
        }
    }
    if (acc[0] == 123.456f) *sink = acc[1];
}

int main() {
    const size_t MB = 1 << 20;
    f32x4 *src, *dst; float* sink;
    hipMalloc(&src, 256 * MB); hipMalloc(&dst, 256 * MB); hipMalloc(&sink, 4);
    hipMemset(src, 0, 256 * MB); hipMemset(dst, 0, 256 * MB);
    struct Case { const char* name; size_t r, w; } cases[] = {
        {"copy 160 -> 160 MB", 160 * MB, 160 * MB}, {"write only 201 MB", 0, 201 * MB}, {"read only 201 MB", 201 * MB, 0},
        {"read 120 + write 201 MB (gather mix)", 120 * MB, 201 * MB}, {"read 72 + write 197 MB (gather, measured HBM traffic)", 72 * MB, 197 * MB}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& c : cases)
        for (int nt = 0; nt < 2; ++nt)
            for (int shape = 0; shape < 2; ++shape) {
                const dim3 grid(shape ? 2048 : 256), block(shape ? 256 : 1024);
                std::vector<float> t;
                for (int it = 0; it < 60; ++it) {
                    hipEventRecord(e0);
                    if (nt) stream<true><<<grid, block>>>(src, c.r / 16, dst, c.w / 16, sink); else stream<false><<<grid, block>>>(src, c.r / 16, dst, c.w / 16, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (it >= 20) t.push_back(ms * 1e3f);
                }
                std::sort(t.begin(), t.end());
                const float med = t[t.size() / 2];
                printf("%-58s %s %-10s median %7.2f us  %6.2f TB/s\n", c.name, nt ? "nt   " : "plain", shape ? "2048x256" : "256x1024", med, (c.r + c.w) / med / 1e6);
            }
    return 0;
}
