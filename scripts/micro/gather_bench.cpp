// gather_bench — torch-free A/B harness for the tri-plane gather through the C ABI (include/ide3d_hip.h).
//
//   hipcc -O2 -o scripts/micro/bin/gather_bench scripts/micro/gather_bench.cpp -ldl
//   scripts/micro/bin/gather_bench <coords.bin> <lib.so> [<lib2.so> ...]
//
// coords.bin = float32 [4, 393216, 3]: the camera-frustum sample positions bench.py's `bench_gather` uses (written by
// scripts/micro/make_gather_coords.py).  For every library: flat kernel and ray-grid kernel at the benchmark shape (N = 4
// images, C = 32, 256 x 256 planes of N(0,1) values, 64 x 64 rays x 96 steps), ray-grid output compared with the flat
// kernel's bit for bit, 30 timed launches each with HIP events.  Starts in well under a second, so a dozen kernel variants
// fit into one gpurun call.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int (*flat_fn)(const float*, const int64_t*, int32_t, int32_t, int32_t, int32_t, const float*, int64_t, float*, void*);
typedef int (*rays_fn)(const float*, const int64_t*, int32_t, int32_t, int32_t, int32_t, const float*, int64_t, float*, int32_t, int32_t, int32_t, void*);
typedef const char* (*err_fn)();

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s coords.bin lib.so [...]\n", argv[0]); return 2; }
    const int n = 4, C = 32, H = 256, W = 256, RH = 64, RW = 64, S = 96;
    const int64_t m = (int64_t)RH * RW * S;
    const int iters = getenv("GB_ITERS") ? atoi(getenv("GB_ITERS")) : 30;
    std::vector<float> coords((size_t)n * m * 3);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(coords.data(), 4, coords.size(), f) != coords.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    const float cscale = getenv("GB_COORD_SCALE") ? (float)atof(getenv("GB_COORD_SCALE")) : 1.0f;
    for (auto& v : coords) v *= cscale;
    // planes: channels_last [n, H, W, 3C] = torch strides (3C*H*W, 1, W*3C, 3C) of a [n, 3C, H, W] tensor
    std::vector<float> planes((size_t)n * H * W * 3 * C);
    std::mt19937 rng(0); std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : planes) v = nd(rng);
    const int64_t stride[4] = {(int64_t)3 * C * H * W, 1, (int64_t)W * 3 * C, 3 * C};
    float *d_planes, *d_coords, *d_out, *d_ref;
    CK(hipMalloc(&d_planes, planes.size() * 4)); CK(hipMalloc(&d_coords, coords.size() * 4));
    CK(hipMalloc(&d_out, (size_t)n * m * C * 4)); CK(hipMalloc(&d_ref, (size_t)n * m * C * 4));
    CK(hipMemcpy(d_planes, planes.data(), planes.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_coords, coords.data(), coords.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    const double algo = (3.0 * C * H * W + 3.0 * m + (double)C * m) * 4 * n;
    std::vector<float> h_out((size_t)n * m * C), h_ref((size_t)n * m * C);
    for (int a = 2; a < argc; ++a) {
        void* lib = dlopen(argv[a], RTLD_NOW | RTLD_LOCAL);
        if (!lib) { fprintf(stderr, "dlopen %s: %s\n", argv[a], dlerror()); continue; }
        flat_fn flat = (flat_fn)dlsym(lib, "ide3d_triplane_sample");
        rays_fn rays = (rays_fn)dlsym(lib, "ide3d_triplane_sample_rays");
        err_fn last = (err_fn)dlsym(lib, "ide3d_last_error");
        if (!flat || !rays) { fprintf(stderr, "%s: symbols missing\n", argv[a]); continue; }
        for (int tiled = 0; tiled < 2; ++tiled) {
            float* dst = tiled ? d_out : d_ref;
            CK(hipMemsetAsync(dst, 0xff, (size_t)n * m * C * 4, st));
            auto call = [&]() { return tiled ? rays(d_planes, stride, n, C, H, W, d_coords, m, dst, RH, RW, S, st) : flat(d_planes, stride, n, C, H, W, d_coords, m, dst, st); };
            int rc = 0;
            for (int i = 0; i < 3 && !rc; ++i) rc = call();
            if (rc) { fprintf(stderr, "%s: launch failed rc=%d %s\n", argv[a], rc, last ? last() : ""); break; }
            CK(hipStreamSynchronize(st));
            std::vector<hipEvent_t> e0(iters), e1(iters);
            for (int i = 0; i < iters; ++i) { CK(hipEventCreate(&e0[i])); CK(hipEventCreate(&e1[i])); }
            for (int i = 0; i < iters; ++i) { CK(hipEventRecord(e0[i], st)); call(); CK(hipEventRecord(e1[i], st)); }
            CK(hipStreamSynchronize(st));
            std::vector<float> ms(iters);
            for (int i = 0; i < iters; ++i) { CK(hipEventElapsedTime(&ms[i], e0[i], e1[i])); hipEventDestroy(e0[i]); hipEventDestroy(e1[i]); }
            if (getenv("GB_VERBOSE")) { printf("  series us:"); for (int i = 0; i < iters; i += (i < 40 ? 4 : 20)) printf(" %d:%.1f", i, ms[i] * 1e3); printf("\n"); }
            std::sort(ms.begin(), ms.end());
            double avg = 0; for (float v : ms) avg += v; avg /= iters;
            const char* eq = "";
            if (tiled) {
                CK(hipMemcpy(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(h_ref.data(), d_ref, h_ref.size() * 4, hipMemcpyDeviceToHost));
                eq = memcmp(h_out.data(), h_ref.data(), h_out.size() * 4) == 0 ? "  bit-equal to flat" : "  *** DIFFERS FROM FLAT ***";
            }
            printf("%-48s %-5s avg %7.2f us  min %7.2f  med %7.2f  %6.0f GB/s  %5.1f%% of 8 TB/s%s\n", argv[a], tiled ? "tile" : "flat",
                   avg * 1e3, ms[0] * 1e3, ms[iters / 2] * 1e3, algo / (avg * 1e-3) / 1e9, algo / (avg * 1e-3) / 8e12 * 100, eq);
            fflush(stdout);
        }
        // phase trace of libraries built with -DIDE3D_TT_TRACE (cycle stamps of one wave, see triplane_tile.hip)
        typedef int (*dbg_fn)(unsigned long long*);
        if (dbg_fn dbg = (dbg_fn)dlsym(lib, "ide3d_debug_tt")) {
            unsigned long long v[256];
            if (dbg(v) == 0) {
                printf("  chunk: cycles between stamps [0-1 bbox reduce, 1-2 barrier 1, 2-3 region table + issue A/B, 3-4 tap table + fetch + LDS fill, 4-5 barrier 2, 5-6 blend], total\n");
                for (int ch = 0; ch < 24; ++ch) {
                    const unsigned long long* r = v + ch * 8;
                    if (!r[0]) continue;
                    printf("  %2d:", ch);
                    for (int i = 0; i < 6; ++i) printf(" %6llu", r[i + 1] - r[i]);
                    printf("  total %6llu", r[6] - r[0]);
                    if (ch > 0 && v[(ch - 1) * 8]) printf("  (start-to-start %6llu)", r[0] - v[(ch - 1) * 8]);
                    printf("\n");
                }
            }
        }
        // producer / consumer kernel (round 3): stamps of wave 0 of each role of one workgroup, per iteration
        if (dbg_fn dbgp = (dbg_fn)dlsym(lib, "ide3d_debug_tt_pc")) {
            static unsigned long long v[3][32][8];
            if (dbgp(&v[0][0][0]) == 0 && v[0][1][0]) {
                printf("  pc iteration: T [tap table, taps + coords, exchange + regions, barrier wait] total | F [issue, commit, barrier wait] total | B [blend, barrier wait] total\n");
                for (int it = 0; it < 24; ++it) {
                    const unsigned long long *t = v[0][it], *f = v[1][it], *b = v[2][it];
                    if (!t[0]) continue;
                    printf("  %2d:", it);
                    for (int i = 0; i < 4; ++i) printf(" %6lld", (long long)(t[i + 1] - t[i]));
                    printf("  total %6lld |", (long long)(t[4] - t[0]));
                    for (int i = 0; i < 3; ++i) printf(" %6lld", (long long)(f[i + 1] - f[i]));
                    printf("  total %6lld |", (long long)(f[3] - f[0]));
                    for (int i = 0; i < 2; ++i) printf(" %6lld", (long long)(b[i + 1] - b[i]));
                    printf("  total %6lld\n", (long long)(b[2] - b[0]));
                }
            }
        }
        if (dbg_fn dbgr = (dbg_fn)dlsym(lib, "ide3d_debug_tt_pc_r")) {
            static unsigned long long v[32][8];
            if (dbgr(&v[0][0]) == 0 && v[1][0]) {
                printf("  region-builder wave (IDE3D_PC_RWAVE builds): [copy + issue next loads, index arithmetic, reduce, table, barrier wait] total\n");
                for (int it = 0; it < 24; ++it) {
                    const unsigned long long* r = v[it];
                    if (!r[0]) continue;
                    printf("  %2d:", it);
                    for (int i = 0; i < 5; ++i) printf(" %6lld", (long long)(r[i + 1] - r[i]));
                    printf("  total %6lld\n", (long long)(r[5] - r[0]));
                }
            }
        }
        if (dbg_fn dbgw = (dbg_fn)dlsym(lib, "ide3d_debug_tt_wg")) {
            static unsigned long long w[1024][4];
            if (dbgw(&w[0][0]) == 0 && w[0][0]) {
                unsigned long long cmin = ~0ull, cmax = 0, csum = 0, r0 = ~0ull, r1 = 0, part = 0; int nw = 0;
                for (int i = 0; i < 1024 && w[i][0]; ++i, ++nw) {
                    cmin = std::min(cmin, w[i][0]); cmax = std::max(cmax, w[i][0]); csum += w[i][0];
                    r0 = std::min(r0, w[i][1]); r1 = std::max(r1, w[i][2]); part += w[i][3] >> 32;
                }
                printf("  per workgroup (%d): cycles min %llu avg %llu max %llu; first entry -> last end %.2f us (100 MHz clock); chunks with a plane not staged: %llu\n",
                       nw, cmin, csum / nw, cmax, (double)(r1 - r0) / 100.0, part);
                // histogram of workgroup durations and of start skew
                int hist[8] = {0}; double skew_max = 0;
                for (int i = 0; i < nw; ++i) { int b = (int)((w[i][0] - cmin) * 8 / (cmax - cmin + 1)); hist[b]++; skew_max = std::max(skew_max, (double)(w[i][1] - r0) / 100.0); }
                printf("  duration histogram (8 bins min..max):"); for (int b = 0; b < 8; ++b) printf(" %d", hist[b]); printf("; latest start +%.2f us\n", skew_max);
                printf("  slowest workgroups:"); 
                for (int k = 0; k < 8; ++k) { int bi = 0; for (int i = 0; i < nw; ++i) if (w[i][0] > w[bi][0]) bi = i; printf(" %d:%llu(np %llu)", bi, w[bi][0], w[bi][3] >> 32); w[bi][0] = 1; }
                printf("\n");
            }
        }
        // keep the library mapped (its kernels are registered with the runtime)
    }
    return 0;
}
