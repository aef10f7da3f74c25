// Microbenchmark: how fast can 512 workgroups x 4 waves write a [4, 256, 128, 128] fp32 tensor (67 MB) when each
// workgroup owns a 128-channel x 16x16-pixel tile (the epilogue of csrc/modconv.hip), for several store shapes.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/store_burst.hip -o scripts/micro/bin/store_burst
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: dword stores, lane = pixel (32 px = 2 rows of 16), 16 channels per lane group (MFMA layout)
// mode 1: 16-byte stores, 8 lanes per channel row of 32 px (the LDS-transposed epilogue)
// mode 2: 16-byte stores, tile = 8 x 32 px (128-byte rows)
// mode 3: 16-byte stores, a wave writes whole 16x16 planes: 64 lanes = 16 rows x 4 x 16 B (1 KB contiguous per 4 rows? no: 64-B rows)
__global__ void __launch_bounds__(256, 2) burst(float* y, int mode, float v) {
    const int H = 128, W = 128, C = 256;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x, mb = b & 1, tile = (b >> 1) & 63, n = b >> 7;
    const int ty = tile >> 3, tx = tile & 7;
    float* base = y + ((size_t)n * C + mb * 128) * H * W;
    if (mode == 0) {
        const int wm = wid >> 1, wn = wid & 1, half = lane >> 5, l32 = lane & 31;
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) {
            const int pix = (wn * 4 + j) * 32 + l32, py = pix >> 4, px = pix & 15;
            const int co = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            base[(size_t)co * H * W + (ty * 16 + py) * W + tx * 16 + px] = v;
        }
    } else if (mode == 1) {
        const int wm = wid >> 1, wn = wid & 1;
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 2; ++i) for (int k = 0; k < 4; ++k) {
            const int idx = lane + 64 * k, rowl = idx >> 3, c4 = idx & 7;
            const int pix = (wn * 4 + j) * 32 + c4 * 4, py = pix >> 4, px = pix & 15;
            const int co = (wm * 2 + i) * 32 + rowl;
            *reinterpret_cast<f32x4*>(base + (size_t)co * H * W + (ty * 16 + py) * W + tx * 16 + px) = f32x4{v, v, v, v};
        }
    } else if (mode == 2) {
        const int wm = wid >> 1, wn = wid & 1;
        const int ty2 = tile >> 2, tx2 = tile & 3;             // 16 x 4 tiles of 8 x 32
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 2; ++i) for (int k = 0; k < 4; ++k) {
            const int idx = lane + 64 * k, rowl = idx >> 3, c4 = idx & 7;
            const int py = wn * 4 + j, px = c4 * 4;
            const int co = (wm * 2 + i) * 32 + rowl;
            *reinterpret_cast<f32x4*>(base + (size_t)co * H * W + (ty2 * 8 + py) * W + tx2 * 32 + px) = f32x4{v, v, v, v};
        }
    } else {
        // a wave writes 32 channels; per instruction one channel plane tile: 16 rows x 64 B
        for (int c = 0; c < 32; ++c) {
            const int co = wid * 32 + c, py = lane >> 2, px = (lane & 3) * 4;
            *reinterpret_cast<f32x4*>(base + (size_t)co * H * W + (ty * 16 + py) * W + tx * 16 + px) = f32x4{v, v, v, v};
        }
    }
}

int main() {
    const size_t bytes = (size_t)4 * 256 * 128 * 128 * 4;
    float* y; (void)hipMalloc(&y, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[4] = {"dword, MFMA layout (old epilogue)", "16 B, 8 lanes x 32 px per channel", "16 B, 8 x 32 tiles (128-B rows)", "16 B, one plane tile per store"};
    for (int mode = 0; mode < 4; ++mode) {
        burst<<<512, 256>>>(y, mode, 1.f); (void)hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0); burst<<<512, 256>>>(y, mode, 2.f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-40s %7.1f us  %6.2f TB/s\n", names[mode], best * 1e3, bytes / best / 1e9);
    }
    (void)hipMemsetAsync(y, 0, bytes, 0);
    (void)hipEventRecord(e0); (void)hipMemsetAsync(y, 0, bytes, 0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); printf("%-40s %7.1f us  %6.2f TB/s\n", "hipMemsetAsync", ms * 1e3, bytes / ms / 1e9);
    return 0;
}
