// camera.hip — the pose helpers of training/volumetric_rendering.py as two launches.
//
// The reference builds a camera pose out of ~45 one-element tensor operations (sample_camera_positions :147-193, create_cam2world_matrix
// :195-213, LookAtPoseSampler.sample :268-295): sin / cos / clamp / cross / norm / eye / repeat / slice assignment / a 4 x 4 matmul.  On the GPU
// each of them is a launch of its own: ~0.3 ms of host time and ~0.15 ms of GPU time per pose in the drivers' per-image loops
// (gen_images.py:104-106, gen_videos.py:120-124), where a batch-1 image costs 1.3 ms.  Here: one thread per camera,
//   ide3d_sphere_points : (theta, pitch) -> clamp, [arccos(1 - 2 v / pi)], r (sin phi cos theta, cos phi, sin phi sin theta)
//   ide3d_cam2world     : forward (or look-at point) + origin -> normalise, two cross products, the assembled 4 x 4 matrix
// Every operation rounds on its own, in the reference's order (no contraction into FMAs): the results differ from the ATen chain by what
// ATen's own reduction / contraction choices differ from that, a few ulp (tests/test_gpu_ops.py: 1e-6).
#include "common.h"

namespace ide3d {

#pragma clang fp contract(off)

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 normalized(V3 v) {                       // vectors / torch.norm(vectors, dim=-1, keepdim=True) (:21-25)
    const float n = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return V3{v.x / n, v.y / n, v.z / n};
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

__global__ void __launch_bounds__(64)
sphere_points_kernel(const float* __restrict__ theta, const float* __restrict__ pitch, int n, float r, int pitch_is_v,
                     float* __restrict__ pos, float* __restrict__ phi_out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float kPi = 3.14159265358979323846f;
    // torch.clamp(x, 1e-5, math.pi - 1e-5): the bounds are doubles rounded to float once
    float p = fminf(fmaxf(pitch[i], 1e-5f), (float)(3.14159265358979323846 - 1e-5));
    if (pitch[i] != pitch[i]) p = pitch[i];                           // clamp keeps NaN
    if (pitch_is_v) p = acosf(1.f - 2.f * (p * (1.f / kPi)));          // LookAtPoseSampler :284-285 (ATen divides by a host scalar as a product with its fp32 reciprocal)
    const float t = theta[i], sp = r * sinf(p);
    pos[3 * i + 0] = sp * cosf(t);
    pos[3 * i + 2] = sp * sinf(t);
    pos[3 * i + 1] = r * cosf(p);
    if (phi_out) phi_out[i] = p;
}

__global__ void __launch_bounds__(64)
cam2world_kernel(const float* __restrict__ forward, const float* __restrict__ origin, const float* __restrict__ lookat, int lookat_stride,
                 int n, float* __restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const V3 o{origin[3 * i], origin[3 * i + 1], origin[3 * i + 2]};
    V3 f;
    if (lookat) {                                                      // normalize_vecs(lookat_position - origins) (:295), then the callee's own
        const float* l = lookat + (int64_t)i * lookat_stride;
        f = normalized(V3{l[0] - o.x, l[1] - o.y, l[2] - o.z});
    } else {
        f = V3{forward[3 * i], forward[3 * i + 1], forward[3 * i + 2]};
    }
    f = normalized(f);
    const V3 left = normalized(cross(V3{0.f, 1.f, 0.f}, f));
    const V3 up = normalized(cross(f, left));
    // translation @ rotation with rotation[:3, :3] = columns (-left, up, -forward): [R | origin; 0 0 0 1] exactly
    float* m = out + (int64_t)i * 16;
    m[0] = -left.x; m[1] = up.x; m[2]  = -f.x; m[3]  = o.x;
    m[4] = -left.y; m[5] = up.y; m[6]  = -f.y; m[7]  = o.y;
    m[8] = -left.z; m[9] = up.z; m[10] = -f.z; m[11] = o.z;
    m[12] = 0.f;    m[13] = 0.f; m[14] = 0.f;  m[15] = 1.f;
}

}  // namespace ide3d

extern "C" int ide3d_sphere_points(const float* theta, const float* pitch, int32_t n, float r, int32_t pitch_is_v,
                                   float* pos, float* phi_out, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(theta && pitch && pos, "sphere_points: null pointer");
    IDE3D_CHECK_ARG(n > 0, "sphere_points: bad shape");
    hipLaunchKernelGGL(sphere_points_kernel, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, theta, pitch, n, r, pitch_is_v, pos, phi_out);
    IDE3D_CHECK_LAUNCH("sphere_points");
    return IDE3D_OK;
}

extern "C" int ide3d_cam2world(const float* forward, const float* origin, const float* lookat, int32_t lookat_stride, int32_t n,
                               float* out, void* stream) {
    using namespace ide3d;
    IDE3D_CHECK_ARG(origin && out && (forward || lookat), "cam2world: null pointer");
    IDE3D_CHECK_ARG(n > 0 && (lookat_stride == 0 || lookat_stride == 3), "cam2world: bad shape");
    hipLaunchKernelGGL(cam2world_kernel, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, forward, origin, lookat, lookat_stride, n, out);
    IDE3D_CHECK_LAUNCH("cam2world");
    return IDE3D_OK;
}
